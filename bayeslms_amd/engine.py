"""Training engine: flat parameter / gradient / momentum buffers, fused clip+SGD, and the
data-parallel gradient exchange (one process per GPU, RCCL over xGMI).

The reference trains in one process on one GPU (train.py:306-438).  New here: the step is
data-parallel over global-batch columns; all gradients live in ONE flat fp32
buffer that the wgrad kernels accumulate into in place, cut into buckets that are all-reduced on a
side stream as soon as backward has produced them (reverse parameter order), overlapping the
remaining backward GEMMs; then one fused global-norm clip + SGD-momentum kernel pass over the
three flat buffers (train.py:419-420,466 semantics, averaged over ranks).
"""
import math

import torch
import torch.distributed as dist

from . import ops


class FlatBuffers:
    """Re-homes every distinct parameter of ``model`` into one contiguous buffer and gives each a
    ``.grad`` view into a second one.  Tied tensors (decoder.weight is encoder.weight) appear once."""

    def __init__(self, model):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev, dt = self.params[0].device, self.params[0].dtype
        sizes = [p.numel() for p in self.params]
        # 16-byte aligned slots so every tensor keeps the vectorised kernel paths
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.total = off
        self.flat_param = torch.zeros(off, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(off, device=dev, dtype=dt)
        self.flat_mom = torch.zeros(off, device=dev, dtype=dt)
        for p, o, n in zip(self.params, self.offsets, sizes):
            view = self.flat_param[o:o + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_grad[o:o + n].view_as(p)

    def zero_grad(self):
        self.flat_grad.zero_()


class GradReducer:
    """Bucketed asynchronous all-reduce of a flat gradient buffer.

    ``mark_ready(param)`` is called by the backward kernels' launchers (ops._notify_grad) once a
    parameter's gradient has been enqueued; a parameter with several contributions (tied
    embedding/decoder: decoder wgrad first, embedding scatter last) is ready after the last one.
    When every parameter of a bucket is ready the bucket is all-reduced (sum) on ``comm_stream``
    behind an event recorded on the compute stream.  Works on any device / backend (the gloo tests
    drive it with CPU tensors); on the GPU the backend is "nccl" = RCCL.
    """

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, expected=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        # buckets = contiguous runs of parameters, built from the END of the buffer (backward order)
        per = max(1, bucket_bytes // 4)
        self.buckets = []  # (start, end, [param indices])
        idxs, end = [], flat.total
        for i in range(len(flat.params) - 1, -1, -1):
            idxs.append(i)
            if end - flat.offsets[i] >= per or i == 0:
                self.buckets.append((flat.offsets[i], end, list(idxs)))
                idxs, end = [], flat.offsets[i]
        self.bucket_of = {}
        for b, (_, _, ids) in enumerate(self.buckets):
            for i in ids:
                self.bucket_of[id(flat.params[i])] = b
        # how many backward kernels write each gradient is learnt from the first step (calibration:
        # count notifications, reduce everything at finish()); ``expected`` can pin it up front
        self.expected = {id(p): 0 for p in flat.params}
        self.calibrating = expected is None
        for p, n in (expected or {}).items():
            self.expected[id(p)] = n
        if expected is not None:
            for k in self.expected:
                self.expected[k] = self.expected[k] or 1
        self.reset()

    def reset(self):
        self.seen = {k: 0 for k in self.expected}
        # parameters nothing ever writes (expected 0 after calibration) never hold a bucket back
        self.pending = [sum(1 for i in ids if self.calibrating or self.expected[id(self.flat.params[i])] > 0)
                        for _, _, ids in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []

    def mark_ready(self, param):
        k = id(param)
        if k not in self.seen:
            return
        self.seen[k] += 1
        if self.calibrating or self.seen[k] != self.expected[k]:
            return
        b = self.bucket_of[k]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.world == 1:
            return
        s, e, _ = self.buckets[b]
        view = self.flat.flat_grad[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever was not triggered (parameters without gradient this step), wait for all
        buckets and make the compute stream wait for the communication stream."""
        if self.calibrating:
            self.expected = dict(self.seen)
            self.calibrating = False
        for b in range(len(self.buckets)):
            self._launch(b)
        for h in self.handles:
            h.wait()
        if self.cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


class Trainer:
    """One optimisation step = forward + CE + KL*seq_len/len(train_data) + backward + (all-reduce)
    + clip + SGD, as train.py:315-420 does, for any of the model families."""

    def __init__(self, model, lr, clip, momentum=0.9, kl_scale=0.0, seed=1111, rank=0, world=1, global_batch=None,
                 bucket_bytes=32 << 20, fused_kl=True, weight_decay=0.0):
        self.model = model
        self.lr, self.clip, self.momentum = lr, clip, momentum
        self.weight_decay = weight_decay  # torch.optim.SGD(weight_decay=...) of train_search_bayes.py:391-392
        self.kl_scale = kl_scale
        self.rank, self.world = rank, world
        self.flat = FlatBuffers(model)
        self.reducer = GradReducer(self.flat, bucket_bytes)
        ops.set_grad_ready_hook(self.reducer.mark_ready if world > 1 else None)
        self.table = ops.PtrTable([self.flat.flat_param], [self.flat.flat_grad], [self.flat.flat_mom])
        self.step_no = 0
        self.first = True
        model.set_seed(seed)
        self.fused_kl = fused_kl
        self.kl_layers = self._find_kl_layers()

    def _find_kl_layers(self):
        from .model import BayesLinear
        return [m for m in self.model.modules() if isinstance(m, BayesLinear)]

    def reset_optimizer(self, lr):
        """train.py:503-505: LR halving re-creates SGD, i.e. the momentum buffers start from zero."""
        self.lr = lr
        self.first = True

    def step(self, data, targets, hidden=None, kl_fn=None, philox_step=None):
        """-> (loss tensor, kl tensor, new hidden).  ``kl_fn(model)`` returns the KL term train.py
        would add for this configuration (train.py:335-399), or None.  ``philox_step`` overrides the
        noise / dropout stream index (default: the optimisation step count)."""
        m = self.model
        m.train()
        m.set_step(self.step_no if philox_step is None else philox_step)
        B = data.shape[1]
        m.set_columns(self.rank * B, self.world * B)
        self.flat.zero_grad()
        fused = self.fused_kl and kl_fn is not None and len(self.kl_layers) > 0 and getattr(kl_fn, "fusable", False)
        for lyr in self.kl_layers:
            lyr.fused_kl_lambda = self.kl_scale if fused else 0.0
        if hidden is None:
            out = m(data)
        else:
            out, hidden = m(data, hidden)
        V = out.shape[-1]
        mle, _ = ops.cross_entropy(out.view(-1, V), targets, unit_grad=True)
        kl = None
        if kl_fn is not None:
            kl = kl_fn(m) * self.kl_scale
        if kl is None or fused:
            loss = mle + kl.detach() if kl is not None else mle
            mle.backward()
        else:
            loss = mle + kl
            loss.backward()
        self.reducer.finish()
        ops.clip_sgd(self.table, self.clip, self.lr, self.momentum, self.first, 1.0 / self.world, self.weight_decay)
        self.first = False
        self.step_no += 1
        return loss.detach(), (kl.detach() if kl is not None else None), hidden


def evaluate(model, source, seq_len, eval_batch_size=None):
    """Eval-mode loss per token exactly as train.py:441-458 sums it."""
    from .data import get_batch
    from .model import repackage_hidden
    model.eval()
    total = torch.zeros((), device=source.device, dtype=torch.float64)
    hidden = model.init_hidden(source.shape[1]) if hasattr(model, "init_hidden") else None
    with torch.no_grad():
        for i in range(0, source.size(0) - 1, seq_len):
            data, targets = get_batch(source, i, seq_len)
            if hidden is None:
                out = model(data)
            else:
                out, hidden = model(data, hidden)
                hidden = repackage_hidden(hidden)
            loss, _ = ops.cross_entropy(out.view(-1, out.shape[-1]), targets)
            total += len(data) * loss.double()
    return float(total.item()) / (len(source) - 1)


def perplexity(loss):
    return math.exp(loss)

"""Training engine: flat parameter / gradient / momentum buffers, fused clip+SGD, and the
data-parallel gradient exchange (one process per GPU, RCCL over xGMI).

The reference trains in one process on one GPU (train.py:306-438).  New here: the step is
data-parallel over global-batch columns; all gradients live in ONE flat fp32
buffer that the wgrad kernels accumulate into in place, cut into buckets that are all-reduced on a
side stream as soon as backward has produced them (reverse parameter order), overlapping the
remaining backward GEMMs; then one fused global-norm clip + SGD-momentum kernel pass over the
three flat buffers (train.py:419-420,466 semantics, averaged over ranks).
"""
import contextlib
import datetime
import math
import os
import sys
import time

import torch
import torch.distributed as dist

from . import ops


RCCL_CHANNELS_DEFAULT = 16


def pin_rccl_channels(n=None):
    """Call BEFORE dist.init_process_group("nccl").  Pins RCCL's channel count (one 256-thread workgroup per channel and
    collective, each holding a CU's registers beside the backward GEMMs) with NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS unless
    the user has set them: ``n`` or BLM_RCCL_CHANNELS or 16.  Why 16 (profiles/r04_comm_occupancy_rehearsal.txt): the
    cfg3 gradient is 202 MB per 20 ms step -- 18 GB/s of bus bandwidth would hide it under backward -- and in the one-GPU
    rehearsal a 16-workgroup stand-in at 150-300 GB/s costs the step 1.1 ms (5 %), 8 workgroups cannot stream a bucket
    fast enough (2.3 ms, most of it exposed) and 32 cost 0.7-1.9 ms.  0 leaves RCCL's own choice.
    -> the settings in force (recorded in bench.py's ``comm``)."""
    n = int(os.environ.get("BLM_RCCL_CHANNELS", RCCL_CHANNELS_DEFAULT if n is None else n))
    if n > 0:
        os.environ.setdefault("NCCL_MIN_NCHANNELS", str(n))
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(n))
    return {k: os.environ.get(k) for k in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS")}


DIST_TIMEOUT_S = 180.0
_T0 = time.time()
_STAGE = {"last": "process start"}


def heartbeat(stage, rank=None):
    """One line on stderr per stage of a multi-rank run (``[blm rank R +S.Ss] stage``) and the stage remembered for
    ``last_stage()``: a run that stops -- a rendezvous nobody joins, a collective one rank never enters -- says where.
    bench.py's launcher parent parses these lines; they never go to stdout (the one JSON line lives there)."""
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    _STAGE["last"] = stage
    sys.stderr.write("[blm rank %d +%.1fs] %s\n" % (rank, time.time() - _T0, stage))
    sys.stderr.flush()


def last_stage():
    return _STAGE["last"]


def dist_timeout_s(timeout_s=None):
    return float(os.environ.get("BLM_DIST_TIMEOUT_S", DIST_TIMEOUT_S if timeout_s is None else timeout_s))


def init_distributed(backend, device=None, timeout_s=None, **kw):
    """dist.init_process_group with a BOUNDED rendezvous and collective timeout (``timeout_s`` or BLM_DIST_TIMEOUT_S or
    180 s, not torch's 10 / 30 minutes: a rank that never arrives ends the job inside the timeout -- the store raises, and the
    RCCL watchdog aborts the process when a collective outlives it), RCCL's channel count pinned first
    (pin_rccl_channels), and two heartbeats: ``rendezvous ok`` when the group exists, ``first all-reduce ok`` after an
    8-byte all-reduce whose result is checked (the first real contact of the ranks' communicators -- on RCCL the ring /
    tree set-up over xGMI happens here).  -> the RCCL channel settings in force (None for another backend).
    The reference has no counterpart: one process, one device (train.py:136)."""
    timeout = datetime.timedelta(seconds=dist_timeout_s(timeout_s))
    rccl_env = None
    if backend == "nccl":
        rccl_env = pin_rccl_channels()  # before RCCL reads its environment
        if os.environ.get("NCCL_MAX_NCHANNELS"):
            sys.stderr.write("[blm] RCCL channels pinned: %s (BLM_RCCL_CHANNELS=0 leaves RCCL's own choice)\n" % (rccl_env,))
        dist.init_process_group("nccl", device_id=device, timeout=timeout, **kw)  # nccl == RCCL on ROCm
    else:
        dist.init_process_group(backend, timeout=timeout, **kw)
    world, rank = dist.get_world_size(), dist.get_rank()
    heartbeat("rendezvous ok (world %d, backend %s, timeout %.0f s)" % (world, backend, timeout.total_seconds()), rank)
    on_dev = backend == "nccl"
    t = torch.ones(1, dtype=torch.float64, device=device if on_dev else "cpu")
    dist.all_reduce(t)
    if float(t.item()) != float(world):
        raise RuntimeError("first all-reduce returned %r on rank %d, world %d" % (float(t.item()), rank, world))
    heartbeat("first all-reduce ok", rank)
    return rccl_env


def allreduce_busbw(nbytes, reps=5, device=None, group=None):
    """Stand-alone all-reduce of ``nbytes`` (fp32 sum), ``reps`` timed repetitions after one warm-up, nothing else on the
    device: what the links give the gradient exchange when it does not share the chip with backward.
    busbw = 2 (N-1)/N x bytes / time (the ring's per-link traffic, RCCL-tests' convention).  Collective: every rank calls."""
    world = dist.get_world_size(group)
    n = max(1, int(nbytes) // 4)
    cuda = device is not None and torch.device(device).type == "cuda" and dist.get_backend(group) == "nccl"  # gloo: host tensors
    x = torch.zeros(n, dtype=torch.float32, device=device if cuda else "cpu")
    dist.all_reduce(x, group=group)
    if cuda:
        torch.cuda.synchronize()
    dist.barrier(group)
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(x, group=group)
    if cuda:
        torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    t = torch.tensor([el], dtype=torch.float64, device=device if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    el = float(t.item())
    alg = n * 4 / el / 1e9
    return {"mb": round(n * 4 / 1e6, 2), "reps": reps, "ms": round(1e3 * el, 4), "algbw_gbps": round(alg, 2),
            "busbw_gbps": round(alg * 2.0 * (world - 1) / max(world, 1), 2)}


def rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # noqa: BLE001
        return None


def rccl_channels():
    """The channel count RCCL was pinned to (0: unknown / RCCL's own choice -- the planner then keeps the whole chip)."""
    try:
        return max(0, int(os.environ.get("NCCL_MAX_NCHANNELS", "0")))
    except ValueError:
        return 0


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


class _HostStagedReduce:
    """Handle of one all-reduce on a host-staged transport (gloo with device gradients): the transport reduced a host copy;
    wait() waits for it and writes the sums back on the communication stream with a blocking copy."""

    def __init__(self, work, host, view, stream):
        self.work, self.host, self.view, self.stream = work, host, view, stream

    def wait(self):
        self.work.wait()
        with torch.cuda.stream(self.stream):
            self.view.copy_(self.host)  # pageable source: returns when the bytes are on the device


class GradReducer:
    """Bucketed asynchronous all-reduce of a flat gradient buffer.

    ``mark_ready(param)`` is called once per gradient contribution: by the backward kernels' launchers
    (ops._notify, for gradients the wgrad kernels accumulate in place) and by a post-accumulate-grad hook on
    every parameter (``hook_autograd``, for gradients autograd's AccumulateGrad adds: LSTM weights, GP
    coefficients, VNN noise rows, ...).  A parameter with several contributions is ready after the last one;
    how many there are is counted in the first step (calibration: everything is reduced at ``finish()``).
    When every parameter of a bucket is ready the bucket is all-reduced (sum) on ``comm_stream`` behind an
    event recorded on the compute stream.  Safety rules (a stale or racing all-reduce makes ranks diverge
    silently, so none of these is inferred):
      * a parameter that was never marked in the calibration step holds its bucket back until ``finish()``;
      * a contribution that arrives after its bucket was launched raises (the write pattern changed:
        rebuild the Trainer or pass ``overlap=False``);
      * ``overlap=False`` reduces every bucket at ``finish()``.
    Works on any device / backend (the gloo tests drive it with CPU tensors); on the GPU the backend is
    "nccl" = RCCL.
    """

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, expected=None, overlap=True, comm_cus=0, collective=None,
                 comm_plan=None, comm_gbps=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat.flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.overlap = overlap
        # CU contention: RCCL runs one channel workgroup per channel (gfx950: 256 threads, ~280 registers per
        # lane, 19.7 KB LDS -- a CU that hosts one cannot also host an eight-wave GEMM workgroup, and a launch that is
        # exactly one round of 256 one-per-CU workgroups spills a second round: 1.74x, tools/comm_occupancy_rehearsal.py).
        # comm_plan "window" (default when comm_cus > 0): every bucket opens / extends the planner's comm window by its
        # expected time on the links (bytes / comm_gbps); the GEMMs enqueued behind it take the plans measured beside a
        # resident channel stand-in (csrc/gemm_plans_comm.inc) until their modelled time has used the window up.
        # "narrow": the planner counts on 256 - comm_cus CUs from the first bucket of a step until finish() (measured:
        # loses 1.5-10 % in situ -- buckets are in flight for a tenth of backward; kept for A/B).  "off": nothing.
        self.comm_cus = int(comm_cus) if self.cuda else 0
        comm_plan = comm_plan or os.environ.get("BLM_COMM_PLAN", "window")
        if comm_plan not in ("window", "narrow", "off"):
            raise ValueError("comm_plan / BLM_COMM_PLAN must be window, narrow or off, not %r" % (comm_plan,))
        self.comm_plan = comm_plan if self.comm_cus > 0 else "off"
        self.comm_gbps = float(os.environ.get("BLM_COMM_GBPS", "100") if comm_gbps is None else comm_gbps)
        self._narrowed = False
        # rehearsal (tools/comm_occupancy_rehearsal.py): fn(view, comm_stream) stands in for dist.all_reduce, world 1
        self.collective = collective
        self.late = None  # LateRows, set by the Trainer
        self.no_dense = set()  # buckets whose only tensor gets ALL of its gradient through LateRows (untied encoder)
        self.measure = False  # record (backward end, communication end) event pairs on the compute stream
        self.exposed_events = []
        # diagnostics, one step at a time (bench.py's comm block): every collective of the step bracketed on the communication
        # stream -> bucket_report().  The bracket makes the communication stream wait for each collective in turn, so it is
        # never on inside a timed region
        self.measure_buckets = False
        self.bucket_events, self.bucket_marks = [], []
        # BLM_DP_CHECK=1 (diagnosis, off by default): in front of every collective of the gradient exchange the ranks compare
        # (sequence number, element count) through a small all-gather; a disagreement raises with every rank's pair instead of
        # ending in a transport error (gloo) or a hang until the collective timeout (RCCL)
        # collectives ordered on the device's streams (RCCL) or staged through the host with a copy back on the transport's own stream (gloo)
        self.stream_ordered = not (dist.is_initialized() and dist.get_backend(group) != "nccl")
        self.check = os.environ.get("BLM_DP_CHECK", "0") == "1"
        self.seq = 0
        # buckets = contiguous runs of parameters, built from the END of the buffer (backward order).  Two refinements
        # for what is exposed at the end of backward:
        #  * a tensor of a bucket's size or more travels ALONE (the tied encoder / decoder weight, 67.6 MB at cfg3: with
        #    LateRows its dense half is complete after the FIRST backward kernel -- sharing a bucket with the first
        #    layer's parameters would hold it, and 100+ MB of all-reduce, until the last one);
        #  * the parameters that become ready last (the lowest offsets) go in quarter-size buckets: the final all-reduce,
        #    which nothing can hide, moves 8 MB instead of 32.
        per = max(1, bucket_bytes // 4)
        n = len(flat.params)
        sizes = [(flat.offsets[i + 1] if i + 1 < n else flat.total) - flat.offsets[i] for i in range(n)]
        first_small = next((i for i in range(n) if sizes[i] < per), n)  # first parameter behind the leading big tensor(s)
        tail_end = flat.offsets[first_small] + 2 * per if first_small < n else 0  # offsets below this: quarter-size buckets
        self.buckets = []  # (start, end, [param indices])
        idxs, end = [], flat.total
        for i in range(n - 1, -1, -1):
            if sizes[i] >= per and idxs:  # close the run behind a big tensor first
                self.buckets.append((flat.offsets[i + 1], end, list(idxs)))
                idxs, end = [], flat.offsets[i + 1]
            idxs.append(i)
            limit = per // 4 if (flat.offsets[i] < tail_end and sizes[i] < per) else per
            if end - flat.offsets[i] >= max(1, limit) or i == 0:
                self.buckets.append((flat.offsets[i], end, list(idxs)))
                idxs, end = [], flat.offsets[i]
        self.bucket_of = {}
        for b, (_, _, ids) in enumerate(self.buckets):
            for i in ids:
                self.bucket_of[id(flat.params[i])] = b
        self.expected = {id(p): 0 for p in flat.params}
        self.calibrating = expected is None
        for p, n in (expected or {}).items():
            self.expected[id(p)] = n
        self._hooks = []
        self.reset()

    def hook_autograd(self):
        """Every leaf parameter reports the gradients autograd accumulates into it (AccumulateGrad does not run for
        the ``None`` our in-place wgrad Functions return, so nothing is counted twice)."""
        if self._hooks:
            return
        for p in self.flat.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self.mark_ready))

    def unhook(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def reset(self):
        self.seen = {k: 0 for k in self.expected}
        # a never-marked parameter keeps its bucket for finish(): "never written" is not "ready"
        self.pending = [len(ids) for _, _, ids in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self.last_reduced_elems = getattr(self, "reduced_elems", 0)  # elements the finished step exchanged
        self.reduced_elems = 0

    def mark_ready(self, param):
        k = id(param)
        if k not in self.seen:
            return
        self.seen[k] += 1
        if self.calibrating or not self.overlap:
            return
        b = self.bucket_of[k]
        if self.launched[b]:
            raise ops.BayesLMError(
                "GradReducer: a gradient of a %s parameter was written after its bucket had been all-reduced (the "
                "set of backward kernels changed since the calibration step); build the Trainer with overlap=False "
                "for models whose graph changes between steps" % (tuple(param.shape),))
        if self.seen[k] != self.expected[k]:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.world == 1 and self.collective is None:
            return
        if b in self.no_dense and self.late is not None and self.late.used:
            return  # nothing dense was written: the compact exchange carries the whole gradient of this tensor
        s, e, _ = self.buckets[b]
        self._all_reduce(self.flat.flat_grad[s:e])

    def _check_agreement(self, view, what):
        self.seq += 1
        if os.environ.get("BLM_DP_TRACE"):  # diagnosis: this rank's sequence of collectives, one line each, no synchronisation added
            import threading
            with open("%s.rank%d" % (os.environ["BLM_DP_TRACE"], dist.get_rank() if dist.is_initialized() else 0), "a") as f:
                f.write("%d %s %d thread=%s\n" % (self.seq, what, view.numel(), threading.current_thread().name))
        if not self.check or self.world <= 1 or self.collective is not None:
            return
        on_dev = self.cuda and dist.get_backend(self.group) == "nccl"
        mine = torch.tensor([self.seq, view.numel()], dtype=torch.int64, device=view.device if on_dev else "cpu")
        every = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(every, mine, group=self.group)
        pairs = [tuple(int(x) for x in e.tolist()) for e in every]
        if any(p != pairs[0] for p in pairs):
            raise ops.BayesLMError("GradReducer (BLM_DP_CHECK): the ranks disagree on collective %s: (sequence number, elements) "
                                   "per rank = %s" % (what, pairs))

    def _all_reduce(self, view, what="bucket"):
        self._check_agreement(view, what)
        self.reduced_elems += view.numel()
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=self.measure_buckets)
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            if self.comm_plan == "window":
                ops.gemm_comm_window(view.numel() * view.element_size() / (self.comm_gbps * 1e3))
                self._narrowed = True
            elif self.comm_plan == "narrow" and not self._narrowed:
                ops.set_gemm_cus(max(8, ops.get_gemm_cus() - self.comm_cus))
                self._narrowed = True
            if self.collective is not None:
                self.collective(view, self.comm_stream)
                return
            if not self.stream_ordered:  # gloo on device tensors: staged through the host HERE (see LateRows.begin), same handles and order
                with torch.cuda.stream(self.comm_stream):
                    t0 = time.perf_counter()
                    host = view.to("cpu")  # waits for the communication stream, which waits for `ev`
                    h = _HostStagedReduce(dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True), host, view,
                                          self.comm_stream)
                if self.measure_buckets:
                    h.wait()
                    self.bucket_marks.append((view.numel() * view.element_size(), t0, time.perf_counter()))
                else:
                    self.handles.append(h)
                return
            with torch.cuda.stream(self.comm_stream):
                if self.measure_buckets:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                    h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    h.wait()  # the communication stream (not the host, on RCCL) waits for this collective
                    e1.record(self.comm_stream)
                    self.bucket_events.append((view.numel() * view.element_size(), ev, e0, e1))
                    return
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            if self.measure_buckets:
                t0 = time.perf_counter()
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                self.bucket_marks.append((view.numel() * view.element_size(), t0, time.perf_counter()))
                return
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Launch whatever was not triggered (parameters without gradient this step), wait for all
        buckets and make the compute stream wait for the communication stream."""
        if self.calibrating:
            self.expected = dict(self.seen)
            self.calibrating = False
            if self.late is not None and self.late.used:
                # an UNTIED encoder weight under LateRows has no dense contribution at all (the tied one has the
                # decoder's): when it travels alone, its bucket would be all-reduced as zeros, fully exposed
                k = id(self.late.weight)
                b = self.bucket_of.get(k)
                if b is not None and self.expected[k] == 0 and len(self.buckets[b][2]) == 1:
                    self.no_dense.add(b)
        ev0 = None
        if (self.measure or self.measure_buckets) and self.cuda:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream())
            self.bwd_end_event = ev0
        self.bwd_end_mark = time.perf_counter()
        for b in range(len(self.buckets)):
            self._launch(b)
        for h in self.handles:
            h.wait()
        if self.cuda and (self.world > 1 or self.collective is not None):
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self._narrowed:  # whatever is enqueued from here on starts after the last bucket has landed
            if self.comm_plan == "window":
                ops.gemm_comm_window(0)
            else:
                ops.set_gemm_cus(0)
            self._narrowed = False
        if ev0 is not None and self.measure:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream())
            self.exposed_events.append((ev0, ev1))
        if self.late is not None:
            self.late.apply()
        self.reset()

    def bucket_report(self):
        """After ONE step taken with ``measure_buckets`` raised: one entry per collective of that step, in launch order --
        its size, launch -> done on the communication stream (ms), and where it sat relative to the end of backward
        (``ready_before_bwd_end_ms``: how long before the last backward kernel the bucket's gradients were complete;
        ``done_after_bwd_end_ms`` > 0: that much of it was exposed).  Synchronises; clears the record."""
        out = []
        if self.cuda and self.stream_ordered:
            torch.cuda.synchronize()
            end = getattr(self, "bwd_end_event", None)
            for nbytes, ready, e0, e1 in self.bucket_events:
                r = {"mb": round(nbytes / 1e6, 2), "ms": round(e0.elapsed_time(e1), 4)}
                if end is not None:
                    r["ready_before_bwd_end_ms"] = round(ready.elapsed_time(end), 4)
                    r["done_after_bwd_end_ms"] = round(end.elapsed_time(e1), 4)
                out.append(r)
        else:
            for nbytes, t0, t1 in self.bucket_marks:
                out.append({"mb": round(nbytes / 1e6, 4), "ms": round(1e3 * (t1 - t0), 4),
                            "ready_before_bwd_end_ms": round(1e3 * (self.bwd_end_mark - t0), 4),
                            "done_after_bwd_end_ms": round(1e3 * (t1 - self.bwd_end_mark), 4)})
        self.bucket_events, self.bucket_marks = [], []
        return out

    def comm_exposed_ms(self):
        """Average time per step the compute stream sat between the last backward kernel and the end of the
        gradient exchange (what the overlap did not hide).  Synchronises."""
        if not self.exposed_events:
            return None
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self.exposed_events]
        return sum(v) / len(v)


class LateRows:
    """The embedding half of the (usually tied) encoder gradient, exchanged as a compact matrix.

    The tied encoder/decoder weight (model.py:1240; V x d = 67.6 MB at cfg3) gets its gradient from the FIRST
    backward kernel (decoder wgrad) and from the LAST one (embedding scatter), so as one tensor its all-reduce
    starts when backward is over and is fully exposed.  Here the decoder contribution is reduced with the
    other buckets while backward runs, and the embedding contribution -- at most T*B distinct rows per rank --
    goes to a compact (U, d) matrix, U = number of distinct token ids of the GLOBAL batch this step, which is
    all-reduced at the end of backward and added into the flat gradient (blm_rows_gather_add).  The row
    numbering is agreed at the START of the step, off the critical path: ids are all-gathered on the
    communication stream, presence bitmap -> prefix sum -> slot per vocabulary row (fixed-size tensor ops, no
    host synchronisation); only U travels to the host (pinned copy + event, read when backward reaches the
    embedding).  The sums are the same as the dense exchange's (order of float additions aside).
    """

    def __init__(self, reducer, weight):
        self.red = reducer
        self.weight = weight
        V, D = weight.shape
        self.V, self.D = V, D
        dev = weight.device
        self.cuda = weight.is_cuda
        self.gworld = dist.get_world_size(reducer.group)
        # persistent buffers (allocated on the compute stream's pool, used on both streams under events)
        self.buf = torch.zeros(V * D, device=dev, dtype=weight.dtype)
        self.mark = torch.zeros(V, device=dev, dtype=torch.int64)
        self.csum = torch.zeros(V, device=dev, dtype=torch.int64)
        self.slot_of_vocab = torch.full((V,), -1, device=dev, dtype=torch.int64)
        self.slots = self.allids = None
        self.count_host = torch.zeros(1, dtype=torch.int64, pin_memory=self.cuda)
        self.event = torch.cuda.Event() if self.cuda else None
        self.ids = None
        self.U = 0
        self.used = False

    def begin(self, ids):
        """Step start: agree on the compact row numbering of this step's global batch."""
        red = self.red
        self.ids, self.used, self.U = ids, False, 0
        n = ids.numel()
        if self.slots is None or self.slots.numel() != n:
            self.slots = torch.empty(n, device=ids.device, dtype=torch.int64)
            self.allids = torch.empty(self.gworld * n, device=ids.device, dtype=torch.int64)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            red.comm_stream.wait_event(ev)
            ctx = torch.cuda.stream(red.comm_stream)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            mine = ids.reshape(-1).clamp(0, self.V - 1)
            red._check_agreement(mine, "all-gather of the step's token ids")
            if self.cuda and not red.stream_ordered:
                # a host-staged transport (gloo: the one-GPU rehearsals) is handed HOST tensors.  c10d's own staging of device
                # tensors copies the result back on a stream of its own, and with 4 ranks sharing a GPU the numbering below read
                # `allids` while that copy was still landing in 1 of ~30 steps -- the ranks then disagreed on U and the compact
                # all-reduce died with a size mismatch (round 5: 4-8 of 16 `bench.py --gpus 4 --backend gloo` runs; 0 of 16 since).
                # Neither wait() on an explicit work object nor a device-wide synchronisation behind it closed the window.
                host = torch.empty(self.allids.numel(), dtype=torch.int64)
                dist.all_gather_into_tensor(host, mine.cpu(), group=red.group)
                self.allids.copy_(host)
            else:  # RCCL: ordered on this stream by construction
                dist.all_gather_into_tensor(self.allids, mine, group=red.group)
            self.mark.zero_()
            self.mark.index_fill_(0, self.allids, 1)
            torch.cumsum(self.mark, 0, out=self.csum)
            self.slot_of_vocab.copy_(self.csum).sub_(1)
            self.slot_of_vocab.masked_fill_(self.mark == 0, -1)
            torch.index_select(self.slot_of_vocab, 0, mine, out=self.slots)
            self.count_host.copy_(self.csum[-1:], non_blocking=True)
            if self.cuda:
                self.event.record(red.comm_stream)

    def sink(self, weight, ids):
        """ops.set_embed_grad_sink callback (runs inside backward, at the embedding's node)."""
        if weight is not self.weight or self.ids is None or self.used or ids.data_ptr() != self.ids.data_ptr() \
                or ids.shape != self.ids.shape:
            return None
        self.used = True
        if self.cuda:
            self.event.synchronize()  # the host needs U; the GPU passed this point at the start of the step
            torch.cuda.current_stream().wait_event(self.event)
        self.U = int(self.count_host.item())
        view = self.buf[: self.U * self.D]
        view.zero_()
        return view, self.slots.view(ids.shape), self.U, self._done

    def _done(self):
        self.red._all_reduce(self.buf[: self.U * self.D], "compact embedding rows (U = %d)" % self.U)

    def apply(self):
        """After the exchange: flat gradient rows += compact rows."""
        if not self.used or self.U == 0:
            self.ids = None
            return
        g = self.weight.grad
        src = self.buf[: self.U * self.D].view(self.U, self.D)
        if self.cuda:
            ops.rows_gather_add(g, self.slot_of_vocab, src, self.U)
        else:
            rows = torch.nonzero(self.slot_of_vocab >= 0).reshape(-1)
            g.index_add_(0, rows, src[self.slot_of_vocab[rows]])
        self.ids = None


class Trainer:
    """One optimisation step = forward + CE + KL*seq_len/len(train_data) + backward + (all-reduce)
    + clip + SGD, as train.py:315-420 does, for any of the model families."""

    def __init__(self, model, lr, clip, momentum=0.9, kl_scale=0.0, seed=1111, rank=0, world=1, global_batch=None,
                 bucket_bytes=32 << 20, fused_kl=True, weight_decay=0.0, overlap=True, late_rows=True, comm_cus=None,
                 collective=None, comm_plan=None, comm_gbps=None):
        self.model = model
        self.lr, self.clip, self.momentum = lr, clip, momentum
        self.weight_decay = weight_decay  # torch.optim.SGD(weight_decay=...) of train_search_bayes.py:391-392
        self.kl_scale = kl_scale
        self.rank, self.world = rank, world
        self.flat = FlatBuffers(model)
        if comm_cus is None:  # RCCL: one channel workgroup per channel (pin_rccl_channels); gloo moves bytes on the host
            comm_cus = rccl_channels() if (world > 1 and dist.is_initialized() and dist.get_backend() == "nccl") else 0
        self.reducer = GradReducer(self.flat, bucket_bytes, overlap=overlap, comm_cus=comm_cus if overlap else 0,
                                   collective=collective, comm_plan=comm_plan, comm_gbps=comm_gbps)
        hooked = world > 1 or collective is not None
        ops.set_grad_ready_hook(self.reducer.mark_ready if hooked else None)
        ops.set_embed_grad_sink(None)
        if hooked:
            self.reducer.hook_autograd()
            enc = getattr(getattr(model, "encoder", None), "weight", None)
            if late_rows and enc is not None and any(enc is p for p in self.flat.params) and dist.is_initialized():
                self.reducer.late = LateRows(self.reducer, enc)
                ops.set_embed_grad_sink(self.reducer.late.sink)
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
        if self.reducer.late is not None:
            self.reducer.late.begin(data)
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


def evaluate(model, source, seq_len, eval_batch_size=None, rank=0, world=1, group=None):
    """Eval-mode loss per token exactly as train.py:441-458 sums it.  Data parallel (``world`` > 1): the columns of the
    evaluation batch are independent streams (batchify, train.py:167-179; an LSTM's carried state is per column), so rank r
    evaluates columns [r C / W, (r + 1) C / W) and the token-weighted sums meet in one 8-byte all-reduce -- the reference's
    single process walks the whole stream, and with W ranks doing that each the validation pass would be the part of an epoch
    that does not scale (it is ~1 % of a one-GPU epoch of the AMI recipe, ~10 % of an 8-GPU one).  Every rank returns the
    same value up to the all-reduce (callers that branch on it take rank 0's: train.ValidationSchedule)."""
    from .data import get_batch
    from .model import repackage_hidden
    model.eval()
    cols = source.shape[1]
    lo, hi = (rank * cols) // world, ((rank + 1) * cols) // world
    mine = source if world == 1 else source[:, lo:hi].contiguous()
    total = torch.zeros((), device=source.device, dtype=torch.float64)
    hidden = model.init_hidden(mine.shape[1]) if (hasattr(model, "init_hidden") and hi > lo) else None
    # the decoder returns the per-token NLL itself when it can (ops.linear_nll: the (T*B, V) logits are never stored);
    # BLM_EVAL_FUSED_NLL=0 keeps decoder + cross-entropy kernel
    dec = getattr(model, "decoder", None)
    fused = (os.environ.get("BLM_EVAL_FUSED_NLL", "1") != "0" and dec is not None and hasattr(dec, "nll_targets")
             and ops.linear_nll_supported(dec.weight, dec.bias))
    # A model without carried state (the Transformers) sees every window on its own: G full windows are evaluated as ONE batch of
    # G x columns independent columns (same positions, same causal mask per column) -- the reference's eval batch of 10-20 columns
    # gives the matrix cores 1280-2560 rows per product, G of them ~16384 (headline model, 12 windows of 20 x 128: 1.085 M tokens/s
    # window by window, 1.178 M in threes, 1.197 M in sixes, tools/eval_windows_probe.py).  total = sum over windows of len * mean is unchanged up
    # to the order of the additions.  BLM_EVAL_WINDOWS=1 walks the windows one by one as train.py:441-458 does.
    n_full = max(0, (source.size(0) - 1) // seq_len)  # windows of exactly seq_len rows; the ragged last one goes alone
    n_win = 1
    if hi > lo:
        rows = 16384 if hidden is None else 2400  # recurrent: the time loop is issued step by step, three windows of 700 rows pay, more do not
        n_win = int(os.environ.get("BLM_EVAL_WINDOWS", "0")) or max(1, min(max(n_full, 1), rows // max(1, seq_len * (hi - lo))))
    # A recurrent model carries its state from window to window (repackage_hidden), so G consecutive windows ARE one window of
    # G seq_len steps: the same recurrence step for step, with the input / decoder products over G times the rows (the LSTM
    # language models' evaluation batch gives them 700 rows per window) and the layer wavefront over a longer stretch
    # (configs[1]'s model, 12 windows of 20 x 35: 668 k tokens/s window by window, 765 k in threes; no better and host-bound beyond).
    stride = seq_len * (n_win if hidden is not None else 1)
    with torch.no_grad():
        try:
            i, left = 0, n_full
            while hidden is None and n_win > 1 and left > 1 and hi > lo:
                g = min(n_win, left)
                starts = range(i, i + g * seq_len, seq_len)
                data = torch.cat([mine[k:k + seq_len] for k in starts], 1)
                targets = torch.cat([mine[k + 1:k + 1 + seq_len] for k in starts], 1).reshape(-1)
                if fused:
                    dec.nll_targets = targets
                out = model(data)
                loss = out.mean() if fused else ops.cross_entropy(out.view(-1, out.shape[-1]), targets)[0]
                total += (g * seq_len) * loss.double()  # equally sized windows: sum_k len * mean_k = G len * mean of all
                i, left = i + g * seq_len, left - g
            for i in range(i, source.size(0) - 1, stride):
                if hi <= lo:
                    break  # more ranks than columns: nothing of this stream is mine
                data, targets = get_batch(mine, i, stride)
                if fused:
                    dec.nll_targets = targets
                if hidden is None:
                    out = model(data)
                else:
                    out, hidden = model(data, hidden)
                    hidden = repackage_hidden(hidden)
                loss = out.mean() if fused else ops.cross_entropy(out.view(-1, out.shape[-1]), targets)[0]
                total += len(data) * loss.double()
        finally:
            if fused:
                dec.nll_targets = None
    if world > 1:  # sum_r (columns of r) * (its per-window means, summed) / all columns = the whole batch's per-window means, summed
        total = total * float(hi - lo) / float(cols)
        on_dev = dist.get_backend(group) == "nccl"
        t = total.reshape(1) if on_dev else total.reshape(1).cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        total = t[0]
    return float(total.item()) / (len(source) - 1)


def perplexity(loss):
    return math.exp(loss)

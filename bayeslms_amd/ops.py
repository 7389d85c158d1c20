"""Operators of the hot path as torch.autograd.Functions over the C ABI.

Each Function launches hand-written gfx950 kernels from libbayeslm_hip.so on
torch's current HIP stream with raw device pointers.  PyTorch is used for
device memory and the autograd tape only.  Parameter gradients are accumulated
straight into ``param.grad`` by the wgrad kernels (BLM_GEMM_ACCUMULATE), so the
flat gradient buffer the data-parallel all-reduce works on is filled in place;
the Functions return ``None`` for parameters.

There is no CPU path: every entry point raises on non-GPU tensors.
"""
import ctypes as C
import math
import os
import threading
import weakref
from dataclasses import dataclass

import torch

from . import _lib as L
from ._lib import BayesLMError, check, dev_tensor, lib, ptr, stream

__all__ = ["Drop", "NoiseSpec", "ResidualLink", "linear", "bayes_linear", "ffn", "ffn_gp", "attention", "attention_qkv", "add_dropout_ln",
           "embed", "add_pe", "dropout", "cross_entropy", "kl_mean", "philox_normal", "sample_weight", "sampled", "lstm_layer", "lstm_stack2", "lstm_stack2_ok", "set_lstm_wavefront", "lstm_cell", "gp_mix", "add_rowvec",
           "clip_sgd", "gemm", "PtrTable", "set_grad_ready_hook", "set_embed_grad_sink", "rows_gather_add", "KernelTimer", "set_kernel_timer"]


# ----------------------------------------------------------------------------
# small descriptors
# ----------------------------------------------------------------------------
@dataclass
class Drop:
    """A dropout site: probability + Philox key + the global column window of
    this rank (masks are keyed by global column)."""
    p: float = 0.0
    seed: int = 0
    site: int = 0
    step: int = 0
    col_offset: int = 0
    global_cols: int = 0
    keep: torch.Tensor = None  # attention only (parity mode): the dropout factors of the probabilities themselves, (global_cols * nhead, T, T)

    @property
    def on(self):
        return self.p > 0.0

    def rng(self):
        return L.rng(self.seed, L.STREAM_DROPOUT + self.site, self.step)


NO_DROP = Drop()


@dataclass
class NoiseSpec:
    """Where eps of a variational tensor comes from: an injected tensor
    (parity tests) or the Philox stream (seed, tensor id, step)."""
    eps: torch.Tensor = None
    seed: int = 0
    tensor_id: int = 0
    step: int = 0

    def rng(self):
        return L.rng(self.seed, L.STREAM_WEIGHT + self.tensor_id, self.step)


def _variational(lgstd, noise, row_lo, srows):
    v = L.Variational()
    v.lgstd = ptr(lgstd)
    if noise is not None and noise.eps is not None:
        e = noise.eps
        # an injected eps is read by the kernels with lgstd's extent on trust: a wrong shape would read past it
        if not (torch.is_tensor(e) and e.is_cuda and e.dtype == torch.float32 and e.is_contiguous() and e.numel() == lgstd.numel()):
            raise BayesLMError("injected eps must be a contiguous fp32 GPU tensor with lgstd's %d elements, got %s"
                               % (lgstd.numel(), (tuple(e.shape), e.dtype, e.device) if torch.is_tensor(e) else type(e)))
    v.eps = ptr(noise.eps) if (noise is not None and noise.eps is not None) else None
    v.row_lo = int(row_lo)
    v.srows = int(srows)
    v.rng = noise.rng() if noise is not None else L.rng(0, 0, 0)
    return v


class _GradSlab:
    """Where ``param.grad`` comes from when it is None -- the state ``zero_grad()`` (set_to_none, torch's default and what the
    reference's loop leaves behind, train.py:401) puts every parameter in at the start of every step.  A zeros_like per
    parameter is one fill launch each: 78 launches of ~5 us per step of the headline model under the reference's own loop
    (INTEGRATION level 1; 1.9 % of the step, rocprofv3 of tools/level1_probe.py).  Instead: the parameters seen in one step
    get a 256-byte aligned slot each, and from the next step on the first request of a step allocates ONE zeroed slab for
    all of them (one fill) and every request is a view into it.  A parameter asking again for a slot the current slab has
    already handed out means a new step has begun: new slab.  Dead parameters (weak references) leave the layout when a
    slab is built.  The data-parallel trainer never comes here: its gradients are views of engine.FlatBuffers."""

    def __init__(self):
        self.lock = threading.Lock()
        self.on = True
        self.dev = {}

    @staticmethod
    def eligible(param):
        return param.is_cuda and param.dtype == torch.float32

    def take(self, param):
        if not (self.on and self.eligible(param)):
            return torch.zeros_like(param, memory_format=torch.contiguous_format)
        key = id(param)
        with self.lock:
            st = self.dev.setdefault(param.device, {"slots": {}, "total": 0, "buf": None, "claimed": set()})
            slot = st["slots"].get(key)
            if slot is not None and (slot[0]() is not param or slot[2] != param.numel()):
                slot = None  # the id belongs to another tensor now
            if slot is None:  # first sight: plain zeros now, a slot from the next step on
                n = param.numel()
                st["slots"][key] = (weakref.ref(param), st["total"], n)
                st["total"] += (n + 63) // 64 * 64
                st["buf"] = None  # the slab in use does not know the new layout
                return torch.zeros_like(param, memory_format=torch.contiguous_format)
            if st["buf"] is None or key in st["claimed"]:
                off = 0
                live = {}
                for k, (ref, _, n) in st["slots"].items():
                    if ref() is not None:
                        live[k] = (ref, off, n)
                        off += (n + 63) // 64 * 64
                st["slots"], st["total"] = live, off
                st["buf"] = torch.zeros(off, device=param.device, dtype=torch.float32)
                st["claimed"] = set()
                slot = live[key]
            st["claimed"].add(key)
            return st["buf"][slot[1]:slot[1] + slot[2]].view(param.shape)


_GRAD_SLAB = _GradSlab()


def set_grad_slab(on):
    """False: every missing ``param.grad`` is its own zeros_like again (tests compare the two)."""
    _GRAD_SLAB.on = bool(on)
    with _GRAD_SLAB.lock:
        _GRAD_SLAB.dev.clear()


def _grad_buf(param):
    """param.grad (zeroed when it does not exist yet: a view of the step's gradient slab): wgrad kernels accumulate into it."""
    if param.grad is None:
        param.grad = _GRAD_SLAB.take(param)
    return param.grad


_GRAD_HOOK = None


def set_grad_ready_hook(fn):
    """fn(param) is called after a backward kernel that writes param.grad has been enqueued (the
    data-parallel reducer uses it to start bucket all-reduces while backward is still running)."""
    global _GRAD_HOOK
    _GRAD_HOOK = fn


_EMBED_SINK = None


def set_embed_grad_sink(fn):
    """Data-parallel training only.  ``fn(weight, ids)`` is asked by the embedding backward where its rows go: None =
    scatter into weight.grad as usual; or (buffer, slot_ids, n_rows, done) = scatter dy rows into the compact
    (n_rows, D) ``buffer`` at row ``slot_ids[t,b]`` and call ``done()`` once the kernel is enqueued (engine.LateRows:
    the embedding half of the tied encoder/decoder gradient is exchanged as the rows this step touched)."""
    global _EMBED_SINK
    _EMBED_SINK = fn


def rows_gather_add(dst, slot, src, n_src):
    """dst[v,:] += src[slot[v],:] where 0 <= slot[v] < n_src."""
    L.require_gfx950()
    V, D = dst.shape
    check(lib().blm_rows_gather_add(ptr(dst), ptr(dev_tensor(slot, "slot", torch.int64)), ptr(src), V, D, int(n_src),
                                    stream()), "blm_rows_gather_add")


def _notify(*params):
    if _GRAD_HOOK is not None:
        for p in params:
            if p is not None and p.is_leaf and p.requires_grad:
                _GRAD_HOOK(p)


_VP8, _I64x8 = C.c_void_p * 8, C.c_int64 * 8


def _init_multi(items):
    """``items``: (dst, src | None, src2 | None) triples of equally sized vectors -- dst = (src or 0) + (src2 or 0) -- set by
    one launch per eight (blm_init_multi): the initial states into row 0 of the state histories, b_ih + b_hh, zeroed gradient
    carries of a recurrent layer.  dst must be contiguous fp32 views on the GPU; sources of another layout / type are converted."""
    L.require_gfx950()
    for k in range(0, len(items), 8):
        part = items[k:k + 8]
        d, a, b, n = _VP8(), _VP8(), _VP8(), _I64x8()
        keep = []
        for i, (dst, s1, s2) in enumerate(part):
            if not (dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous()):
                raise BayesLMError("_init_multi: destinations are contiguous fp32 GPU tensors")
            d[i], n[i] = dst.data_ptr(), dst.numel()
            for arr, src in ((a, s1), (b, s2)):
                if src is None:
                    arr[i] = None
                    continue
                src = _f32(src, "src")
                if src.numel() != dst.numel():
                    raise BayesLMError("_init_multi: %d elements into %d" % (src.numel(), dst.numel()))
                keep.append(src)
                arr[i] = src.data_ptr()
        check(lib().blm_init_multi(len(part), d, a, b, n, stream()), "blm_init_multi")


def _wgrad_target(w):
    """-> (buffer, accumulate, value_to_return): leaf parameters accumulate in place into .grad and
    autograd gets None; a non-leaf weight (e.g. a sampled tensor) gets a fresh gradient tensor."""
    if w.is_leaf:
        return _grad_buf(w), True, None
    buf = torch.empty_like(w, memory_format=torch.contiguous_format)
    return buf, False, buf


def _f32(t, name):
    return dev_tensor(t, name, torch.float32)


def _rows2d(t, N, name):
    """-> ((M, N) view of ``t``, its row stride).  The logits of a vocabulary that is not a multiple of 4 words (wikitext-2: 33278)
    live in a buffer whose rows are padded to one (below): the kernels take the row stride, so such a tensor is used as it is; any
    other layout is made contiguous first, as `_f32` does."""
    if t.dim() >= 2 and t.shape[-1] == N and not t.is_contiguous() and t.stride(-1) == 1 and t.is_cuda and t.dtype == torch.float32:
        try:
            v = t.view(-1, N)  # fails if the leading dimensions are not uniformly strided
        except RuntimeError:
            v = None
        if v is not None and v.stride(0) >= N and v.stride(0) % 4 == 0 and v.data_ptr() % 16 == 0:
            dev_tensor(v[:1], name, torch.float32)  # the device checks of the product path
            return v, v.stride(0)
    c = _f32(t, name)
    return c.reshape(-1, N), N


def _row_chunks(M, ld):
    """Row ranges of an (M, ld) fp32 operand that each stay under 2^32 bytes.  The LDS-DMA loaders of the GEMM kernels address an
    operand with 32-bit byte offsets (include/bayeslm.h: larger operands take the guarded loaders, at roughly half the rate); the
    logits' gradient of a 256 x 128 token batch over 33,000 words is 4.3 GB, and the 288 GB of an MI355X invite such batches
    (tools/batch_size_probe.py: 410 k tokens/s at B 128, 353 k at B 256).  The decoder's backward products run chunk by chunk instead."""
    lim = ((1 << 32) - 1) // (4 * max(int(ld), 1))
    if M <= lim or lim < 256:
        return [(0, M)]
    n = -(-M // lim)
    step = (-(-M // n) + 127) // 128 * 128
    while step > lim:
        n += 1
        step = (-(-M // n) + 127) // 128 * 128
    return [(r, min(M, r + step)) for r in range(0, M, step)]


def _padded_rows(lead_shape, N, device):
    """An (..., N) fp32 tensor whose rows start 16 bytes aligned: a view of a buffer with the row stride rounded up to 4 floats."""
    M = 1
    for d in lead_shape:
        M *= int(d)
    Np = (N + 3) // 4 * 4
    return torch.empty(M, Np, device=device, dtype=torch.float32)[:, :N].view(*lead_shape, N), Np


class _P:
    """Row addresses of a contiguous tensor: ``_P(t)[i]`` == ``t[i].data_ptr()`` without building a view per time step
    (six views per LSTM step cost more host time than the 12 us kernel they feed)."""
    __slots__ = ("b", "s")

    def __init__(self, t):
        self.b = t.data_ptr()
        self.s = t.stride(0) * t.element_size()

    def __getitem__(self, i):
        return self.b + i * self.s


class ResidualLink:
    """Joins the two backward paths of a post-LN residual block  out = LN(x + drop(f(x)))  (model.py:1041-1045).
    Autograd would add the residual path's gradient (from the LayerNorm backward) and the branch's input gradient
    (the last dgrad GEMM of f) with a separate elementwise pass over (T,B,d) -- 12 launches per cfg3 step.  The
    same ``link`` object is handed to the first op of the branch (``linear`` / ``ffn`` / ``ffn_gp`` ``link=``) and to
    ``add_dropout_ln(..., link=)``.  The branch op ARMS the link in its forward with the identity of its input; the
    LayerNorm (whose forward runs after the branch, whose backward runs before it) uses the link only when exactly ONE
    op armed it for the very tensor it adds as the residual.  Then its backward parks dx here and returns NO gradient
    for x; the branch's dgrad GEMM accumulates into the parked buffer (stream order: every earlier use is enqueued)
    and returns THAT as its input gradient.  Autograd therefore sees one producer for the pair, so any further
    consumer of x (a hook, a tap on the layer input, a layer variant that reuses it) is summed in correctly whatever
    order the nodes run in; when the link is not armed, or armed twice, or for another tensor, both ops fall back to
    returning their own gradients.  Explicit per-block objects, no global matching."""
    __slots__ = ("dx", "armed", "key")

    def __init__(self):
        self.dx = None
        self.armed = 0
        self.key = None

    @staticmethod
    def _key(x):
        return (x.data_ptr(), tuple(x.shape), x.is_contiguous())

    def arm(self, x):
        """Branch op, forward: 'my backward will run and will fold its input gradient into a parked buffer for x'."""
        self.armed += 1
        self.key = self._key(x)

    def joins(self, x):
        """LayerNorm, forward: is exactly one branch op waiting for this residual's gradient?"""
        return self.armed == 1 and self.key == self._key(x)

    def take(self, like):
        """The parked residual gradient if it fits ``like`` (the branch then returns it, completed, as its dx)."""
        d, self.dx = self.dx, None
        if d is not None and d.shape == like.shape and d.is_contiguous():
            return d
        return None


# ----------------------------------------------------------------------------
# per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers)
# ----------------------------------------------------------------------------
class KernelTimer:
    def __init__(self, all_gemms=False, only=None):
        self.records = []
        self.all_gemms = all_gemms  # also bracket untagged GEMMs, keyed by layout/shape/epilogue
        self.only = None if only is None else frozenset(only)  # bracket these tags only (a bracket costs the step ~8 us)

    def bracket(self, tag):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((tag, a, b))
        return a, b

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for tag, a, b in self.records:
            agg.setdefault(tag, []).append(a.elapsed_time(b))
        return {k: {"avg_ms": sum(v) / len(v), "n": len(v)} for k, v in agg.items()}


_TIMER = None


def set_kernel_timer(t):
    global _TIMER
    _TIMER = t


# ----------------------------------------------------------------------------
# raw GEMM
# ----------------------------------------------------------------------------
def set_gemm_mode(mode):
    """"f32" (default, the parity mode), "bf16x3" or "bf16x6": OPT-IN split-bf16 arithmetic of the GEMM family's matrix
    instruction (include/bayeslm.h blm_set_gemm_mode): lower precision than the reference's fp32 (4.5e-6 against
    3.6e-7 max relative error per GEMM at K = 4096), about twice the GEMM rate.  Process-wide."""
    code = {"f32": 0, "bf16x3": 1, "bf16x6": 2}[mode]
    check(lib().blm_set_gemm_mode(code), "blm_set_gemm_mode")


def set_option(name, value):
    """Kernel-selection options of the library (include/bayeslm.h blm_set_option): "attn_hpw", "attn_short", "attn_valu",
    "lstm_gemv", "lstm_pipe", "lstm_tail", "deterministic" (see set_deterministic).  Each of the others picks between two BUILT and parity-tested forms of a kernel; the defaults are
    the measured winners (INTEGRATION.md lists them with their tests and BLM_* environment variables)."""
    check(lib().blm_set_option(name.encode(), int(value)), "blm_set_option")


def get_option(name):
    v = C.c_int(0)
    check(lib().blm_get_option(name.encode(), C.byref(v)), "blm_get_option")
    return int(v.value)


def set_deterministic(on=True):
    """Deterministic mode (blm_set_option("deterministic", 1) / BLM_DETERMINISTIC=1): every reduction of the library
    in a fixed order -- one K slice per GEMM tile (no float atomics into C), bias / GP-coefficient column sums in one row chunk,
    KL sums through block partials added by one block, the embedding gradient by one wave per vocabulary row in position order
    -- and, on this side, the two-layer LSTM stack on one stream (no layer wavefront).  Two runs from one seed then give
    bit-identical losses and parameters (tests/test_gpu_deterministic.py), also two data-parallel runs of the same world
    size; a run at another world size sums the batch columns in another order and is equal to rounding only.  Costs a few
    per cent of the step (bench.py extra_configs `deterministic`).  Process-wide."""
    set_option("deterministic", 1 if on else 0)


def is_deterministic():
    return get_option("deterministic") == 1


def set_gemm_cus(n):
    """Compute units the GEMM planner may count on (0: the whole chip).  engine.GradReducer narrows it while gradient buckets
    are in flight: the collective's channel workgroups hold CUs beside the backward GEMMs (include/bayeslm.h
    blm_gemm_plan_set_cus).  Host-side state of the planner: it affects launches enqueued AFTER the call."""
    check(lib().blm_gemm_plan_set_cus(int(n)), "blm_gemm_plan_set_cus")


def get_gemm_cus():
    return int(lib().blm_gemm_plan_get_cus())


def gemm_comm_window(us):
    """A gradient bucket expected to spend ``us`` microseconds on the links has been launched (0: the exchange is over): the
    GEMMs planned while the window is open run beside the collective's channel workgroups and take the plans measured
    there (include/bayeslm.h blm_gemm_plan_comm_window)."""
    check(lib().blm_gemm_plan_comm_window(float(us)), "blm_gemm_plan_comm_window")


def get_gemm_mode():
    return ("f32", "bf16x3", "bf16x6")[int(lib().blm_get_gemm_mode())]


def gemm(op, A, B, Cout, M, N, K, lda, ldb, ldc, *, alpha=1.0, accumulate=False, epilogue=L.EPI_NONE, bias=None,
         aux=None, coef=None, var_b=None, C2=None, wg_mu=None, var_c=None, kl_lambda=0.0, kl_inv_n=0.0,
         drop=None, drop_B=0, tag=None, colsum_a=None):
    if _TIMER is not None and (tag is not None or _TIMER.all_gemms) and (_TIMER.only is None or tag in _TIMER.only):
        if _TIMER.all_gemms:
            tag = f"{('NT', 'NN', 'TN')[op]} {M}x{N}x{K} epi{epilogue}{' acc' if accumulate else ''} [{tag or '-'}]"
        ev0, ev1 = _TIMER.bracket(tag)
        ev0.record()
        _gemm(op, A, B, Cout, M, N, K, lda, ldb, ldc, alpha, accumulate, epilogue, bias, aux, coef, var_b, C2, wg_mu,
              var_c, kl_lambda, kl_inv_n, drop, drop_B, colsum_a)
        ev1.record()
        return
    _gemm(op, A, B, Cout, M, N, K, lda, ldb, ldc, alpha, accumulate, epilogue, bias, aux, coef, var_b, C2, wg_mu,
          var_c, kl_lambda, kl_inv_n, drop, drop_B, colsum_a)


def _gemm(op, A, B, Cout, M, N, K, lda, ldb, ldc, alpha, accumulate, epilogue, bias, aux, coef, var_b, C2, wg_mu,
          var_c, kl_lambda, kl_inv_n, drop, drop_B, colsum_a=None):
    L.require_gfx950()
    a = L.GemmArgs()
    a.abi_version = L.ABI_VERSION
    a.op = op
    a.M, a.N, a.K = int(M), int(N), int(K)
    a.A, a.lda = ptr(A), int(lda)
    a.B, a.ldb = ptr(B), int(ldb)
    a.C, a.ldc = ptr(Cout), int(ldc)
    a.alpha = float(alpha)
    a.flags = L.GEMM_ACCUMULATE if accumulate else 0
    a.epilogue = epilogue
    a.bias, a.aux, a.coef = ptr(bias), ptr(aux), ptr(coef)
    if var_b is not None:
        a.var_b = var_b
    a.C2, a.wg_mu = ptr(C2), ptr(wg_mu)
    if var_c is not None:
        a.var_c = var_c
    a.kl_lambda, a.kl_inv_n = float(kl_lambda), float(kl_inv_n)
    if drop is not None and drop.on:
        a.drop_p = float(drop.p)
        a.drop_rng = drop.rng()
        a.drop_B = int(drop_B)
        a.drop_col_offset = int(drop.col_offset)
        a.drop_global_cols = int(drop.global_cols or drop_B)
    a.colsum_a = ptr(colsum_a)
    check(lib().blm_gemm(C.byref(a), stream()), "blm_gemm")


def _colsum_into(dy2, M, N, out, accumulate=True, ld=None):
    check(lib().blm_colsum(ptr(dy2), N if ld is None else int(ld), ptr(out), M, N, 1 if accumulate else 0, stream()), "blm_colsum")


# ----------------------------------------------------------------------------
# Linear:  y = x W^T + b      (model.py:876,921,975-977,1306 F.linear call sites)
# ----------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, link=None):
        x = _f32(x, "x")
        N, K = w.shape
        M = x.numel() // K
        if N % 4 and N >= 64:
            # an output width that is not a multiple of 4 (a vocabulary of 33278 words): rows of N floats are not 16-byte aligned,
            # which took this product's epilogue, the cross entropy and BOTH backward products (the logits' gradient is their
            # A operand) off their vector paths -- the headline step 19.9 -> 26.4 ms (tools/shape_cliff_probe.py).  The rows are
            # padded to a multiple of 4 floats instead; what the caller sees is the (..., N) view of that buffer.
            # The product itself runs on a copy of the weight with Np - N zero rows behind it (and a zero-padded bias): every extent
            # a multiple of 4, the padding columns of the output exactly 0.
            y, ldy = _padded_rows(x.shape[:-1], N, x.device)
            pad = torch.nn.functional.pad
            w_p = pad(w.detach(), (0, 0, 0, ldy - N))
            b_p = pad(b.detach(), (0, ldy - N)) if b is not None else None
            gemm(L.GEMM_NT, x, w_p, y, M, ldy, K, K, K, ldy, epilogue=L.EPI_BIAS if b is not None else L.EPI_NONE, bias=b_p)
            ctx.w_p = w_p if (ctx.needs_input_grad[0]) else None
        else:
            y, ldy = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32), N
            gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, ldy, epilogue=L.EPI_BIAS if b is not None else L.EPI_NONE, bias=b)
            ctx.w_p = None
        ctx.save_for_backward(x)
        ctx.w, ctx.b, ctx.link = w, b, link
        if link is not None and ctx.needs_input_grad[0]:
            link.arm(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        N, K = w.shape
        M = x.numel() // K
        dy, ldy = _rows2d(dy, N, "dy")  # the padded rows of an odd vocabulary's logits come back as they went out
        Np = (N + 3) // 4 * 4
        if (N % 4 and N >= 64 and ldy == Np and ctx.link is None
                and (dy.storage_offset() + (M - 1) * Np + Np) * 4 <= dy.untyped_storage().nbytes()):
            return _Linear._backward_padded(ctx, x, w, b, dy, M, N, Np, K)
        dx = None
        if ctx.needs_input_grad[0]:
            parked = ctx.link.take(x) if ctx.link is not None else None
            if parked is not None and parked.data_ptr() == dy.data_ptr():
                # this op feeds the LayerNorm directly and no dropout sits between them: the LayerNorm backward hands ONE
                # buffer out as dx and dy, and a GEMM cannot accumulate into its own A operand
                dx = torch.empty_like(x)
                gemm(L.GEMM_NN, dy, w, dx, M, K, N, ldy, K, K)
                dx.add_(parked)
            elif parked is not None:  # residual block: add into the LayerNorm backward's dx (ResidualLink)
                gemm(L.GEMM_NN, dy, w, parked, M, K, N, ldy, K, K, accumulate=True)
                dx = parked
            else:
                dx = torch.empty_like(x)
                dx2 = dx.view(-1, K)
                for r0, r1 in _row_chunks(M, ldy):  # one chunk unless dy is 4 GB or more
                    gemm(L.GEMM_NN, dy[r0:r1], w, dx2[r0:r1], r1 - r0, K, N, ldy, K, K)
        dw = db = None
        fuse_b = w.requires_grad and b is not None and b.requires_grad and b.is_leaf
        if w.requires_grad:
            buf, acc, dw = _wgrad_target(w)
            x2 = x.view(-1, K)
            for r0, r1 in _row_chunks(M, ldy):
                gemm(L.GEMM_TN, dy[r0:r1], x2[r0:r1], buf, N, K, r1 - r0, ldy, K, K, accumulate=acc or r0 > 0,
                     colsum_a=_grad_buf(b) if fuse_b else None)
        if b is not None and b.requires_grad and not fuse_b:
            buf, acc, db = _wgrad_target(b)
            _colsum_into(dy, M, N, buf, accumulate=acc, ld=ldy)
        _notify(w, b)
        return dx, dw, db, None

    @staticmethod
    def _backward_padded(ctx, x, w, b, dy, M, N, Np, K):
        """Backward over the padded layout of an output whose width is not a multiple of 4: dy is the (M, N) window of an
        (M, Np) buffer.  Both products take the whole buffer -- the gradient of the logits is their A operand, and its contiguous
        extent has to be a multiple of 4 for the vector / LDS-DMA loaders -- with the padding columns zeroed (the forward left
        zeros there and the cross entropy writes only real columns, but the buffer may be anybody's) against the zero-row padded
        weight; the weight gradient's padding rows are dropped when it is added to ``w.grad``."""
        full = dy.as_strided((M, Np), (Np, 1))
        full[:, N:].zero_()
        dx = None
        if ctx.needs_input_grad[0]:
            w_p = ctx.w_p if ctx.w_p is not None else torch.nn.functional.pad(w.detach(), (0, 0, 0, Np - N))
            dx = torch.empty_like(x)
            dx2 = dx.view(-1, K)
            for r0, r1 in _row_chunks(M, Np):
                gemm(L.GEMM_NN, full[r0:r1], w_p, dx2[r0:r1], r1 - r0, K, Np, Np, K, K)
        dw = db = None
        want_b = b is not None and b.requires_grad
        if w.requires_grad:
            dw_p = torch.empty(Np, K, device=x.device, dtype=torch.float32)
            db_p = torch.zeros(Np, device=x.device, dtype=torch.float32) if want_b else None
            x2 = x.view(-1, K)
            for r0, r1 in _row_chunks(M, Np):
                gemm(L.GEMM_TN, full[r0:r1], x2[r0:r1], dw_p, Np, K, r1 - r0, Np, K, K, accumulate=r0 > 0, colsum_a=db_p)
            if w.is_leaf:
                _grad_buf(w).add_(dw_p[:N])
            else:
                dw = dw_p[:N]
            if want_b:
                if b.is_leaf:
                    _grad_buf(b).add_(db_p[:N])
                else:
                    db = db_p[:N]
        elif want_b:
            buf, acc, db = _wgrad_target(b)
            _colsum_into(full, M, N, buf, accumulate=acc, ld=Np)
        _notify(w, b)
        return dx, dw, db, None


def linear(x, w, b=None, link=None):
    w = _f32(w, "weight")
    if w.dim() != 2 or x.shape[-1] != w.shape[1] or (b is not None and b.numel() != w.shape[0]):
        # the kernels take M = numel / K on trust: a mismatch would read past a buffer instead of raising like F.linear
        raise BayesLMError("linear: input (..., %d) against weight %s%s" % (x.shape[-1], tuple(w.shape),
                                                                            "" if b is None else " and bias %s" % (tuple(b.shape),)))
    return _Linear.apply(x, w, b, link)


# ----------------------------------------------------------------------------
# BayesLinear:  y = x (mu + exp(lgstd) eps)^T   (model.py:1083-1129)
# ----------------------------------------------------------------------------
def sample_weight(mu, lgstd, noise, row_lo=0, srows=None, out=None, kl_out=None, kl_weight=1.0):
    """W = mu (+ noise on rows [row_lo,row_lo+srows)); optional fused KL accumulation into kl_out."""
    L.require_gfx950()
    mu = _f32(mu, "mu")
    rows = mu.shape[0]
    cols = mu.numel() // rows
    if srows is None:
        srows = rows
    v = _variational(lgstd, noise, row_lo, srows)
    if out is None and (noise is not None or kl_out is None):
        out = torch.empty_like(mu)
    check(lib().blm_sample_weight(ptr(mu), rows, cols, C.byref(v), ptr(out), ptr(kl_out), float(kl_weight), stream()),
          "blm_sample_weight")
    return out


class _SampleWeight(torch.autograd.Function):
    """Differentiable materialisation, used where the sampled tensor feeds something other than one
    GEMM (the LSTM stack, bias vectors, the EMB projection)."""

    @staticmethod
    def forward(ctx, mu, lgstd, noise, row_lo):
        W = sample_weight(mu, lgstd, noise, row_lo, lgstd.shape[0])
        ctx.meta = (mu, lgstd, noise, row_lo)
        return W

    @staticmethod
    def backward(ctx, dW):
        mu, lgstd, noise, row_lo = ctx.meta
        dW = _f32(dW, "dW")
        rows = mu.shape[0]
        cols = mu.numel() // rows
        v = _variational(lgstd, noise, row_lo, lgstd.shape[0])
        check(lib().blm_sample_weight_bwd(ptr(dW), rows, cols, C.byref(v), ptr(_grad_buf(mu)) if mu.requires_grad else None,
                                          ptr(_grad_buf(lgstd)) if lgstd.requires_grad else None, stream()),
              "blm_sample_weight_bwd")
        _notify(mu, lgstd)
        return None, None, None, None


def sampled(mu, lgstd, noise, row_lo=0):
    """W = mu with noise on rows [row_lo, row_lo + lgstd.shape[0]); differentiable w.r.t. mu, lgstd."""
    return _SampleWeight.apply(mu, lgstd, noise, row_lo)


class _VarGroup(torch.autograd.Function):
    """Every variational tensor of a module in ONE launch per direction (blm_variational_group_fwd / _bwd): the
    samples W_i, the module's KL term as a by-product of the same pass, and in backward dmu / dlgstd of all items with
    the KL gradient folded in.  ``specs``: (mu, lgstd, noise, row_lo, kl_weight, kl_minus) per item; ``params`` repeats
    the mu / lgstd tensors so that autograd sees them."""

    @staticmethod
    def forward(ctx, specs, *params):
        L.require_gfx950()
        n = len(specs)
        items = (L.VarItem * n)()
        outs = []
        any_kl = any(sp[4] != 0.0 for sp in specs)
        kl = torch.empty((), device=specs[0][0].device, dtype=torch.float32) if any_kl else None
        for it, (mu, lg, noise, row_lo, klw, klm) in zip(items, specs):
            mu, lg = _f32(mu, "mu"), _f32(lg, "lgstd")
            W = torch.empty_like(mu)
            it.mu, it.rows = ptr(mu), mu.shape[0]
            it.cols = mu.numel() // mu.shape[0]
            it.v = _variational(lg, noise, row_lo, lg.shape[0])
            it.w_out, it.kl_weight, it.kl_minus = ptr(W), float(klw), float(klm)
            outs.append(W)
        check(lib().blm_variational_group_fwd(items, n, ptr(kl), stream()), "blm_variational_group_fwd")
        ctx.specs = specs
        if kl is None:
            kl = torch.zeros((), device=specs[0][0].device, dtype=torch.float32)
            ctx.mark_non_differentiable(kl)
        return (*outs, kl)

    @staticmethod
    def backward(ctx, *grads):
        specs = ctx.specs
        n = len(specs)
        items = (L.VarItem * n)()
        keep, touched = [], []
        for it, (mu, lg, noise, row_lo, klw, klm), dW in zip(items, specs, grads[:n]):
            if dW is not None:
                dW = _f32(dW, "dW")
                keep.append(dW)
            it.mu, it.rows = ptr(mu), mu.shape[0]
            it.cols = mu.numel() // mu.shape[0]
            it.v = _variational(lg, noise, row_lo, lg.shape[0])
            it.kl_weight, it.kl_minus = float(klw), float(klm)
            it.dw = ptr(dW)
            it.dmu = ptr(_grad_buf(mu)) if mu.requires_grad else None
            it.dlgstd = ptr(_grad_buf(lg)) if lg.requires_grad else None
            touched += [mu, lg]
        g = grads[n]
        if g is not None:
            g = _f32(g.reshape(1), "g")
        check(lib().blm_variational_group_bwd(items, n, ptr(g), stream()), "blm_variational_group_bwd")
        _notify(*touched)
        return (None,) * (1 + 2 * n)


def variational_group(specs):
    """specs: [(mu, lgstd, noise, row_lo, kl_weight, kl_minus)] -> ([W_i], kl).  W_i = mu_i with noise on the rows
    [row_lo, row_lo + lgstd.shape[0]); kl = sum_i kl_weight_i * mean_i(mu_s^2 - 2 lgstd + exp(2 lgstd) - kl_minus) / 2
    over the noisy rows (0 when no item carries a weight).  One launch forward, one backward."""
    if len(specs) > L.VAR_GROUP_MAX or any(sp[0].numel() >= 2 ** 31 or sp[2] is None for sp in specs):
        raise BayesLMError("variational_group: at most %d sampled tensors of < 2^31 elements" % L.VAR_GROUP_MAX)
    flat = []
    for sp in specs:
        flat += [sp[0], sp[1]]
    out = _VarGroup.apply(specs, *flat)
    return list(out[:-1]), out[-1]


class _BayesLinear(torch.autograd.Function):
    """``fused=True``: eps is generated inside the GEMM tile loader (no W in HBM) in forward and
    dgrad; ``fused=False``: one materialisation pass writes W, then plain GEMMs.  The wgrad GEMM's
    epilogue turns dW into (dmu, dlgstd) with eps regenerated from the counter and adds the KL
    gradient scaled by kl_lambda."""

    @staticmethod
    def forward(ctx, x, mu, lgstd, noise, kl_lambda, fused):
        x = _f32(x, "x")
        N, K = mu.shape
        M = x.numel() // K
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        W = None
        if noise is None:
            gemm(L.GEMM_NT, x, mu, y, M, N, K, K, K, N)
        elif fused:
            gemm(L.GEMM_NT, x, mu, y, M, N, K, K, K, N, var_b=_variational(lgstd, noise, 0, N))
        else:
            W = sample_weight(mu, lgstd, noise)
            gemm(L.GEMM_NT, x, W, y, M, N, K, K, K, N)
        ctx.save_for_backward(x, W)
        ctx.mu, ctx.lgstd, ctx.noise, ctx.kl_lambda, ctx.fused = mu, lgstd, noise, kl_lambda, fused
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        mu, lgstd, noise = ctx.mu, ctx.lgstd, ctx.noise
        dy = _f32(dy, "dy")
        N, K = mu.shape
        M = x.numel() // K
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if noise is None:
                gemm(L.GEMM_NN, dy, mu, dx, M, K, N, N, K, K)
            elif ctx.fused:
                gemm(L.GEMM_NN, dy, mu, dx, M, K, N, N, K, K, var_b=_variational(lgstd, noise, 0, N))
            else:
                gemm(L.GEMM_NN, dy, W, dx, M, K, N, N, K, K)
        if mu.requires_grad:
            if noise is None:  # eval-mode graph: only the mean gets a gradient
                gemm(L.GEMM_TN, dy, x, _grad_buf(mu), N, K, M, N, K, K, accumulate=True)
            else:
                gemm(L.GEMM_TN, dy, x, _grad_buf(mu), N, K, M, N, K, K, accumulate=True,
                     epilogue=L.EPI_BAYES_WGRAD, C2=_grad_buf(lgstd), wg_mu=mu,
                     var_c=_variational(lgstd, noise, 0, N), kl_lambda=ctx.kl_lambda, kl_inv_n=1.0 / (N * K))
                _notify(lgstd)
            _notify(mu)
        return dx, None, None, None, None, None


def bayes_linear(x, mu, lgstd, noise=None, kl_lambda=0.0, fused=False):
    return _BayesLinear.apply(x, mu, lgstd, noise, kl_lambda, fused)


# ----------------------------------------------------------------------------
# FFN:  y = lin2(drop(gelu(x W1^T + b1)))        (model.py:1043, 1169)
# lin2 is nn.Linear (w2, b2) or BayesLinear (mu2, lgstd2, no bias).
# ----------------------------------------------------------------------------
class _FFN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, lgstd2, noise, kl_lambda, fused, drop, link=None):
        ctx.link = link
        x = _f32(x, "x")
        if link is not None and ctx.needs_input_grad[0]:
            link.arm(x)
        F_, D = w1.shape  # (ff, d)
        N2 = w2.shape[0]
        M = x.numel() // D
        B = x.shape[-2] if x.dim() >= 2 else 1
        need_bwd = any(ctx.needs_input_grad)  # forward itself runs with grad mode off
        z = torch.empty(M, F_, device=x.device, dtype=torch.float32) if need_bwd else None
        h = torch.empty(M, F_, device=x.device, dtype=torch.float32)
        gemm(L.GEMM_NT, x, w1, h, M, F_, D, D, D, F_, epilogue=L.EPI_BIAS_GELU, bias=b1, aux=z, drop=drop, drop_B=B,
             tag="ffn_linear1_fwd")
        y = torch.empty(*x.shape[:-1], N2, device=x.device, dtype=torch.float32)
        W = None
        bayes = lgstd2 is not None
        if bayes and noise is not None:
            if fused:
                gemm(L.GEMM_NT, h, w2, y, M, N2, F_, F_, F_, N2, var_b=_variational(lgstd2, noise, 0, N2),
                     tag="sampled_gemm_fwd")
            else:
                W = sample_weight(w2, lgstd2, noise)
                gemm(L.GEMM_NT, h, W, y, M, N2, F_, F_, F_, N2, tag="sampled_gemm_fwd")
        else:
            gemm(L.GEMM_NT, h, w2, y, M, N2, F_, F_, F_, N2,
                 epilogue=L.EPI_BIAS if b2 is not None else L.EPI_NONE, bias=b2)
        ctx.save_for_backward(x, z, h, W)
        ctx.p = (w1, b1, w2, b2, lgstd2, noise, kl_lambda, fused, drop, B)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, h, W = ctx.saved_tensors
        w1, b1, w2, b2, lgstd2, noise, kl_lambda, fused, drop, B = ctx.p
        dy = _f32(dy, "dy")
        F_, D = w1.shape
        N2 = w2.shape[0]
        M = x.numel() // D
        bayes = lgstd2 is not None and noise is not None
        # dz = (dy W2) * [gelu'(z) * keep]: the bracket was written by the forward epilogue (aux)
        dz = torch.empty(M, F_, device=x.device, dtype=torch.float32)
        if bayes and fused:
            gemm(L.GEMM_NN, dy, w2, dz, M, F_, N2, N2, F_, F_, epilogue=L.EPI_MUL_DGELU, aux=z,
                 var_b=_variational(lgstd2, noise, 0, N2), tag="sampled_gemm_dgrad")
        else:
            gemm(L.GEMM_NN, dy, W if bayes else w2, dz, M, F_, N2, N2, F_, F_, epilogue=L.EPI_MUL_DGELU, aux=z,
                 tag="sampled_gemm_dgrad" if bayes else None)
        # linear2 weight gradients
        if w2.requires_grad:
            if bayes:
                gemm(L.GEMM_TN, dy, h, _grad_buf(w2), N2, F_, M, N2, F_, F_, accumulate=True,
                     epilogue=L.EPI_BAYES_WGRAD, C2=_grad_buf(lgstd2), wg_mu=w2,
                     var_c=_variational(lgstd2, noise, 0, N2), kl_lambda=kl_lambda, kl_inv_n=1.0 / (N2 * F_),
                     tag="sampled_gemm_wgrad")
            else:
                gemm(L.GEMM_TN, dy, h, _grad_buf(w2), N2, F_, M, N2, F_, F_, accumulate=True,
                     colsum_a=_grad_buf(b2) if (b2 is not None and b2.requires_grad) else None)
        elif b2 is not None and b2.requires_grad:
            _colsum_into(dy, M, N2, _grad_buf(b2))
        _notify(w2, b2, lgstd2 if bayes else None)
        # linear1 (bias gradient = column sums of dz, taken inside the wgrad GEMM)
        if w1.requires_grad:
            gemm(L.GEMM_TN, dz, x, _grad_buf(w1), F_, D, M, F_, D, D, accumulate=True,
                 colsum_a=_grad_buf(b1) if b1.requires_grad else None)
        elif b1.requires_grad:
            _colsum_into(dz, M, F_, _grad_buf(b1))
        _notify(w1, b1)
        dx = None
        if ctx.needs_input_grad[0]:
            parked = ctx.link.take(x) if ctx.link is not None else None
            if parked is not None:
                gemm(L.GEMM_NN, dz, w1, parked, M, D, F_, F_, D, D, accumulate=True)
                dx = parked
            else:
                dx = torch.empty_like(x)
                gemm(L.GEMM_NN, dz, w1, dx, M, D, F_, F_, D, D)
        return (dx,) + (None,) * 10


def ffn(x, w1, b1, w2, b2=None, lgstd2=None, noise=None, kl_lambda=0.0, fused=False, drop=NO_DROP, link=None):
    return _FFN.apply(x, w1, b1, w2, b2, lgstd2, noise, kl_lambda, fused, drop, link)


class _FFNGP(torch.autograd.Function):
    """y = lin2(drop(sum_i act_i(x Wg^T + bg) coef[i]))   GaussTransformerEncoderLayer FFN
    (model.py:2283): GPNN replaces GELU(linear1(x)); acts = tanh, sigmoid, relu, gelu (model.py:2263)."""

    @staticmethod
    def forward(ctx, x, wg, bg, coef, w2, b2, drop, link=None):
        ctx.link = link
        x = _f32(x, "x")
        if link is not None and ctx.needs_input_grad[0]:
            link.arm(x)
        F_, D = wg.shape
        N2 = w2.shape[0]
        M = x.numel() // D
        B = x.shape[-2]
        z = torch.empty(M, F_, device=x.device, dtype=torch.float32) if any(ctx.needs_input_grad) else None
        h = torch.empty(M, F_, device=x.device, dtype=torch.float32)
        gemm(L.GEMM_NT, x, wg, h, M, F_, D, D, D, F_, epilogue=L.EPI_GP_MIX, bias=bg, aux=z, coef=coef, drop=drop, drop_B=B)
        y = torch.empty(*x.shape[:-1], N2, device=x.device, dtype=torch.float32)
        gemm(L.GEMM_NT, h, w2, y, M, N2, F_, F_, F_, N2, epilogue=L.EPI_BIAS, bias=b2)
        ctx.save_for_backward(x, z, h)
        ctx.p = (wg, bg, coef, w2, b2, drop, B)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, h = ctx.saved_tensors
        wg, bg, coef, w2, b2, drop, B = ctx.p
        dy = _f32(dy, "dy")
        F_, D = wg.shape
        N2 = w2.shape[0]
        M = x.numel() // D
        dz = torch.empty(M, F_, device=x.device, dtype=torch.float32)
        dhk = torch.empty(M, F_, device=x.device, dtype=torch.float32) if coef.requires_grad else None
        gemm(L.GEMM_NN, dy, w2, dz, M, F_, N2, N2, F_, F_, epilogue=L.EPI_MUL_DGP_MIX, aux=z, coef=coef, C2=dhk,
             drop=drop, drop_B=B)
        # wg / bg / coef are the leaf mean tensors, or -- GPNN.sample raised -- sampled (non-leaf) ones whose gradients go
        # back to the sampling node (ops.variational_group: d mean, d lgstd = dW eps sigma in one launch)
        dwg = dbg = dcoef = None
        if coef.requires_grad:
            buf, acc, dcoef = _wgrad_target(coef)
            if not acc:
                buf.zero_()
            check(lib().blm_gp_coef_grad(ptr(dhk), ptr(z), ptr(buf), M, F_, stream()), "blm_gp_coef_grad")
        if w2.requires_grad:
            gemm(L.GEMM_TN, dy, h, _grad_buf(w2), N2, F_, M, N2, F_, F_, accumulate=True)
        if b2.requires_grad:
            _colsum_into(dy, M, N2, _grad_buf(b2))
        if wg.requires_grad:
            buf, acc, dwg = _wgrad_target(wg)
            gemm(L.GEMM_TN, dz, x, buf, F_, D, M, F_, D, D, accumulate=acc)
        if bg.requires_grad:
            buf, acc, dbg = _wgrad_target(bg)
            _colsum_into(dz, M, F_, buf, accumulate=acc)
        _notify(coef, w2, b2, wg, bg)
        dx = None
        if ctx.needs_input_grad[0]:
            parked = ctx.link.take(x) if ctx.link is not None else None
            if parked is not None:
                gemm(L.GEMM_NN, dz, wg, parked, M, D, F_, F_, D, D, accumulate=True)
                dx = parked
            else:
                dx = torch.empty_like(x)
                gemm(L.GEMM_NN, dz, wg, dx, M, D, F_, F_, D, D)
        return dx, dwg, dbg, dcoef, None, None, None, None


def ffn_gp(x, wg, bg, coef, w2, b2, drop=NO_DROP, link=None):
    return _FFNGP.apply(x, wg, bg, coef, w2, b2, drop, link)


# ----------------------------------------------------------------------------
# causal self-attention core on packed or separate q/k/v   (model.py:889-920)
# ----------------------------------------------------------------------------
class _Attention(torch.autograd.Function):
    """Packed: qkv (T,B,3d) = [q|k|v] from one qkv_net GEMM (model.py:876); separate: three (T,B,d)
    tensors (BayesMultiheadAttention, model.py:975-977).  One input / one gradient tensor in the
    packed case, so autograd never splits or re-concatenates the 3d-wide activation."""

    @staticmethod
    def forward(ctx, a, k, v, nhead, drop):
        a = _f32(a, "qkv")
        packed = k is None
        if packed:
            T, B, d3 = a.shape
            d = d3 // 3
            q, kk, vv, ld = a, a[..., d:], a[..., 2 * d:], d3
        else:
            T, B, d = a.shape
            q, kk, vv, ld = a, _f32(k, "k"), _f32(v, "v"), d
        hd = d // nhead
        out = torch.empty(T, B, d, device=a.device, dtype=torch.float32)
        lse = torch.empty(B * nhead, T, device=a.device, dtype=torch.float32)
        L.require_gfx950()
        r = drop.rng() if drop.on else None
        if drop.keep is not None:  # the mask itself (torch's CPU dropout drew it): the vector-ALU kernels read it
            keep = _f32(drop.keep, "keep")
            if keep.numel() != (drop.global_cols or B) * nhead * T * T or not keep.is_contiguous():
                raise BayesLMError("attention: keep mask must be (global_cols * nhead, T, T) contiguous")
            check(lib().blm_attn_fwd_keep(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), ld, ptr(out), ptr(lse), T, B, nhead, hd,
                                          ptr(keep), drop.col_offset, drop.global_cols or B, stream()), "blm_attn_fwd_keep")
        else:
            check(lib().blm_attn_fwd(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), ld, ptr(out), ptr(lse), T, B, nhead, hd,
                                     float(drop.p), C.byref(r) if r is not None else None, drop.col_offset,
                                     drop.global_cols or B, stream()), "blm_attn_fwd")
        ctx.save_for_backward(q, kk if not packed else None, vv if not packed else None, out, lse)
        ctx.meta = (nhead, drop, packed, d)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, k, v, out, lse = ctx.saved_tensors
        nhead, drop, packed, d = ctx.meta
        T, B = a.shape[0], a.shape[1]
        dout = _f32(dout, "dout")
        if packed:
            q, kk, vv, ld = a, a[..., d:], a[..., 2 * d:], 3 * d
            dqkv = torch.empty(T, B, 3 * d, device=a.device, dtype=torch.float32)
            dq, dk, dv, ldd = dqkv, dqkv[..., d:], dqkv[..., 2 * d:], 3 * d
        else:
            q, kk, vv, ld = a, k, v, d
            dq, dk, dv = (torch.empty(T, B, d, device=a.device, dtype=torch.float32) for _ in range(3))
            ldd = d
        if drop.keep is not None:
            check(lib().blm_attn_bwd_keep(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), ld, ptr(out), ptr(dout), ptr(lse),
                                          dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ldd, T, B, nhead, d // nhead,
                                          ptr(drop.keep), drop.col_offset, drop.global_cols or B, stream()), "blm_attn_bwd_keep")
            return (dqkv, None, None, None, None) if packed else (dq, dk, dv, None, None)
        r = drop.rng() if drop.on else None
        # scratch for dS (B*nhead, T, T): the dK/dV kernel leaves it there, dQ = dS K needs no second recomputation
        nws = int(lib().blm_attn_bwd_ws_floats(T, B, nhead, d // nhead)) if _ATTN_WS else 0
        ws = torch.empty(nws, device=a.device, dtype=torch.float32) if nws > 0 else None
        check(lib().blm_attn_bwd_ws(q.data_ptr(), kk.data_ptr(), vv.data_ptr(), ld, ptr(out), ptr(dout), ptr(lse),
                                    dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ldd, T, B, nhead, d // nhead,
                                    float(drop.p), C.byref(r) if r is not None else None, drop.col_offset,
                                    drop.global_cols or B, ptr(ws), nws, stream()), "blm_attn_bwd_ws")
        if packed:
            return dqkv, None, None, None, None
        return dq, dk, dv, None, None


_ATTN_WS = os.environ.get("BLM_ATTN_WS", "1") != "0"  # 0: the two-recomputation backward (A/B measurements)


def attention(qkv, nhead, drop=NO_DROP):
    """Causal self-attention core on packed (T,B,3d) projections."""
    return _Attention.apply(qkv, None, None, nhead, drop)


def attention_qkv(q, k, v, nhead, drop=NO_DROP):
    """Same, on three separate (T,B,d) projections."""
    return _Attention.apply(q, k, v, nhead, drop)


# ----------------------------------------------------------------------------
# out = LayerNorm(x + drop(y))      (model.py:1041-1042, 1044-1045)
# ----------------------------------------------------------------------------
class _AddDropLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps, drop, link=None):
        x, y = _f32(x, "x"), _f32(y, "y")
        ctx.link = link if (link is not None and link.joins(x)) else None
        D = x.shape[-1]
        B = x.shape[-2]
        rows = x.numel() // (B * D)
        out = torch.empty_like(x)
        s = torch.empty_like(x)
        mean = torch.empty(rows * B, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        L.require_gfx950()
        r = drop.rng() if drop.on else None
        check(lib().blm_add_dropout_ln_fwd(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(out), ptr(s), ptr(mean), ptr(rstd),
                                           rows, B, D, float(eps), float(drop.p), C.byref(r) if r is not None else None,
                                           drop.col_offset, drop.global_cols or B, stream()), "blm_add_dropout_ln_fwd")
        ctx.save_for_backward(s, mean, rstd)
        ctx.meta = (gamma, beta, drop, rows, B, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        s, mean, rstd = ctx.saved_tensors
        gamma, beta, drop, rows, B, D = ctx.meta
        dout = _f32(dout, "dout")
        dx = torch.empty_like(s)
        dy = torch.empty_like(s) if drop.on else None
        ws = torch.empty(int(lib().blm_ln_bwd_ws_floats(rows * B, D)), device=s.device, dtype=torch.float32)
        r = drop.rng() if drop.on else None
        # frozen LayerNorm parameters (architect step): their sums land in a scratch row instead of .grad
        dgamma = _grad_buf(gamma) if gamma.requires_grad else torch.empty_like(gamma)
        dbeta = _grad_buf(beta) if beta.requires_grad else torch.empty_like(beta)
        check(lib().blm_add_dropout_ln_bwd(ptr(dout), ptr(s), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dy),
                                           ptr(dgamma), ptr(dbeta), ptr(ws), rows, B, D,
                                           float(drop.p), C.byref(r) if r is not None else None, drop.col_offset,
                                           drop.global_cols or B, stream()), "blm_add_dropout_ln_bwd")
        _notify(gamma, beta)
        dres = dx
        if ctx.link is not None and ctx.needs_input_grad[0]:
            ctx.link.dx = dx  # the branch's first op completes this buffer and returns it as ITS dx (ResidualLink)
            dres = None
        return dres, (dy if dy is not None else dx), None, None, None, None, None


def add_dropout_ln(x, y, gamma, beta, eps=1e-5, drop=NO_DROP, link=None):
    return _AddDropLN.apply(x, y, gamma, beta, eps, drop, link)


# ----------------------------------------------------------------------------
# embedding (+ sqrt(d) scale + positional table + dropout)   (model.py:1284,116-117,218)
# ----------------------------------------------------------------------------
class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, pe, scale, drop):
        ids = dev_tensor(ids, "ids", torch.int64)
        T, B = ids.shape
        V, D = weight.shape
        if pe is not None and pe.shape[0] < T:
            raise BayesLMError("sequence length %d exceeds the positional table (%d)" % (T, pe.shape[0]))
        out = torch.empty(T, B, D, device=weight.device, dtype=torch.float32)
        L.require_gfx950()
        r = drop.rng() if drop.on else None
        check(lib().blm_embed_fwd(ptr(ids), ptr(weight), ptr(pe), ptr(out), T, B, D, V, float(scale), float(drop.p),
                                  C.byref(r) if r is not None else None, drop.col_offset, drop.global_cols or B,
                                  stream()), "blm_embed_fwd")
        ctx.save_for_backward(ids)
        ctx.meta = (weight, scale, drop)
        return out

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        weight, scale, drop = ctx.meta
        if weight.requires_grad:
            dy = _f32(dy, "dy")
            T, B = ids.shape
            V, D = weight.shape
            r = drop.rng() if drop.on else None
            sink = _EMBED_SINK(weight, ids) if (_EMBED_SINK is not None and weight.is_leaf) else None
            if sink is not None:
                buf, slots, nrows, done = sink
                check(lib().blm_embed_bwd(ptr(slots), ptr(dy), ptr(buf), T, B, D, int(nrows), float(scale),
                                          float(drop.p), C.byref(r) if r is not None else None, drop.col_offset,
                                          drop.global_cols or B, stream()), "blm_embed_bwd")
                done()
                return None, None, None, None, None
            check(lib().blm_embed_bwd(ptr(ids), ptr(dy), ptr(_grad_buf(weight)), T, B, D, V, float(scale),
                                      float(drop.p), C.byref(r) if r is not None else None, drop.col_offset,
                                      drop.global_cols or B, stream()), "blm_embed_bwd")
            _notify(weight)
        return None, None, None, None, None


def embed(ids, weight, pe=None, scale=1.0, drop=NO_DROP):
    return _Embed.apply(ids, weight, pe, scale, drop)


class _AddPE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe, drop):
        x = _f32(x, "x")
        T, B, D = x.shape
        if pe.shape[0] < T:
            raise BayesLMError("sequence length %d exceeds the positional table (%d)" % (T, pe.shape[0]))
        out = torch.empty_like(x)
        L.require_gfx950()
        r = drop.rng() if drop.on else None
        check(lib().blm_add_pe_dropout(ptr(x), ptr(pe), ptr(out), T, B, D, float(drop.p),
                                       C.byref(r) if r is not None else None, drop.col_offset, drop.global_cols or B,
                                       stream()), "blm_add_pe_dropout")
        ctx.drop = drop
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy, "dy")
        return (_dropout_apply(dy, ctx.drop) if ctx.drop.on else dy), None, None


def add_pe(x, pe, drop=NO_DROP):
    """drop(x + pe[:T]) with pe a (max_len, D) table (model.py:116-117)."""
    return _AddPE.apply(x, pe, drop)


# ----------------------------------------------------------------------------
# dropout on a (rows, B, D) activation  (model.py:220 etc.)
# ----------------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop):
        x = _f32(x, "x")
        ctx.drop = drop
        return _dropout_apply(x, drop)

    @staticmethod
    def backward(ctx, dy):
        return _dropout_apply(_f32(dy, "dy"), ctx.drop), None


def _dropout_apply(x, drop, row0=0, out=None):
    """``row0``: x is the block of rows [row0, row0 + rows) of a longer (T, B, D) tensor and gets THAT block's mask."""
    D = x.shape[-1]
    B = x.shape[-2] if x.dim() >= 2 else 1
    if x.numel() == 0:
        raise BayesLMError("dropout: empty tensor %s" % (tuple(x.shape),))
    rows = x.numel() // (B * D)
    y = torch.empty_like(x) if out is None else out
    L.require_gfx950()
    r = drop.rng()
    check(lib().blm_dropout_rows(ptr(x), ptr(y), rows, int(row0), B, D, float(drop.p), C.byref(r), drop.col_offset,
                                 drop.global_cols or B, stream()), "blm_dropout_rows")
    return y


def dropout(x, drop):
    if not drop.on:
        return x
    return _Dropout.apply(x, drop)


# ----------------------------------------------------------------------------
# cross entropy (mean over M tokens) on materialised logits  (train.py:233,332)
# ----------------------------------------------------------------------------
class _CrossEntropy(torch.autograd.Function):
    """unit_grad=True (the trainer's case, loss = CE + KL with coefficient 1, train.py:412): the
    forward pass over the logits also overwrites them in place with d(mean NLL)/d(logits), and
    backward returns that buffer untouched.  unit_grad=False: forward keeps lse, backward runs the
    gradient kernel with the real upstream scalar.  Either way the logits buffer is consumed -- unless
    ``keep`` (the `Logits` short cut of an unchanged torch loop): the logits stay the caller's, are saved through
    autograd (an in-place edit between the loss and backward() raises, as torch's own loss would), the gradient gets
    its own buffer, and the mean is torch's: over the rows whose target is not ignore_index (-100), which get neither
    loss nor gradient (the kernels give any row without a target in [0, V) a zero NLL and a zero gradient row)."""

    @staticmethod
    def forward(ctx, logits, targets, unit_grad, keep=False):
        V = logits.shape[-1]
        whole = logits                      # what autograd gets back as the gradient in the fused form: the caller's tensor ...
        logits, ld = _rows2d(logits, V, "logits")  # rows padded to 4 floats (odd vocabulary, ops._Linear) are taken as they are
        if whole.data_ptr() != logits.data_ptr():
            whole = logits.view(whole.shape)  # ... or the contiguous copy that had to be made of it, in its shape
        targets = dev_tensor(targets, "targets", torch.int64)
        M = logits.numel() // V
        if targets.numel() != M:
            raise BayesLMError("cross_entropy: %d targets for %d rows" % (targets.numel(), M))
        if M == 0:
            raise BayesLMError("cross_entropy: no rows (the mean over zero tokens is undefined; torch returns nan here)")
        nll = torch.empty(M, device=logits.device, dtype=torch.float32)
        loss = torch.zeros((), device=logits.device, dtype=torch.float32)
        grad_mode = ctx.needs_input_grad[0]  # forward itself runs with grad mode off
        fuse = grad_mode and unit_grad and not keep
        lse = torch.empty(M, device=logits.device, dtype=torch.float32) if (grad_mode and not fuse) else None
        L.require_gfx950()
        check(lib().blm_ce_fwd_bwd(ptr(logits), ld, ptr(targets), ptr(nll), ptr(lse), ptr(loss),
                                   ptr(logits) if fuse else None, 1.0 / M, M, V, stream()), "blm_ce_fwd_bwd")
        if fuse:  # the buffer now holds the gradient: any other autograd consumer of the logits must fail, not read it
            torch.autograd.graph.increment_version(whole)
        inv_count = None
        if keep:  # torch's mean: over the targets that are not ignore_index (device-side count, no host synchronisation)
            inv_count = 1.0 / (targets != -100).sum().clamp_(min=1).to(torch.float32)
            ctx.save_for_backward(whole)  # version-checked at backward
            ctx.meta = (None, targets, lse, fuse, M, V, keep, inv_count, ld)
        else:
            ctx.meta = (whole, targets, lse, fuse, M, V, keep, None, ld)
        ctx.mark_non_differentiable(nll)
        ctx.set_materialize_grads(False)  # no zero fill for the per-token NLL's (non-existent) gradient
        return (loss * inv_count if keep else loss / M), nll

    @staticmethod
    def backward(ctx, g, _g_nll):
        logits, targets, lse, fuse, M, V, keep, inv_count, ld = ctx.meta
        if g is None:
            return None, None, None, None
        if fuse:
            return logits, None, None, None
        g = _f32(g.reshape(1), "g")
        if keep:
            (logits,) = ctx.saved_tensors
            g = g * (inv_count * M)  # the kernel scales by 1 / M
        # keep: the caller's logits stay what they are (a user loop may still read them after backward): the gradient gets its own
        # buffer, rows strided as the logits' (one stride serves both in the kernel)
        if not keep:
            out = logits
        elif ld == V:
            out = torch.empty_like(logits)
        else:
            out = torch.empty(M, ld, device=logits.device, dtype=torch.float32)[:, :V].view(logits.shape)
        check(lib().blm_ce_bwd(ptr(logits), ld, ptr(targets), ptr(lse), ptr(g), 1.0 / M, ptr(out), M, V, stream()),
              "blm_ce_bwd")
        if not keep:
            torch.autograd.graph.increment_version(logits)
        return out, None, None, None


def cross_entropy(logits, targets, unit_grad=False):
    """-> (mean NLL, per-token NLL).  In grad mode the logits buffer is CONSUMED: it is overwritten with the gradient
    (its version counter is bumped, so another autograd use of the same tensor raises instead of reading gradients;
    read anything else you need from the logits BEFORE calling this).  ``unit_grad=True`` is the trainers' contract
    only: the loss enters the objective with coefficient exactly 1 (train.py:412) and the upstream gradient is not
    looked at.  The mean is over ALL M rows (the trainers' targets are corpus words, train.py:299-304); a row whose target
    is outside [0, V) contributes no loss and no gradient but still counts in M -- torch's ``ignore_index`` mean (over the
    other rows only) is what ``F.cross_entropy`` on the models' `Logits` gives."""
    if type(logits) is not torch.Tensor:
        logits = logits.as_subclass(torch.Tensor)  # model outputs are `Logits` in grad mode (below); the op wants the plain tensor
    return _CrossEntropy.apply(logits, targets, unit_grad)


class Logits(torch.Tensor):
    """What a language model's decoder returns in grad mode: a plain fp32 tensor to every consumer, except that
    ``torch.nn.functional.cross_entropy`` on it (the reference loop's ``criterion(output.view(-1, ntokens), targets)``,
    train.py:332 with ``nn.CrossEntropyLoss()``) runs the engine's one-pass cross-entropy kernels instead of torch's
    log-softmax + NLL chain over the (M, V) logits -- an unchanged reference training script gets them by importing the shim
    (INTEGRATION.md level 1).  Non-destructive here: the logits keep their values, the gradient gets its own buffer.  Only the
    default loss takes the short cut (mean reduction, no class weights, no label smoothing, ignore_index at its default -100, which is
    honoured as torch honours it: such rows get no loss and no gradient, the mean is over the others); anything else falls through to
    torch.  ``view`` / ``reshape`` / ``contiguous`` /
    ``flatten`` keep the type (so the reshaped logits still reach the loss as ``Logits``); every other operation returns plain
    tensors."""

    _KEEP = None  # filled below: the shape-only ops that preserve the type

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.cross_entropy:
            inp = args[0] if len(args) > 0 else kwargs.get("input")
            tgt = args[1] if len(args) > 1 else kwargs.get("target")
            plain = (kwargs.get("weight") is None and kwargs.get("reduction", "mean") == "mean" and kwargs.get("label_smoothing", 0.0) == 0.0
                     and kwargs.get("ignore_index", -100) == -100 and kwargs.get("size_average") is None and kwargs.get("reduce") is None
                     and len(args) <= 2 and torch.is_tensor(inp) and torch.is_tensor(tgt) and inp.dim() == 2 and tgt.dim() == 1
                     and inp.is_cuda and inp.dtype == torch.float32 and tgt.dtype == torch.int64 and tgt.numel() == inp.shape[0])
            if plain:
                with torch._C.DisableTorchFunctionSubclass():
                    return _CrossEntropy.apply(inp.as_subclass(torch.Tensor), tgt, False, True)[0]
        if func in cls._KEEP:
            return super().__torch_function__(func, types, args, kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


Logits._KEEP = frozenset([torch.Tensor.view, torch.Tensor.reshape, torch.Tensor.contiguous, torch.Tensor.flatten, torch.flatten,
                          torch.reshape, torch.Tensor.view_as, torch.Tensor.reshape_as])


def as_logits(t):
    """Decoder output of a language model in grad mode (see Logits)."""
    return t.as_subclass(Logits) if (torch.is_grad_enabled() and type(t) is torch.Tensor) else t


def cross_entropy_interp(logits_a, logits_b, alpha, targets):
    """Scoring with two models: per-token NLL of the interpolated LOGITS alpha*a + (1-alpha)*b
    (compute_sentence_scores_bayes_jianwei.py:157-168) without storing the mixture.  Forward only.
    -> (mean NLL, per-token NLL)"""
    a, b = _f32(logits_a, "logits_a"), _f32(logits_b, "logits_b")
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError("cross_entropy_interp: logits must be two (M, V) matrices of the same shape")
    L.require_gfx950()
    M, V = a.shape
    tgt = targets.contiguous()
    nll = torch.empty(M, device=a.device, dtype=torch.float32)
    check(lib().blm_ce_interp_fwd(ptr(a), ptr(b), V, float(alpha), ptr(tgt), ptr(nll), M, V, stream()), "blm_ce_interp_fwd")
    return nll.mean(), nll


def linear_nll_supported(weight, bias):
    """Decoders blm_linear_nll takes: fp32 on the GPU, row-major, vocabulary a multiple of 4, 16-byte aligned bias."""
    return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
            and weight.shape[0] % 4 == 0 and (bias is None or (bias.is_contiguous() and bias.data_ptr() % 16 == 0)))


def linear_nll(x, weight, bias, targets):
    """Inference only: per-row NLL of the decoder ``x @ weight.T + bias`` against ``targets`` without materialising the (M, V)
    logits (blm_linear_nll: softmax partials per column tile in the GEMM epilogue + a folding kernel).  What
    decoder -> log_softmax -> gather computes in train.py:452-455 / compute_sentence_scores...py:157-170.  -> (M,) NLL"""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise BayesLMError("linear_nll is an inference-only path (no backward): call it under torch.no_grad()")
    x2 = _f32(x, "x").reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M, K = x2.shape
    V = weight.shape[0]
    if weight.shape[1] != K or targets.numel() != M:
        raise ValueError("linear_nll: x (M, K), weight (V, K) and M targets expected")
    L.require_gfx950()
    tgt = targets.reshape(-1).contiguous()
    nll = torch.empty(M, device=x2.device, dtype=torch.float32)
    ws = torch.empty(int(lib().blm_linear_nll_ws_floats(M, V)), device=x2.device, dtype=torch.float32)
    check(lib().blm_linear_nll(ptr(x2), x2.stride(0), ptr(weight), weight.stride(0), ptr(bias), ptr(tgt), ptr(nll), None, ptr(ws),
                               M, V, K, stream()), "blm_linear_nll")
    return nll


class InterpDecoder:
    """The packed operands of a two-model scoring run (blm_linear_nll2): [W1 | W2] and alpha b1 + (1 - alpha) b2 are
    built by the first call and kept for the following ones (same weights, same alpha: one scoring run)."""

    def __init__(self, w1, b1, w2, b2, alpha):
        self.w1, self.b1, self.w2, self.b2, self.alpha = _f32(w1, "w1"), b1, _f32(w2, "w2"), b2, float(alpha)
        if self.w1.shape[0] != self.w2.shape[0]:
            raise BayesLMError("interpolated decoders need one vocabulary: %d and %d rows" % (self.w1.shape[0], self.w2.shape[0]))
        V, K1, K2 = self.w1.shape[0], self.w1.shape[1], self.w2.shape[1]
        self.wcat = torch.empty(int(lib().blm_linear_nll2_wcat_floats(V, K1, K2)), device=self.w1.device, dtype=torch.float32)
        self.packed = False


def linear_nll_interp_supported(w1, b1, w2, b2):
    """fp32 row-major decoders over ONE vocabulary (any size: the packed copy is padded) with feature counts that are multiples of 4."""
    ok = lambda w, b: (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous() and w.shape[1] % 4 == 0  # noqa: E731
                       and w.data_ptr() % 16 == 0 and (b is None or (b.is_contiguous() and b.numel() == w.shape[0])))
    return ok(w1, b1) and ok(w2, b2) and w1.shape[0] == w2.shape[0]


def linear_nll_interp(x1, x2, dec, targets):
    """Inference only: per-row NLL of the INTERPOLATED logits alpha (x1 W1^T + b1) + (1 - alpha) (x2 W2^T + b2)
    (compute_sentence_scores_bayes_jianwei.py:157-168) from ONE decoder + cross-entropy launch over the packed operands
    [alpha x1 | (1 - alpha) x2] . [W1 | W2]^T: neither model's (M, V) logits are stored.  ``dec``: InterpDecoder.  -> (M,) NLL"""
    if torch.is_grad_enabled() and (x1.requires_grad or x2.requires_grad):
        raise BayesLMError("linear_nll_interp is an inference-only path (no backward): call it under torch.no_grad()")
    a = _f32(x1, "x1").reshape(-1, x1.shape[-1])
    b = _f32(x2, "x2").reshape(-1, x2.shape[-1])
    a = a if a.stride(-1) == 1 else a.contiguous()
    b = b if b.stride(-1) == 1 else b.contiguous()
    M, K1 = a.shape
    K2 = b.shape[1]
    V = dec.w1.shape[0]
    if b.shape[0] != M or dec.w1.shape[1] != K1 or dec.w2.shape[1] != K2 or targets.numel() != M:
        raise ValueError("linear_nll_interp: x1 (M, K1), x2 (M, K2), decoders (V, K1) / (V, K2) and M targets expected")
    L.require_gfx950()
    tgt = dev_tensor(targets.reshape(-1), "targets", torch.int64)
    nll = torch.empty(M, device=a.device, dtype=torch.float32)
    if M == 0:
        return nll
    ws = torch.empty(int(lib().blm_linear_nll2_ws_floats(M, V, K1, K2)), device=a.device, dtype=torch.float32)
    check(lib().blm_linear_nll2(ptr(a), a.stride(0), ptr(dec.w1), dec.w1.stride(0), ptr(dec.b1), K1,
                                ptr(b), b.stride(0), ptr(dec.w2), dec.w2.stride(0), ptr(dec.b2), K2, dec.alpha,
                                ptr(tgt), ptr(nll), None, ptr(dec.wcat), 0 if dec.packed else 1, ptr(ws), M, V, stream()),
          "blm_linear_nll2")
    dec.packed = True
    return nll


# ----------------------------------------------------------------------------
# KL term  mean(mu^2 - 2 lg + exp(2 lg) [-1]) / 2  over a row window of mu
# ----------------------------------------------------------------------------
class _KLMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, lgstd, row_lo, minus_one, count_override):
        mu = _f32(mu, "mu")
        lgstd = _f32(lgstd, "lgstd")
        rows = lgstd.shape[0]
        cols = lgstd.numel() // rows
        ld = mu.numel() // mu.shape[0]
        out = torch.zeros((), device=mu.device, dtype=torch.float32)
        n = rows * cols
        w = float(n) / float(count_override) if count_override else 1.0
        L.require_gfx950()
        check(lib().blm_kl_mean_fwd(mu.data_ptr() + 4 * row_lo * ld, ld, ptr(lgstd), rows, cols, int(minus_one), w,
                                    ptr(out), stream()), "blm_kl_mean_fwd")
        ctx.meta = (mu, lgstd, row_lo, rows, cols, ld, w)
        return out

    @staticmethod
    def backward(ctx, g):
        mu, lgstd, row_lo, rows, cols, ld, w = ctx.meta
        g = _f32(g.reshape(1), "g")
        gm, gl = _grad_buf(mu), _grad_buf(lgstd)
        check(lib().blm_kl_mean_bwd(mu.data_ptr() + 4 * row_lo * ld, ld, ptr(lgstd), rows, cols, ptr(g), w,
                                    gm.data_ptr() + 4 * row_lo * ld, ld, ptr(gl), stream()), "blm_kl_mean_bwd")
        _notify(mu, lgstd)
        return None, None, None, None, None


def kl_mean(mu, lgstd, row_lo=0, minus_one=False, count=None):
    """KL of the rows [row_lo, row_lo+lgstd.shape[0]) of mu against lgstd.  ``count`` replaces the
    element count of the mean (Bayes2LSTM concatenates hh and ih before taking it, model.py:737-740)."""
    return _KLMean.apply(mu, lgstd, row_lo, minus_one, count)


def philox_normal(n, seed, stream_id, step, device="cuda"):
    out = torch.empty(n, device=device, dtype=torch.float32)
    L.require_gfx950()
    r = L.rng(seed, stream_id, step)
    check(lib().blm_philox_normal(ptr(out), n, C.byref(r), stream()), "blm_philox_normal")
    return out


# ----------------------------------------------------------------------------
# 2-layer LSTM stack as _VF.lstm computes it (model.py:812), explicit cell
# ----------------------------------------------------------------------------
_STATE_TAP = None
_PACK = threading.local()  # per thread: two scorers on two threads never see each other's layout


class packed_tokens:
    """Inference helper (the n-best scorer): inside the context the Transformer stacks keep their activations as a
    compact (R, 1, d) matrix of the REAL tokens of a padded (T, N) batch of hypotheses -- every token-wise operation
    (projections, feed-forward, layer norms, decoder) then runs on R rows instead of T * N (padding is 40-50 % of a
    batch of AMI-shaped hypotheses); only the attention core sees the padded layout (scatter before, gather after; the
    padding rows hold zeros and, the attention being causal and column-wise, never reach a real token).
    ``sel`` (R,) int64: flat indices t * N + n of the real tokens, in the order the caller wants the rows.
    The layout is thread-local and does not nest.  (The scorer also sets ``decoder.rows`` on the model it scores with:
    a model object serves one scoring call at a time.)"""

    def __init__(self, sel, T, N):
        self.sel, self.T, self.N = sel, int(T), int(N)
        self._rowmap = None
        self._rows_ok = True  # the packed-row attention kernel takes this batch (decided by the library at the first call)

    def __enter__(self):
        if torch.is_grad_enabled():
            raise BayesLMError("ops.packed_tokens is an inference-only layout")
        if getattr(_PACK, "cur", None) is not None:
            raise BayesLMError("ops.packed_tokens does not nest")
        _PACK.cur = self
        return self

    def __exit__(self, *exc):
        _PACK.cur = None
        return False

    def pack(self, x):
        """(T, N, W) -> (R, 1, W)"""
        return x.reshape(self.T * self.N, x.shape[-1]).index_select(0, self.sel).unsqueeze(1)

    def unpack(self, xc):
        """(R, 1, W) -> (T, N, W), zeros in the padding"""
        out = xc.new_zeros(self.T * self.N, xc.shape[-1])
        out.index_copy_(0, self.sel, xc.reshape(-1, xc.shape[-1]))
        return out.view(self.T, self.N, -1)

    def rowmap(self):
        """(T * N,) int32: the packed row of the token at padded position t * N + n, -1 for padding (built once per batch)."""
        if self._rowmap is None:
            m = torch.full((self.T * self.N,), -1, device=self.sel.device, dtype=torch.int32)
            m[self.sel] = torch.arange(self.sel.numel(), device=self.sel.device, dtype=torch.int32)
            self._rowmap = m
        return self._rowmap

    def attention(self, q, k, v, nhead):
        """The attention core of a packed batch: q / k / v (R, 1, d) (or q = the fused (R, 1, 3d) projection, k = v = None) ->
        (R, 1, d).  Short batches (T <= 32, head_dim 64: what n-best hypotheses are) run on the packed rows themselves
        (blm_attn_fwd_rows: the kernel finds a token's row through rowmap); anything else is scattered into the padded layout,
        run through the ordinary kernels and gathered back -- a zero fill, an index_copy and a gather per layer."""
        if self._rows_ok:
            if k is None:
                qkv = _f32(q, "qkv")
                d = qkv.shape[-1] // 3
                qq, kk, vv, ld = qkv, qkv[..., d:], qkv[..., 2 * d:], 3 * d
            else:
                qq, kk, vv = _f32(q, "q"), _f32(k, "k"), _f32(v, "v")
                d = ld = qq.shape[-1]
            R = qq.numel() // qq.shape[-1]
            out = torch.empty(R, 1, d, device=qq.device, dtype=torch.float32)
            L.require_gfx950()
            rc = lib().blm_attn_fwd_rows(qq.data_ptr(), kk.data_ptr(), vv.data_ptr(), ld, ptr(out), ptr(self.rowmap()), self.T, self.N,
                                         nhead, d // nhead, stream())
            if rc == 0:
                return out
            if rc != L.ERR_UNSUPPORTED:
                check(rc, "blm_attn_fwd_rows")
            self._rows_ok = False
        if k is None:
            return self.pack(attention(self.unpack(q), nhead))
        return self.pack(attention_qkv(self.unpack(q), self.unpack(k), self.unpack(v), nhead))


def packing():
    return getattr(_PACK, "cur", None)


class state_tap:
    """Inference helper: inside the context every fused LSTM layer forward also records its (h, c) AFTER the time
    steps ``idx`` (int64 device tensor) -- the scorer walks the carry chain of a whole n-best file as one long B = 1
    sequence and reads the states at the utterance boundaries (compute_sentence_scores.py).  ``layers`` holds one
    (h (n,B,H), c (n,B,H)) pair per layer call, in call order; step-wise cells do not report (the caller checks)."""

    def __init__(self, idx):
        self.idx = idx + 1  # row t+1 of the (T+1,B,H) state buffers = state after step t
        self.layers = []

    def __enter__(self):
        global _STATE_TAP
        _STATE_TAP = self
        return self

    def __exit__(self, *exc):
        global _STATE_TAP
        _STATE_TAP = None
        return False


def _bias_pair_grads(b_refs, dbs, needs):
    """b_ih and b_hh of an LSTM layer receive the same gradient ``db`` (one per layer in ``dbs``; ``b_refs`` the (b_ih, b_hh)
    parameters layer by layer).  Leaf biases take it in place -- ONE launch for all of them (blm_init_multi: grad = grad + db) --
    and autograd gets None; anything else gets db itself.  -> the values to return, in b_refs' order."""
    items, out, told = [], [], []
    for i, b in enumerate(b_refs):
        db = dbs[i // 2]
        if not needs[i]:
            out.append(None)
        elif b.is_leaf and b.is_contiguous() and b.dtype == torch.float32:
            g = _grad_buf(b)
            items.append((g, g, db))
            told.append(b)
            out.append(None)
        else:
            out.append(db)
    # a cell that passes ONE parameter as both biases (VLSTMCell / GPLSTMCell add bias_ih twice, model.py:2519, :1750-1752) gets
    # db twice: the two updates of one buffer must not share a launch
    while items:
        seen, now, later = set(), [], []
        for it in items:
            (later if it[0].data_ptr() in seen else now).append(it)
            seen.add(it[0].data_ptr())
        _init_multi(now)
        items = later
    _notify(*told)
    return out


class _LSTMLayer(torch.autograd.Function):
    """One layer over T steps.  Input GEMM batched over T (M = T*B), recurrent GEMM + fused cell per
    step.  Weights arrive already sampled (W = mu + noise on the gate rows)."""

    @staticmethod
    def forward(ctx, x, h0, c0, w_ih, w_hh, b_ih, b_hh, noise_rows=None):
        x = _f32(x, "x")
        T, B, E = x.shape
        if noise_rows is not None:  # (T, H): row t is added to every batch row of h_t (VLSTMCell)
            noise_rows = _f32(noise_rows, "noise_rows")
        H = w_hh.shape[1]
        G = 4 * H
        dev = x.device
        bias = torch.empty(G, device=dev, dtype=torch.float32)
        hs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        cs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        _init_multi([(bias, b_ih, b_hh), (hs[0], h0, None), (cs[0], c0, None)])  # one launch: b_ih + b_hh, the initial state
        xw = torch.empty(T, B, G, device=dev, dtype=torch.float32)
        gemm(L.GEMM_NT, x, w_ih, xw, T * B, G, E, E, E, G, epilogue=L.EPI_BIAS, bias=bias)
        ga = torch.empty(T, B, G, device=dev, dtype=torch.float32)
        st = stream()
        # one launch per step (recurrent product + cell, blm_lstm_step_fwd) when the shape allows it,
        # else skinny GEMM + cell kernel
        fused_step = H % 32 == 0 and w_hh.data_ptr() % 16 == 0 and hs.data_ptr() % 16 == 0 and w_hh.is_contiguous()
        if fused_step and not torch.is_grad_enabled() and B >= _UNFUSED_STEP_B:
            # wide inference batches (the scorer packs hundreds of hypotheses per step): the fused step kernel re-reads W_hh
            # once per 32 batch rows and runs at 0.57 of the matrix peak there; the tiled GEMM + cell kernel pair is faster
            # from B ~ 200 on (n-best rescoring 64.5 k -> 70-72 k hypotheses/s)
            fused_step = False
        if fused_step:  # the whole layer from one call: T launches issued by the library
            ev = _TIMER.bracket("lstm_seq_fwd T=%d" % T) if _TIMER is not None else None
            if ev:
                ev[0].record()
            check(lib().blm_lstm_seq_fwd(ptr(xw), ptr(w_hh), ptr(hs), ptr(cs), ptr(ga), ptr(noise_rows), T, B, H, st),
                  "blm_lstm_seq_fwd")
            if ev:
                ev[1].record()
        else:
            hw = torch.empty(B, G, device=dev, dtype=torch.float32)
            for t in range(T):
                gemm(L.GEMM_NT, hs[t], w_hh, hw, B, G, H, H, H, G)
                check(lib().blm_lstm_cell_fwd(ptr(xw[t]), ptr(hw), ptr(cs[t]), ptr(hs[t + 1]), ptr(cs[t + 1]), ptr(ga[t]),
                                              B, H, st), "blm_lstm_cell_fwd")
                if noise_rows is not None:
                    check(lib().blm_add_rowvec(ptr(hs[t + 1]), ptr(noise_rows[t]), B, H, st), "blm_add_rowvec")
        if _STATE_TAP is not None:
            _STATE_TAP.layers.append((hs.index_select(0, _STATE_TAP.idx), cs.index_select(0, _STATE_TAP.idx)))
        ctx.save_for_backward(x, hs, cs, ga, w_ih, w_hh)
        ctx.w_refs = (w_ih, w_hh)  # the parameter objects themselves (their .grad is the accumulation target)
        ctx.b_refs = (b_ih, b_hh)
        ctx.has_noise = noise_rows is not None
        ctx.set_materialize_grads(False)  # the final states usually feed nothing: their gradients arrive as None, not as zero fills
        return hs[1:], hs[T], cs[T]

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        x, hs, cs, ga, w_ih, w_hh = ctx.saved_tensors
        T, B, E = x.shape
        H = w_hh.shape[1]
        G = 4 * H
        dev = x.device
        dy = torch.zeros(T, B, H, device=dev, dtype=torch.float32) if dy is None else _f32(dy, "dy")
        dgates = torch.empty(T, B, G, device=dev, dtype=torch.float32)
        st = stream()
        dh = torch.empty(B, H, device=dev, dtype=torch.float32)
        dcs = torch.empty(2, B, H, device=dev, dtype=torch.float32)  # ping-pong dc buffers
        db = torch.empty(G, device=dev, dtype=torch.float32)
        _init_multi([(dh, dhT, None), (dcs[0], dcT, None), (db, None, None)])  # one launch: incoming state gradients (or zeros), db = 0
        fused_step = (H % 32 == 0 and w_hh.is_contiguous() and w_hh.data_ptr() % 16 == 0 and dgates.data_ptr() % 16 == 0)
        noise = getattr(ctx, "has_noise", False)
        # dhr[t] = gradient reaching h_t from step t+1 (dhr[T-1] = dhT); kept for every step only when the
        # additive noise rows need their gradient (column sums of dhr + dy)
        dhr = torch.zeros(T, B, H, device=dev, dtype=torch.float32) if (noise or not fused_step) else None
        if dhr is not None:
            dhr[T - 1].copy_(dh)
        if fused_step:
            # one launch per step: dh_{t-1} = dgates_t . W_hh on the matrix cores with the cell backward
            # of step t-1 fused behind it (blm_lstm_step_bwd); W_hh is transposed once per layer
            w_t = torch.empty(H, G, device=dev, dtype=torch.float32)
            check(lib().blm_transpose(ptr(w_hh), ptr(w_t), G, H, st), "blm_transpose")
            ev = _TIMER.bracket("lstm_seq_bwd T=%d" % T) if _TIMER is not None else None
            if ev:
                ev[0].record()
            # the whole chain from one call (step T-1: the plain cell backward; every earlier step one fused launch)
            check(lib().blm_lstm_seq_bwd(ptr(dh), ptr(dy), ptr(cs), ptr(ga), ptr(w_t), ptr(dgates), ptr(dcs), 0,
                                         ptr(dhr) if noise else None, T, T, 0, B, H, st), "blm_lstm_seq_bwd")
            k = T & 1
            dh = torch.empty(B, H, device=dev, dtype=torch.float32)
            check(lib().blm_lstm_step_bwd(ptr(dgates[0]), ptr(w_t), None, None, None, None, None, None, None, ptr(dh),
                                          B, H, st), "blm_lstm_step_bwd")
            if ev:
                ev[1].record()
            dc = dcs[k]
        else:
            # recurrent dh of every step accumulates (split-K atomics) into one pre-zeroed buffer: a
            # single memset per layer instead of one in front of every skinny GEMM
            dh0 = torch.zeros(B, H, device=dev, dtype=torch.float32)
            for t in range(T - 1, -1, -1):
                k = (T - 1 - t) & 1
                check(lib().blm_lstm_cell_bwd2(ptr(dhr[t]), ptr(dy[t]), ptr(dcs[k]), ptr(cs[t]), ptr(cs[t + 1]), ptr(ga[t]),
                                               ptr(dgates[t]), ptr(dcs[k ^ 1]), B, H, st), "blm_lstm_cell_bwd2")
                gemm(L.GEMM_NN, dgates[t], w_hh, dhr[t - 1] if t > 0 else dh0, B, H, G, G, H, H, accumulate=True)
            dh = dh0
            dc = dcs[T & 1]
        d_noise = None
        if noise:  # noise row t was added to every batch row of h_t: its gradient is the column sum of dh_t
            d_noise = torch.empty(T, H, device=dev, dtype=torch.float32)
            tot = dhr + dy
            for t in range(T):
                _colsum_into(tot[t], B, H, d_noise[t], accumulate=False)
        dx = torch.empty_like(x)
        gemm(L.GEMM_NN, dgates, w_ih, dx, T * B, E, G, G, E, E)
        # leaf weights (nn.LSTM-style parameters) accumulate straight into .grad like every other wgrad of the engine -- returned
        # to autograd they cost an AccumulateGrad add over 16.8 MB each at H = 1024 (~10 us per weight and step); sampled
        # (non-leaf) weights of Bayes2LSTM get a fresh gradient tensor for their sampling node
        w_ih_p, w_hh_p = ctx.w_refs
        dw_ih = dw_hh = None
        if ctx.needs_input_grad[3]:
            buf_ih, acc_ih, dw_ih = _wgrad_target(w_ih_p)
            # the bias gradient = column sums of dgates, taken from the A tiles this weight-gradient GEMM stages anyway
            gemm(L.GEMM_TN, dgates, x, buf_ih, G, E, T * B, G, E, E, accumulate=acc_ih, colsum_a=db)
        else:
            _colsum_into(dgates, T * B, G, db, accumulate=False)
        if ctx.needs_input_grad[4]:
            buf_hh, acc_hh, dw_hh = _wgrad_target(w_hh_p)
            gemm(L.GEMM_TN, dgates, hs, buf_hh, G, H, T * B, G, H, H, accumulate=acc_hh)  # hs[0:T] = h_{t-1}
        _notify(w_ih_p, w_hh_p)
        db_ih, db_hh = _bias_pair_grads(ctx.b_refs, [db], ctx.needs_input_grad[5:7])
        return dx, dh, dc, dw_ih, dw_hh, db_ih, db_hh, d_noise


def _pad_gate_blocks(t, H, Hp, cols=False):
    """(4H, ...) -> (4Hp, ...): gate block g = rows [g H, (g + 1) H) moves to [g Hp, g Hp + H), zeros behind it (``cols``: the last
    dimension H -> Hp as well).  Differentiable (F.pad): its backward is the slice that drops the padding's gradient."""
    t4 = t.reshape(4, H, *t.shape[1:])
    pad = [0, Hp - H] if t4.dim() == 2 else ([0, Hp - H if cols else 0] + [0, 0] * (t4.dim() - 3) + [0, Hp - H])
    return torch.nn.functional.pad(t4, pad).reshape(4 * Hp, *([Hp] if cols else t.shape[1:]))


_PAD_HIDDEN_FROM = 64  # smaller layers stay on the GEMM + cell composition (padding 8 units to 32 would quadruple them)


def lstm_layer(x, h0, c0, w_ih, w_hh, b_ih, b_hh, noise_rows=None):
    """One LSTM layer over T steps.  ``noise_rows`` (T, H), optional: row t is added to every batch row
    of h_t after the cell and carried into step t+1 (VLSTMCell, reference model.py:2523-2527).

    A hidden size that is not a multiple of 32 (the classic word-language-model sizes: 200, 650, 1500 -- train.py's own default is
    200) does not fit the fused step kernels' tiles and used to fall back to one skinny GEMM + one cell kernel per step: 1.4-1.9 x
    the step time of the next multiple of 32 (tools/lstm_hidden_size_probe.py).  It is zero-padded to that multiple instead: padded
    units have zero weights and a zero bias, so their gates are (1/2, 1/2, 0, 1/2), their cell and output stay exactly 0, and they
    feed nothing back (their columns of W_hh are zero) -- the real units compute what they computed before.  The padding and the
    slices that undo it are torch ops: autograd routes the gradients back to the unpadded parameters."""
    H = w_hh.shape[1]
    B = x.shape[1] if x.dim() == 3 else 0
    if (H % 32 and H >= _PAD_HIDDEN_FROM and x.dim() == 3 and w_hh.dim() == 2 and w_hh.shape[0] == 4 * H
            and (torch.is_grad_enabled() or B < _UNFUSED_STEP_B)):
        Hp = (H + 31) // 32 * 32
        pad = torch.nn.functional.pad
        y, hT, cT = _LSTMLayer.apply(x, pad(h0, (0, Hp - H)), pad(c0, (0, Hp - H)), _pad_gate_blocks(w_ih, H, Hp),
                                     _pad_gate_blocks(w_hh, H, Hp, cols=True), _pad_gate_blocks(b_ih, H, Hp),
                                     _pad_gate_blocks(b_hh, H, Hp), None if noise_rows is None else pad(noise_rows, (0, Hp - H)))
        return y[..., :H], hT[..., :H], cT[..., :H]
    return _LSTMLayer.apply(x, h0, c0, w_ih, w_hh, b_ih, b_hh, noise_rows)


_SIDE_STREAMS = {}


def _side_stream(i=0):
    """Side stream i of the layer wavefront (0: the other layer's recurrence, 1: the per-chunk GEMMs between the layers)."""
    st_ = _SIDE_STREAMS.get(i)
    if st_ is None:
        st_ = _SIDE_STREAMS[i] = torch.cuda.Stream()
    return st_


# no-grad forwards at B >= 256 (the scorer's packed batches) take the tiled GEMM + cell kernel: the fused step kernel re-reads
# W_hh once per 32 batch rows (measured: n-best rescoring 64.5 k -> 70-72 k hypotheses/s with the threshold at ~200-256)
_UNFUSED_STEP_B = 256


def _stack_chunks(T):
    """Time chunks of the layer wavefront: layer 2 runs one chunk behind layer 1.  Small enough that the lag is a
    small part of T, large enough that the host's per-chunk work (two library calls, one GEMM, two events) stays cheap."""
    # T 35 (BASELINE configs[0]: a 2.2 ms step of which the host needs 1.6-1.9 ms to issue): 3 x 12; smaller chunks win 2 % when the
    # host keeps up and lose 10-25 % when it does not (round 4, tools/wf_probe.py).  T 100 (the LSTM recipe: 10 ms step, 5 ms of
    # host): 13 chunks of 8 beat 7 of 15 by 1.2-1.5 % now that the per-chunk GEMMs run on a stream of their own (9.80-9.86 against
    # 9.94-10.02 ms, four runs each).  A B = 1 chain of thousands of steps (the scorer's carry chain) runs best on 128-step
    # chunks (62 k -> 66 k hypotheses/s).
    cmax = 16 if T < 64 else (8 if T <= 256 else 128)
    n = (T + cmax - 1) // cmax
    c = (T + n - 1) // n
    return [(t0, min(T, t0 + c)) for t0 in range(0, T, c)]


class _LSTMStack2(torch.autograd.Function):
    """Two stacked LSTM layers as a WAVEFRONT on two streams (what _VF.lstm / nn.LSTM(num_layers=2) compute,
    model.py:812, :35).  A fused step kernel occupies the chip for ~11 us but is bound by its launch -> load -> MFMA ->
    store latency chain, and two of them fit a CU (70 KB LDS, <= 256 VGPRs each): the steps of layer 2 over time chunk c
    run on a side stream WHILE layer 1 walks chunk c + 1 on the main stream (a step PAIR takes 15.4 us forward /
    17.4 us backward against 21.7 / 26.5 us back to back; tools/lstm_step_bench.py).  Per chunk: layer 1's steps, event,
    [inter-layer dropout of that chunk, RNNModel only] and layer 2's input GEMM over the chunk's rows on a THIRD stream (so
    that layer 2's recurrence only ever waits for the first of these GEMMs), layer 2's steps.  Layer 1's next chunk is
    issued before the host turns to layer 2's previous one.  Backward mirrors it (layer 2's chain leads on the side stream,
    the dgrad GEMM of the chunk on the stream between, layer 1 follows a chunk behind); both chains are issued by the library
    (blm_lstm_seq_fwd / blm_lstm_seq_bwd: ~3.3 us of host time per launch against ~5 us of device time per launch when
    two recurrences are in flight); the weight-gradient GEMMs stay batched over all T at the end.  Same kernels and the same
    arithmetic per step as two ops.lstm_layer calls: the forward is bit-identical to them; the backward is equal up to
    summation order (the per-chunk dgrad GEMMs accumulate into a pre-zeroed dy1, with K slices through float atomics where the
    planner picks them -- 1e-6 run to run; deterministic mode, ops.set_deterministic, takes the one-stream path instead)."""

    @staticmethod
    def forward(ctx, x, h0a, c0a, h0b, c0b, w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, drop):
        x = _f32(x, "x")
        T, B, E = x.shape
        H = w_hh1.shape[1]
        G = 4 * H
        dev = x.device
        st = stream
        lib_ = lib()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        bias1, bias2 = new(G), new(G)
        hs1, cs1, ga1 = new(T + 1, B, H), new(T + 1, B, H), new(T, B, G)
        hs2, cs2, ga2 = new(T + 1, B, H), new(T + 1, B, H), new(T, B, G)
        # one launch: both layers' b_ih + b_hh and the four initial states into row 0 of the state histories
        _init_multi([(bias1, b_ih1, b_hh1), (bias2, b_ih2, b_hh2), (hs1[0], h0a, None), (cs1[0], c0a, None), (hs2[0], h0b, None),
                     (cs2[0], c0b, None)])
        xw1 = new(T, B, G)
        gemm(L.GEMM_NT, x, w_ih1, xw1, T * B, G, E, E, E, G, epilogue=L.EPI_BIAS, bias=bias1)
        xw2 = new(T, B, G)
        x2 = new(T, B, H) if drop.on else None  # layer 2's input = drop(h1) (nn.LSTM's inter-layer dropout)
        main, side = torch.cuda.current_stream(), _side_stream()
        tev = _TIMER.bracket("lstm_stack2_fwd T=%d" % T) if _TIMER is not None else None
        if tev:
            tev[0].record()
        bh, bg = B * H * 4, B * G * 4  # bytes per time row
        p_xw1, p_xw2 = xw1.data_ptr(), xw2.data_ptr()
        p = {k: v.data_ptr() for k, v in (("hs1", hs1), ("cs1", cs1), ("ga1", ga1), ("hs2", hs2), ("cs2", cs2), ("ga2", ga2))}
        chunks = _stack_chunks(T)
        if B <= 4:
            # tiny batches (the scorer's carry chain): layer 1 over chunk c and layer 2 over chunk c - 1 from ONE stream, a step
            # of each per launch (blm_lstm_seq_pair_fwd) -- no second stream, no events, half the launches
            prev = None
            for cur in chunks + [None]:
                if prev is not None:
                    a0, a1 = prev
                    inp = hs1[a0 + 1:a1 + 1]
                    if drop.on:
                        inp = _dropout_apply(inp, drop, row0=a0, out=x2[a0:a1])
                    gemm(L.GEMM_NT, inp, w_ih2, xw2[a0:a1], (a1 - a0) * B, G, H, H, H, G, epilogue=L.EPI_BIAS, bias=bias2)
                t0, t1 = cur if cur is not None else (0, 0)
                a0, a1 = prev if prev is not None else (0, 0)
                check(lib_.blm_lstm_seq_pair_fwd(p_xw1 + t0 * bg, ptr(w_hh1), p["hs1"] + t0 * bh, p["cs1"] + t0 * bh, p["ga1"] + t0 * bg, t1 - t0,
                                                 p_xw2 + a0 * bg, ptr(w_hh2), p["hs2"] + a0 * bh, p["cs2"] + a0 * bh, p["ga2"] + a0 * bg, a1 - a0,
                                                 B, H, st()), "blm_lstm_seq_pair_fwd")
                prev = cur
            chunks = []
        else:
            # many chunks (T >= 64): the per-chunk GEMMs get a stream of their own; few (T 35: three chunks, a step the host can
            # barely issue in time): they stay in front of layer 2's steps on the side stream, two events per chunk fewer
            three = len(chunks) > 4
            between = _side_stream(1) if three else side
            side.wait_stream(main)
            if three:
                between.wait_stream(main)

        def layer1(t0, t1):
            check(lib_.blm_lstm_seq_fwd(p_xw1 + t0 * bg, ptr(w_hh1), p["hs1"] + t0 * bh, p["cs1"] + t0 * bh, p["ga1"] + t0 * bg,
                                        None, t1 - t0, B, H, st()), "blm_lstm_seq_fwd")
            ev = torch.cuda.Event()
            ev.record(main)
            return ev

        def layer2(t0, t1, ev):
            n = t1 - t0
            # layer 2's input rows of this chunk on a stream of their own: layer 2's recurrence (one chunk behind) never waits
            # for a GEMM except in front of its first chunk
            with torch.cuda.stream(between):
                between.wait_event(ev)
                inp = hs1[t0 + 1:t1 + 1]
                if drop.on:
                    inp = _dropout_apply(inp, drop, row0=t0, out=x2[t0:t1])
                gemm(L.GEMM_NT, inp, w_ih2, xw2[t0:t1], n * B, G, H, H, H, G, epilogue=L.EPI_BIAS, bias=bias2)
                if three:
                    ev2 = torch.cuda.Event()
                    ev2.record(between)
            with torch.cuda.stream(side):
                if three:
                    side.wait_event(ev2)
                check(lib_.blm_lstm_seq_fwd(p_xw2 + t0 * bg, ptr(w_hh2), p["hs2"] + t0 * bh, p["cs2"] + t0 * bh,
                                            p["ga2"] + t0 * bg, None, n, B, H, st()), "blm_lstm_seq_fwd")
        # issue order with three streams: layer 1's NEXT chunk goes to its stream before the host turns to layer 2's previous one,
        # so the leading recurrence never waits for the host
        pend = None
        for (t0, t1) in chunks:
            ev = layer1(t0, t1)
            if not three:
                layer2(t0, t1, ev)
                continue
            if pend is not None:
                layer2(*pend)
            pend = (t0, t1, ev)
        if pend is not None:
            layer2(*pend)
        if chunks:
            main.wait_stream(side)
        if tev:
            tev[1].record()
        if _STATE_TAP is not None:
            _STATE_TAP.layers.append((hs1.index_select(0, _STATE_TAP.idx), cs1.index_select(0, _STATE_TAP.idx)))
            _STATE_TAP.layers.append((hs2.index_select(0, _STATE_TAP.idx), cs2.index_select(0, _STATE_TAP.idx)))
        ctx.save_for_backward(x, hs1, cs1, ga1, hs2, cs2, ga2, x2, w_ih1, w_hh1, w_ih2, w_hh2)
        ctx.w_refs = (w_ih1, w_hh1, w_ih2, w_hh2)
        ctx.b_refs = (b_ih1, b_hh1, b_ih2, b_hh2)
        ctx.drop = drop
        ctx.set_materialize_grads(False)  # the final states usually feed nothing: their gradients arrive as None, not as zero fills
        return hs2[1:], hs1[T], cs1[T], hs2[T], cs2[T]

    @staticmethod
    def backward(ctx, dy, dh1T, dc1T, dh2T, dc2T):
        x, hs1, cs1, ga1, hs2, cs2, ga2, x2, w_ih1, w_hh1, w_ih2, w_hh2 = ctx.saved_tensors
        drop = ctx.drop
        T, B, E = x.shape
        H = w_hh1.shape[1]
        G = 4 * H
        dev = x.device
        lib_ = lib()
        st = stream
        dy = torch.zeros(T, B, H, device=dev, dtype=torch.float32) if dy is None else _f32(dy, "dy")
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731

        def state(w_hh):
            w_t = new(H, G)
            check(lib_.blm_transpose(ptr(w_hh), ptr(w_t), G, H, st()), "blm_transpose")
            return {"dh": new(B, H), "dcs": new(2, B, H), "k": 0, "w_t": w_t, "dg": new(T, B, G)}
        s1, s2 = state(w_hh1), state(w_hh2)
        dy1 = new(T, B, H)  # gradient reaching layer 1's outputs = layer 2's input gradient
        db1, db2 = new(G), new(G)
        # one launch: the incoming state gradients (or zeros) of both layers, zeroed bias gradients, and dy1 = 0 -- its chunks are
        # written by K-sliced products that would each need a memset of their own otherwise
        _init_multi([(s1["dh"], dh1T, None), (s1["dcs"][0], dc1T, None), (s2["dh"], dh2T, None), (s2["dcs"][0], dc2T, None),
                     (db1, None, None), (db2, None, None), (dy1, None, None)])
        bh, bg = B * H * 4, B * G * 4

        def chain(s, dyp, cs, ga, t_hi, t_lo):
            """dgates[t] for t = t_hi-1 .. t_lo (descending) of one layer from one call: the first launch of the whole chain is the
            plain cell backward of step T-1, every other one the fused step (dh_t = dgates[t+1] . W_hh, then the cell of step t)."""
            check(lib_.blm_lstm_seq_bwd(ptr(s["dh"]), dyp, ptr(cs), ptr(ga), ptr(s["w_t"]), ptr(s["dg"]), ptr(s["dcs"]), s["k"], None,
                                        T, t_hi, t_lo, B, H, st()), "blm_lstm_seq_bwd")
            s["k"] ^= (t_hi - t_lo) & 1
        main, side, between = torch.cuda.current_stream(), _side_stream(), _side_stream(1)
        dh01, dh02 = new(B, H), new(B, H)  # allocated on the main stream's pool, like everything else both streams touch
        tev = _TIMER.bracket("lstm_stack2_bwd T=%d" % T) if _TIMER is not None else None
        if tev:
            tev[0].record()
        chunks = _stack_chunks(T)
        three = len(chunks) > 4  # as in forward
        if not three:
            between = main
        side.wait_stream(main)
        if three:
            between.wait_stream(main)
        for (t0, t1) in reversed(chunks):
            with torch.cuda.stream(side):
                chain(s2, dy.data_ptr(), cs2, ga2, t1, t0)
                ev = torch.cuda.Event()
                ev.record(side)
            n = t1 - t0
            # layer 1's incoming gradient rows of this chunk: on the stream between the layers when there are many chunks (layer 1's
            # chain, one chunk behind, then only ever waits for the first of these GEMMs), else in front of layer 1's steps
            with torch.cuda.stream(between):
                between.wait_event(ev)
                gemm(L.GEMM_NN, s2["dg"][t0:t1], w_ih2, dy1[t0:t1], n * B, H, G, G, H, H, accumulate=True)  # into the zeroed rows
                if drop.on:
                    _dropout_apply(dy1[t0:t1], drop, row0=t0, out=dy1[t0:t1])
                if three:
                    ev2 = torch.cuda.Event()
                    ev2.record(between)
            if three:
                main.wait_event(ev2)
            chain(s1, dy1.data_ptr(), cs1, ga1, t1, t0)
        # gradient w.r.t. the initial states: dh_{-1} = dgates[0] . W_hh
        for s, strm, d in ((s2, side, dh02), (s1, main, dh01)):
            with torch.cuda.stream(strm):
                check(lib_.blm_lstm_step_bwd(ptr(s["dg"][0]), ptr(s["w_t"]), None, None, None, None, None, None, None, ptr(d),
                                             B, H, st()), "blm_lstm_step_bwd")
        main.wait_stream(side)
        if tev:
            tev[1].record()
        dc02, dc01 = s2["dcs"][s2["k"]], s1["dcs"][s1["k"]]
        dg1, dg2 = s1["dg"], s2["dg"]
        inp2 = x2 if drop.on else hs1[1:]
        dx = torch.empty_like(x)
        gemm(L.GEMM_NN, dg1, w_ih1, dx, T * B, E, G, G, E, E)
        # leaf weights accumulate straight into .grad (see _LSTMLayer.backward); sampled ones get fresh tensors
        outs = []
        summed = [False, False]  # the bias gradient = column sums of dgates: taken from the A tiles a weight-gradient GEMM stages anyway
        for j, (w_p, A, Bm, kdim) in enumerate(((ctx.w_refs[0], dg1, x, E), (ctx.w_refs[1], dg1, hs1, H), (ctx.w_refs[2], dg2, inp2, H),
                                                (ctx.w_refs[3], dg2, hs2, H))):
            if not w_p.requires_grad:
                outs.append(None)
                continue
            buf, acc, ret = _wgrad_target(w_p)
            cs_a = None
            if not summed[j // 2]:
                cs_a, summed[j // 2] = (db1, db2)[j // 2], True
            gemm(L.GEMM_TN, A, Bm, buf, G, kdim, T * B, G, kdim, kdim, accumulate=acc, colsum_a=cs_a)  # hs[0:T] = h_{t-1}
            outs.append(ret)
        _notify(*ctx.w_refs)
        dw_ih1, dw_hh1, dw_ih2, dw_hh2 = outs
        for done, dgl, dbl in ((summed[0], dg1, db1), (summed[1], dg2, db2)):
            if not done:
                _colsum_into(dgl, T * B, G, dbl, accumulate=True)
        db_ih1, db_hh1, db_ih2, db_hh2 = _bias_pair_grads(ctx.b_refs, [db1, db2], [ctx.needs_input_grad[i] for i in (7, 8, 11, 12)])
        return dx, dh01, dc01, dh02, dc02, dw_ih1, dw_hh1, db_ih1, db_hh1, dw_ih2, dw_hh2, db_ih2, db_hh2, None


def lstm_stack2_ok(x, w_hh1, w_hh2, w_ih2):
    """Shapes the layer wavefront takes: fused step kernels (H % 32 == 0, aligned, contiguous) and equal hidden sizes --
    and, unless it was switched on or off explicitly, the shapes it PAYS for (see below)."""
    T, B, _ = x.shape
    H = w_hh1.shape[1]
    if _STACK2_ON is None:
        # deterministic mode keeps the stack on one stream unless the wavefront is forced on (set_lstm_wavefront(True)): with one K
        # slice per tile its per-chunk products are bit-identical to the sequential layers', which is what makes the forced form
        # a bitwise race check of the stream schedule (tests/test_gpu_deterministic.py)
        on = B <= 32 and T >= 32 and H >= 640 and not is_deterministic()
    else:
        on = _STACK2_ON
    return (on and T >= 8 and H % 32 == 0 and w_hh2.shape[1] == H and w_ih2.shape[1] == H and w_hh1.is_contiguous() and w_hh2.is_contiguous()
            and w_hh1.data_ptr() % 16 == 0 and w_hh2.data_ptr() % 16 == 0 and (B * H) % 4 == 0)


# When does the wavefront pay?  A fused step kernel launches ceil(B / 32) * H / 8 workgroups (forward) -- at B <= 32 and
# H = 1024 that is 128, HALF the chip -- and is bound by its launch -> load -> MFMA -> store latency chain, so two of them
# side by side cost what one does.  Measured (tools/run_workload.py, same box, off / on):
#   recipe shape (run_nnlm_ami_lstm.sh: T 100, B 32): 11.07 -> 9.96 ms per training step (289 k -> 321 k tokens/s);
#   cfg2 (T 35, B 64: the kernels fill the chip, layer 2's per-chunk input GEMM of M = 320 rows and the one-chunk lag give the
#   overlap back): 6.65 -> 6.74 ms;  cfg1 (T 35, B 20): 2.31 -> 2.21 ms with three chunks of 12 steps (2.30 with 5-step
#   chunks, 2.31 with 16 + 16 + 3), evaluate() at T 35 / B 20 1.14 -> 1.07 ms.
#   smaller layers (round 5, tools/lstm_hidden_size_probe.py, T 35, B 20, off / on): H 128: 528 / 421 k tokens/s, 256: 500 / 447 k,
#   512: 465 / 441 k, 768: 362 / 361 k, 672: 349 / 386 k, 1536: 197 / 201 k -- the step kernels of a small layer are a few dozen
#   workgroups and the step is host-bound: the second stream's events cost more than the overlap returns.
# Rule: on for B <= 32, T >= 32 and H >= 640; BLM_LSTM_WAVEFRONT=0|1 / set_lstm_wavefront(True | False) force it, None = rule.
_env_wf = os.environ.get("BLM_LSTM_WAVEFRONT")
_STACK2_ON = None if _env_wf is None else _env_wf == "1"


def set_lstm_wavefront(on):
    """True / False: two-layer LSTM stacks always / never run as a wavefront on two streams (ops.lstm_stack2); None: by the
    measured rule above."""
    global _STACK2_ON
    _STACK2_ON = None if on is None else bool(on)


def lstm_stack2(x, h0, c0, layer1, layer2, drop=NO_DROP):
    """Two stacked LSTM layers; ``h0`` / ``c0`` are (2, B, H), ``layer*`` = (w_ih, w_hh, b_ih, b_hh), ``drop`` the
    inter-layer dropout (nn.LSTM's; NO_DROP for _VF.lstm(dropout=0.)).  -> (y (T,B,H), (h1T, h2T), (c1T, c2T))"""
    y, h1, c1, h2, c2 = _LSTMStack2.apply(x, h0[0], c0[0], h0[1], c0[1], *layer1, *layer2, drop)
    return y, (h1, h2), (c1, c2)


class _LSTMRecurrentGP(torch.autograd.Function):
    """Recurrent part of a GP-LSTM layer on the fused step kernels.  ``xw`` (T,B,4H) holds the input-side
    pre-activations of all steps (biases inside), ``w_rec`` (4H,H) the recurrent rows.  ``ovr``:
      0..3  GPNN on that gate (GPLSTMCell gate types 1-4, reference model.py:1754-1771): block ``ovr`` of
            xw / w_rec is the GPNN's input / hidden part, ``coef4`` (4,H) its mixture coefficients;
      4     gate type 6 (model.py:1744-1752): the whole hidden projection h w_rec^T + rbias goes through
            the mixture (``coef4`` (4,4H)) before it is added to xw;
      -1    plain recurrence on a given xw (gate type 7: xw is the GPNN of the inputs)."""

    @staticmethod
    def forward(ctx, xw, h0, c0, w_rec, coef4, ovr, rbias, w_cell=None):
        xw, w_rec = _f32(xw, "xw"), _f32(w_rec, "w_rec")
        ovr = int(ovr)
        coef4 = _f32(coef4, "coef4") if ovr >= 0 else None
        rbias = _f32(rbias, "rbias") if ovr >= 4 else None
        w_cell = _f32(w_cell, "w_cell") if ovr == 5 else None
        T, B, G = xw.shape
        H = G // 4
        if tuple(w_rec.shape) != (G, H) or (ovr == 5 and tuple(w_cell.shape) != (H, H)):
            # the step kernels take these as (4H, H) / (H, H) operands on trust
            raise BayesLMError("lstm_recurrent_gp: w_rec %s / w_cell %s do not fit xw (T, B, %d)"
                               % (tuple(w_rec.shape), None if w_cell is None else tuple(w_cell.shape), G))
        dev = xw.device
        L.require_gfx950()
        hs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        cs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        hs[0].copy_(h0)
        cs[0].copy_(c0)
        ga = torch.empty(T, B, G, device=dev, dtype=torch.float32)
        zs = torch.empty(T, B, G if ovr == 4 else H, device=dev, dtype=torch.float32) if ovr >= 0 else None
        st = stream()
        xw_p, hs_p, cs_p, ga_p = _P(xw), _P(hs), _P(cs), _P(ga)
        zs_p = None if zs is None else _P(zs)
        w_p, co_p, rb_p = ptr(w_rec), ptr(coef4), ptr(rbias)
        step_fwd = lib().blm_lstm_step_fwd_gp
        step_dh = lib().blm_lstm_step_dh
        wc_p = ptr(w_cell)
        for t in range(T):
            if ovr == 5:  # z_t = c_{t-1} Wg^T: the second recurrent product of the step, one skinny launch in front of it
                check(step_dh(cs_p[t], wc_p, zs_p[t], B, H, H, st), "blm_lstm_step_dh")
            check(step_fwd(xw_p[t], w_p, hs_p[t], cs_p[t], hs_p[t + 1], cs_p[t + 1], ga_p[t], None, ovr, co_p, rb_p,
                           None if zs_p is None else zs_p[t], B, H, st), "blm_lstm_step_fwd_gp")
        if _STATE_TAP is not None:
            _STATE_TAP.layers.append((hs.index_select(0, _STATE_TAP.idx), cs.index_select(0, _STATE_TAP.idx)))
        ctx.save_for_backward(hs, cs, ga, w_rec, *([zs, coef4] if ovr >= 0 else []), *([w_cell] if ovr == 5 else []))
        ctx.ovr = ovr
        return hs[1:], hs[T], cs[T]

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        ovr = ctx.ovr
        w_cell = None
        if ovr == 5:
            hs, cs, ga, w_rec, zs, coef4, w_cell = ctx.saved_tensors
        elif ovr >= 0:
            hs, cs, ga, w_rec, zs, coef4 = ctx.saved_tensors
        else:
            hs, cs, ga, w_rec = ctx.saved_tensors
            zs = coef4 = None
        T, B, G = ga.shape
        H = G // 4
        dev = ga.device
        dy = _f32(dy, "dy")
        st = stream()
        dgates = torch.empty(T, B, G, device=dev, dtype=torch.float32)
        dact = torch.empty(T, B, H, device=dev, dtype=torch.float32) if (0 <= ovr < 4 or ovr == 5) else None
        # A operands of the next products: ovr 4 -- d z of the hidden projection (B,4H); ovr 5 -- d z of the cell-state GPNN (B,H)
        dzs = torch.empty(T, B, G if ovr == 4 else H, device=dev, dtype=torch.float32) if ovr >= 4 else None
        dh = torch.zeros(B, H, device=dev, dtype=torch.float32) if dhT is None else _f32(dhT, "dhT").clone()
        dcs = torch.zeros(2, B, H, device=dev, dtype=torch.float32)
        if dcT is not None:
            dcs[0].copy_(dcT)
        w_t = torch.empty(H, G, device=dev, dtype=torch.float32)
        check(lib().blm_transpose(ptr(w_rec), ptr(w_t), G, H, st), "blm_transpose")
        # last step: plain cell backward (the GP gate as an external activation), then the mixture's derivative
        if 0 <= ovr < 4:
            check(lib().blm_axpy(ptr(dy[T - 1]), ptr(dh), B * H, 1.0, st), "blm_axpy")
            check(lib().blm_lstm_cell_ovr_bwd(ptr(dh), ptr(dcs[0]), ptr(cs[T - 1]), ptr(cs[T]), ptr(ga[T - 1]), ovr,
                                              ptr(dgates[T - 1]), ptr(dact[T - 1]), ptr(dcs[1]), B, H, st), "blm_lstm_cell_ovr_bwd")
            dz = torch.empty(B, H, device=dev, dtype=torch.float32)
            check(lib().blm_gp_mix_bwd(ptr(dact[T - 1]), ptr(zs[T - 1]), ptr(coef4), ptr(dz), B, H, st), "blm_gp_mix_bwd")
            dgates[T - 1][:, ovr * H:(ovr + 1) * H].copy_(dz)
        else:
            c_in = cs[T - 1]
            if ovr == 5:  # the last step's cell saw the mixture of z_{T-1}
                c_in = torch.empty(B, H, device=dev, dtype=torch.float32)
                check(lib().blm_gp_mix_fwd(ptr(zs[T - 1]), ptr(coef4), ptr(c_in), B, H, st), "blm_gp_mix_fwd")
            check(lib().blm_lstm_cell_bwd2(ptr(dh), ptr(dy[T - 1]), ptr(dcs[0]), ptr(c_in), ptr(cs[T]), ptr(ga[T - 1]),
                                           ptr(dgates[T - 1]), ptr(dcs[1]), B, H, st), "blm_lstm_cell_bwd2")
            if ovr == 5:
                w_cell_t = torch.empty(H, H, device=dev, dtype=torch.float32)
                check(lib().blm_transpose(ptr(w_cell), ptr(w_cell_t), H, H, st), "blm_transpose")
                dact[T - 1].copy_(dcs[1])
                check(lib().blm_gp_mix_bwd(ptr(dcs[1]), ptr(zs[T - 1]), ptr(coef4), ptr(dzs[T - 1]), B, H, st), "blm_gp_mix_bwd")
                check(lib().blm_lstm_step_dh(ptr(dzs[T - 1]), ptr(w_cell_t), ptr(dcs[1]), B, H, H, st), "blm_lstm_step_dh")
            if ovr == 4:
                check(lib().blm_gp_mix_bwd(ptr(dgates[T - 1]), ptr(zs[T - 1]), ptr(coef4), ptr(dzs[T - 1]), B, G, st),
                      "blm_gp_mix_bwd")
        A = dzs if ovr == 4 else dgates
        k = 1
        A_p, dy_p, cs_p, ga_p, dg_p, dcs_p = _P(A), _P(dy), _P(cs), _P(ga), _P(dgates), _P(dcs)
        zs_p = None if zs is None else _P(zs)
        da_p = None if dact is None else _P(dact)
        dzs_p = None if dzs is None else _P(dzs)
        wt_p, co_p = ptr(w_t), ptr(coef4)
        step_bwd = lib().blm_lstm_step_bwd_gp
        wct_p = ptr(w_cell_t) if ovr == 5 else None
        step_dh = lib().blm_lstm_step_dh
        for t in range(T - 1, 0, -1):
            check(step_bwd(A_p[t], wt_p, dy_p[t - 1], dcs_p[k], cs_p[t - 1], cs_p[t], ga_p[t - 1], dg_p[t - 1], dcs_p[k ^ 1], None,
                           ovr, co_p, None if zs_p is None else zs_p[t - 1], None if da_p is None else da_p[t - 1],
                           None if dzs_p is None else dzs_p[t - 1], B, H, st), "blm_lstm_step_bwd_gp")
            if ovr == 5:  # raw cell-state gradient of the earlier step: d z . Wg
                check(step_dh(dzs_p[t - 1], wct_p, dcs_p[k ^ 1], B, H, H, st), "blm_lstm_step_dh")
            k ^= 1
        dh0 = torch.empty(B, H, device=dev, dtype=torch.float32)
        check(lib().blm_lstm_step_bwd(ptr(A[0]), ptr(w_t), None, None, None, None, None, None, None, ptr(dh0), B, H, st),
              "blm_lstm_step_bwd")
        dw = torch.empty_like(w_rec)
        gemm(L.GEMM_TN, A, hs, dw, G, H, T * B, G, H, H)  # hs[0:T] = h_{t-1}
        dcoef = drb = dwc = None
        if 0 <= ovr < 4 or ovr == 5:
            dcoef = torch.zeros_like(coef4)
            check(lib().blm_gp_coef_grad(ptr(dact), ptr(zs), ptr(dcoef), T * B, H, st), "blm_gp_coef_grad")
        if ovr == 5:  # the cell-state GPNN's affine map, batched over all steps: dWg = dz^T c_{t-1}, db = column sums of dz
            dwc = torch.empty_like(w_cell)
            gemm(L.GEMM_TN, dzs, cs, dwc, H, H, T * B, H, H, H)  # cs[0:T] = c_{t-1}
            drb = torch.empty(H, device=dev, dtype=torch.float32)
            _colsum_into(dzs, T * B, H, drb, accumulate=False)
        elif ovr == 4:
            dcoef = torch.zeros_like(coef4)
            check(lib().blm_gp_coef_grad(ptr(dgates), ptr(zs), ptr(dcoef), T * B, G, st), "blm_gp_coef_grad")
            drb = torch.empty(G, device=dev, dtype=torch.float32)
            _colsum_into(dzs, T * B, G, drb, accumulate=False)
        return dgates, dh0, dcs[k], dw, dcoef, None, drb, dwc


class _LSTMRecurrentGPNN2(torch.autograd.Function):
    """GP-LSTM layer whose cell holds a GPNN2 that draws FRESH frequencies at every time step (GPLSTMCell with type digit 4,
    model.py:1698-1702, 1744-1771; GPNN2 :2061-2076), from ONE autograd node.  GPNN2_t(x) = (actsum(x F_t) | 1) [W | b]^T,
    F_t = mean + eps_t * exp(lgstd).  ``mode``:
      0  gate types 1-4: gate ``g``'s activation is GPNN2_t of its pre-activation  xw_t[:, g] + (h_{t-1} W_hh^T)[:, g]
      1  gate type 5:    the cell state enters the update as GPNN2_t(c_{t-1})
      2  gate type 6:    the hidden projection of all four gates is GPNN2_t(h_{t-1})  (w_hh unused)
    Per step 4-6 skinny launches (products on blm_lstm_step_dh: fixed summation order, no split-K atomics, no memset)
    instead of ~10 launches plus ten autograd nodes; the time loops run in C (blm_lstm_gpnn2_seq_fwd / _bwd).  The T
    frequency matrices are sampled in one launch (blm_gpnn2_sample_steps) with the Philox counters the step-wise path
    uses, so both paths see the same noise; the weight gradients are batched over all steps after the loop, the per-step
    frequency gradients x_t^T d f_t become d mean / d lgstd in one launch (blm_gpnn2_freq_grad, eps_t regenerated)."""

    MP, GP = 160, 192  # feature columns padded to the products' tile rules (output % 16, contraction % 64); 150 MC terms

    @staticmethod
    def _noise(noises, dev):
        """-> (eps_all (T,H,M) or None, rng of call 0 or None): injected tensors are stacked, Philox specs must be the
        consecutive steps GPNN2.step_noises hands out."""
        if noises is None:
            return None, None
        if any(n.eps is not None for n in noises):
            if any(n.eps is None for n in noises):  # a per-call eps_override list with holes
                raise BayesLMError("lstm_recurrent_gpnn2: the per-step noise is either injected for EVERY step or drawn from Philox for every step")
            return torch.stack([_f32(n.eps, "eps") for n in noises]).contiguous(), None
        for t, n in enumerate(noises):
            if n.eps is not None or n.seed != noises[0].seed or n.tensor_id != noises[0].tensor_id or n.step != ((noises[0].step + t) & 0xFFFFFFFF):
                raise BayesLMError("lstm_recurrent_gpnn2: the per-step noise must be one Philox stream at consecutive steps")
        return None, noises[0].rng()

    @staticmethod
    def _desc(mode, g, acts, T, B, H, M, nF, **bufs):
        q = L.Gpnn2Seq()
        q.abi_version, q.mode, q.gate, q.acts = L.ABI_VERSION, mode, g, acts
        q.T, q.B, q.H, q.M, q.MP, q.GP, q.nF = T, B, H, M, _LSTMRecurrentGPNN2.MP, _LSTMRecurrentGPNN2.GP, nF
        for k, v in bufs.items():
            setattr(q, k, ptr(v))
        return q

    @staticmethod
    def forward(ctx, xw, h0, c0, w_hh, coef_w, coef_b, fmean, flgstd, noises, g, acts, mode):
        xw = _f32(xw, "xw")
        w_hh = _f32(w_hh, "w_hh") if mode != 2 else None
        T, B, G4 = xw.shape
        H = G4 // 4
        M = fmean.shape[1]
        MP, GP = _LSTMRecurrentGPNN2.MP, _LSTMRecurrentGPNN2.GP
        NO = G4 if mode == 2 else H  # outputs of the GPNN2
        if M >= MP or H % 64 != 0 or fmean.shape[0] != H or coef_w.shape != (NO, M):
            raise BayesLMError("lstm_recurrent_gpnn2: needs n_MC_terms < %d, H %% 64 == 0, a GPNN2(H, %d)" % (MP, NO))
        dev = xw.device
        L.require_gfx950()
        lib_, st = lib(), stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        nF = T if noises is not None else 1
        # F_t^T with zero rows m >= M (the w_t operand of the feature product) and F_t padded along m (the w_t operand of
        # d x = d f . F_t^T), all T of them in ONE launch, with the noise of calls 0..T-1 of the step-wise path
        FT, Fp = new(nF, MP, H), new(nF, H, GP)
        eps_all, rng0 = _LSTMRecurrentGPNN2._noise(noises, dev)
        if noises is None:
            eps_all = torch.zeros(1, H, M, device=dev, dtype=torch.float32)  # mean frequencies at every step
        check(lib_.blm_gpnn2_sample_steps(ptr(_f32(fmean, "frequency_mean")), ptr(_f32(flgstd, "frequency_lgstd")), ptr(eps_all),
                                          C.byref(rng0) if rng0 is not None else None, nF, H, M, MP, GP, ptr(FT), ptr(Fp), st),
              "blm_gpnn2_sample_steps")
        cwp = torch.zeros(NO, GP, device=dev, dtype=torch.float32)  # [coef.weight | coef.bias | 0]
        cwp[:, :M].copy_(coef_w)
        cwp[:, M].copy_(coef_b)
        hs, cs = new(T + 1, B, H), new(T + 1, B, H)
        hs[0].copy_(h0)
        cs[0].copy_(c0)
        ga = new(T, B, G4)
        z4 = new(T, B, G4) if mode == 0 else None
        pre = new(T, B, H) if mode == 0 else None
        feat, gout = new(T, B, MP), new(T, B, NO)
        sact = torch.zeros(T, B, GP, device=dev, dtype=torch.float32)  # the feature product writes columns < MP; the padding stays 0
        q = _LSTMRecurrentGPNN2._desc(mode, g, acts, T, B, H, M, nF, xw=xw, w_hh=w_hh, FT=FT, Fp=Fp, cwp=cwp, hs=hs, cs=cs, ga=ga, z4=z4,
                                      pre=pre, feat=feat, sact=sact, gout=gout)
        check(lib_.blm_lstm_gpnn2_seq_fwd(C.byref(q), st), "blm_lstm_gpnn2_seq_fwd")
        if _STATE_TAP is not None:
            _STATE_TAP.layers.append((hs.index_select(0, _STATE_TAP.idx), cs.index_select(0, _STATE_TAP.idx)))
        ctx.save_for_backward(hs, cs, ga, feat, sact, Fp, cwp, *([w_hh] if mode != 2 else []), *([pre] if mode == 0 else []),
                              *([gout] if mode == 1 else []))
        ctx.meta = (mode, g, acts, M, fmean, flgstd, noises, coef_w.requires_grad, coef_b.requires_grad)
        return hs[1:], hs[T], cs[T]

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        mode, g, acts, M, fmean, flgstd, noises, need_cw, need_cb = ctx.meta
        hs, cs, ga, feat, sact, Fp, cwp = ctx.saved_tensors[:7]
        rest = list(ctx.saved_tensors[7:])
        w_hh = rest.pop(0) if mode != 2 else None
        pre = rest.pop(0) if mode == 0 else None
        gout = rest.pop(0) if mode == 1 else None
        T, B, G4 = ga.shape
        H = G4 // 4
        MP, GP = _LSTMRecurrentGPNN2.MP, _LSTMRecurrentGPNN2.GP
        NO = cwp.shape[0]
        dev = ga.device
        dy = _f32(dy, "dy")
        lib_, st = lib(), stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        nF = Fp.shape[0]
        w_t = None
        if mode != 2:
            w_t = new(H, G4)
            check(lib_.blm_transpose(ptr(w_hh), ptr(w_t), G4, H, st), "blm_transpose")
        cwt = new(GP, NO)
        check(lib_.blm_transpose(ptr(cwp), ptr(cwt), NO, GP, st), "blm_transpose")
        dgates, df = new(T, B, G4), new(T, B, GP)
        da = new(T, B, H) if mode != 2 else None
        dh = torch.zeros(B, H, device=dev, dtype=torch.float32) if dhT is None else _f32(dhT, "dhT").clone()
        dcs = torch.zeros(2, B, H, device=dev, dtype=torch.float32)
        if dcT is not None:
            dcs[0].copy_(dcT)
        q = _LSTMRecurrentGPNN2._desc(mode, g, acts, T, B, H, M, nF, w_hh_t=w_t, Fp=Fp, cwt=cwt, cs=cs, ga=ga, feat=feat, gout=gout, dy=dy,
                                      dh=dh, dcs2=dcs, dgates=dgates, da=da, df=df)
        check(lib_.blm_lstm_gpnn2_seq_bwd(C.byref(q), st), "blm_lstm_gpnn2_seq_bwd")
        k = T & 1
        dw = None
        if mode != 2:
            dw = torch.empty_like(w_hh)
            gemm(L.GEMM_TN, dgates, hs, dw, G4, H, T * B, G4, H, H)  # hs[0:T] = h_{t-1}
        dcw = dcb = None
        if need_cw or need_cb:
            dout = dgates if mode == 2 else da  # the gradient of the GPNN2's output, all steps
            dcwp = new(NO, GP)
            gemm(L.GEMM_TN, dout, sact, dcwp, NO, GP, T * B, NO, GP, GP)  # column M of s is the constant 1: its row is d coef.bias
            dcw = dcwp[:, :M].contiguous() if need_cw else None
            dcb = dcwp[:, M].contiguous() if need_cb else None
        # (mean frequencies -- eval mode or a deterministic GPNN2 -- give the lgstd no gradient: with the mean frozen there is nothing to do)
        if fmean.requires_grad or (flgstd.requires_grad and noises is not None):
            # d F_t = x_t^T d f_t (contraction over the B rows of ONE step: the frequencies differ per step); d mean is their
            # sum, d lgstd their eps_t-weighted sum times sigma -- one launch, eps_t regenerated from the Philox counters
            x_in = pre if mode == 0 else (cs if mode == 1 else hs)  # rows 0..T-1: the GPNN2's inputs
            eps_all, rng0 = _LSTMRecurrentGPNN2._noise(noises, dev)
            Tn, Bn = T, B
            if noises is None:  # mean frequencies (deterministic GPNN2): d mean only, one contraction over all T*B rows
                eps_all = torch.zeros(1, H, M, device=dev, dtype=torch.float32)
                Tn, Bn = 1, T * B
            check(lib_.blm_gpnn2_freq_grad(ptr(x_in), ptr(df), ptr(eps_all), C.byref(rng0) if rng0 is not None else None,
                                           ptr(flgstd), ptr(_grad_buf(fmean)) if fmean.requires_grad else None,
                                           ptr(_grad_buf(flgstd)) if (flgstd.requires_grad and noises is not None) else None,
                                           Tn, Bn, H, M, GP, st), "blm_gpnn2_freq_grad")
            _notify(fmean, flgstd)
        return dgates, dh, dcs[k], dw, dcw, dcb, None, None, None, None, None, None


class _GPNN2Steps(torch.autograd.Function):
    """out[t] = GPNN2_t(x[t]) for all T steps of a window -- a GPNN2 that is called once per time step and draws fresh
    frequencies at every call, on inputs that do not depend on the recurrence (GPLSTMCell gate type 7 with type digit 4:
    the input projection, model.py:1747-1748).  Only the feature product is per step (T skinny launches, independent of
    each other); the activation sum, the coefficient product and every gradient are batched over the T*B rows."""

    @staticmethod
    def forward(ctx, x, coef_w, coef_b, fmean, flgstd, noises, acts):
        x = _f32(x, "x")
        T, B, E = x.shape
        NO, M = coef_w.shape
        MP, GP = _LSTMRecurrentGPNN2.MP, _LSTMRecurrentGPNN2.GP
        if M >= MP or E % 64 != 0 or fmean.shape != (E, M):
            raise BayesLMError("gpnn2_steps: needs n_MC_terms < %d, input width %% 64 == 0" % MP)
        dev = x.device
        L.require_gfx950()
        lib_, st = lib(), stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        nF = T if noises is not None else 1
        FT, Fp = new(nF, MP, E), new(nF, E, GP)
        eps_all, rng0 = _LSTMRecurrentGPNN2._noise(noises, dev)
        if noises is None:
            eps_all = torch.zeros(1, E, M, device=dev, dtype=torch.float32)
        check(lib_.blm_gpnn2_sample_steps(ptr(_f32(fmean, "frequency_mean")), ptr(_f32(flgstd, "frequency_lgstd")), ptr(eps_all),
                                          C.byref(rng0) if rng0 is not None else None, nF, E, M, MP, GP, ptr(FT), ptr(Fp), st),
              "blm_gpnn2_sample_steps")
        cwp = torch.zeros(NO, GP, device=dev, dtype=torch.float32)
        cwp[:, :M].copy_(coef_w)
        cwp[:, M].copy_(coef_b)
        feat, sact = new(T, B, MP), new(T, B, GP)
        x_p, f_p, FT_p = _P(x), _P(feat), _P(FT)
        for t in range(T):
            check(lib_.blm_lstm_step_dh(x_p[t], FT_p[t if nF > 1 else 0], f_p[t], B, MP, E, st), "blm_lstm_step_dh")
        check(lib_.blm_gpnn2_actsum_fwd(ptr(feat), ptr(sact), T * B, M, MP, GP, 1.0 / math.sqrt(M), acts, st), "blm_gpnn2_actsum_fwd")
        out = new(T, B, NO)
        gemm(L.GEMM_NT, sact, cwp, out, T * B, NO, GP, GP, GP, NO)
        ctx.save_for_backward(x, feat, sact, Fp, cwp)
        ctx.meta = (acts, M, fmean, flgstd, noises, coef_w.requires_grad, coef_b.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, feat, sact, Fp, cwp = ctx.saved_tensors
        acts, M, fmean, flgstd, noises, need_cw, need_cb = ctx.meta
        T, B, E = x.shape
        NO = cwp.shape[0]
        MP, GP = _LSTMRecurrentGPNN2.MP, _LSTMRecurrentGPNN2.GP
        dev = x.device
        dout = _f32(dout, "dout")
        lib_, st = lib(), stream()
        new = lambda *sh: torch.empty(*sh, device=dev, dtype=torch.float32)  # noqa: E731
        nF = Fp.shape[0]
        ds = new(T, B, GP)
        gemm(L.GEMM_NN, dout, cwp, ds, T * B, GP, NO, NO, GP, GP)
        df = new(T, B, GP)
        check(lib_.blm_gpnn2_actsum_bwd(ptr(ds), ptr(feat), ptr(df), T * B, M, MP, GP, 1.0 / math.sqrt(M), acts, st), "blm_gpnn2_actsum_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new(T, B, E)
            df_p, dx_p, Fp_p = _P(df), _P(dx), _P(Fp)
            for t in range(T):
                check(lib_.blm_lstm_step_dh(df_p[t], Fp_p[t if nF > 1 else 0], dx_p[t], B, E, GP, st), "blm_lstm_step_dh")
        dcw = dcb = None
        if need_cw or need_cb:
            dcwp = new(NO, GP)
            gemm(L.GEMM_TN, dout, sact, dcwp, NO, GP, T * B, NO, GP, GP)
            dcw = dcwp[:, :M].contiguous() if need_cw else None
            dcb = dcwp[:, M].contiguous() if need_cb else None
        if fmean.requires_grad or (flgstd.requires_grad and noises is not None):
            eps_all, rng0 = _LSTMRecurrentGPNN2._noise(noises, dev)
            Tn, Bn = T, B
            if noises is None:
                eps_all = torch.zeros(1, E, M, device=dev, dtype=torch.float32)
                Tn, Bn = 1, T * B
            check(lib_.blm_gpnn2_freq_grad(ptr(x), ptr(df), ptr(eps_all), C.byref(rng0) if rng0 is not None else None, ptr(flgstd),
                                           ptr(_grad_buf(fmean)) if fmean.requires_grad else None,
                                           ptr(_grad_buf(flgstd)) if (flgstd.requires_grad and noises is not None) else None,
                                           Tn, Bn, E, M, GP, st), "blm_gpnn2_freq_grad")
            _notify(fmean, flgstd)
        return dx, dcw, dcb, None, None, None, None


def gpnn2_steps(x, coef_w, coef_b, fmean, flgstd, noises, acts):
    """(T,B,E) -> (T,B,NO): GPNN2_t(x[t]) with the noise of calls 0..T-1 (``noises`` as for lstm_recurrent_gpnn2)."""
    return _GPNN2Steps.apply(x, coef_w, coef_b, fmean, flgstd, noises, int(acts))


def lstm_recurrent_gpnn2_supported(H, n_mc):
    return H % 64 == 0 and n_mc < _LSTMRecurrentGPNN2.MP


def lstm_recurrent_gpnn2(xw, h0, c0, w_hh, coef_w, coef_b, fmean, flgstd, noises, gate, acts, mode=0):
    """-> (y (T,B,H), hT, cT).  ``noises``: list of T NoiseSpec (training) or None (mean frequencies); ``acts``: bit set of the
    GPNN2's activations in the mixture's slot order (1 tanh, 2 sigmoid, 4 relu, 8 gelu); ``mode``: see _LSTMRecurrentGPNN2."""
    return _LSTMRecurrentGPNN2.apply(xw, h0, c0, w_hh, coef_w, coef_b, fmean, flgstd, noises, int(gate), int(acts), int(mode))


def lstm_recurrent_gp_supported(H, w_rec):
    return H % 32 == 0 or H >= _PAD_HIDDEN_FROM  # other sizes from 64 up are zero-padded to the next multiple of 32 (below)


def _pad_last_gate_blocks(t, H, Hp):
    """(..., 4H) -> (..., 4Hp), gate block g of the last dimension moved to [g Hp, g Hp + H)."""
    return torch.nn.functional.pad(t.reshape(*t.shape[:-1], 4, H), (0, Hp - H)).reshape(*t.shape[:-1], 4 * Hp)


def lstm_recurrent_gp(xw, h0, c0, w_rec, coef4=None, ovr=-1, rbias=None, w_cell=None):
    """The GP-LSTM recurrence on the fused step kernels (_LSTMRecurrentGP).  A hidden size that is not a multiple of 32 is
    zero-padded as in ``lstm_layer`` (650: 71 k -> the padded rate, tools/lstm_family_hidden_size_probe.py): with zero recurrent
    rows / columns, zero pre-activations and ZERO mixture coefficients a padded unit's GP gate is 0 and its other gates 1/2, so its
    cell and output stay exactly 0 whichever gate the GPNN sits on; pad and slice are torch ops (autograd undoes them)."""
    G = xw.shape[-1]
    H = G // 4
    if H % 32 and H >= _PAD_HIDDEN_FROM and G == 4 * H and tuple(w_rec.shape) == (G, H):
        Hp = (H + 31) // 32 * 32
        pad = torch.nn.functional.pad
        ovr = int(ovr)
        c4 = coef4
        if coef4 is not None and ovr >= 0:
            c4 = _pad_last_gate_blocks(coef4, H, Hp) if coef4.shape[-1] == 4 * H else pad(coef4, (0, Hp - H))
        rb = _pad_last_gate_blocks(rbias, H, Hp) if (rbias is not None and ovr >= 4) else rbias
        wc = pad(w_cell, (0, Hp - H, 0, Hp - H)) if (w_cell is not None and ovr == 5) else w_cell
        y, hT, cT = _LSTMRecurrentGP.apply(_pad_last_gate_blocks(xw, H, Hp), pad(h0, (0, Hp - H)), pad(c0, (0, Hp - H)),
                                           _pad_gate_blocks(w_rec, H, Hp, cols=True), c4, ovr, rb, wc)
        return y[..., :H], hT[..., :H], cT[..., :H]
    return _LSTMRecurrentGP.apply(xw, h0, c0, w_rec, coef4, ovr, rbias, w_cell)


# ----------------------------------------------------------------------------
# Step-wise cells for the GP / Variational LSTMs (the reference runs them as Python time loops,
# model.py:1734-1777, 2503-2531; so does this host code, one fused cell kernel per step)
# ----------------------------------------------------------------------------
class _LSTMCellStep(torch.autograd.Function):
    """(xw, hw, c_prev[, gate_ovr]) -> (h, c).  gates = xw + hw (biases inside), order i,f,g,o; gate
    ``gate_idx`` may take an externally computed activation (GPNN output)."""

    @staticmethod
    def forward(ctx, xw, hw, c_prev, gate_ovr, gate_idx):
        xw, hw, c_prev = _f32(xw, "xw"), _f32(hw, "hw"), _f32(c_prev, "c_prev")
        B, G = xw.shape
        H = G // 4
        h = torch.empty(B, H, device=xw.device, dtype=torch.float32)
        c = torch.empty_like(h)
        ga = torch.empty(B, G, device=xw.device, dtype=torch.float32)
        L.require_gfx950()
        if gate_ovr is None:
            check(lib().blm_lstm_cell_fwd(ptr(xw), ptr(hw), ptr(c_prev), ptr(h), ptr(c), ptr(ga), B, H, stream()),
                  "blm_lstm_cell_fwd")
        else:
            gate_ovr = _f32(gate_ovr, "gate_ovr")
            check(lib().blm_lstm_cell_ovr_fwd(ptr(xw), ptr(hw), ptr(c_prev), ptr(gate_ovr), int(gate_idx), ptr(h), ptr(c),
                                              ptr(ga), B, H, stream()), "blm_lstm_cell_ovr_fwd")
        ctx.save_for_backward(c_prev, c, ga)
        ctx.meta = (gate_ovr is not None, int(gate_idx), B, H)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        c_prev, c, ga = ctx.saved_tensors
        has_ovr, gidx, B, H = ctx.meta
        dev = c.device
        dh = torch.zeros(B, H, device=dev) if dh is None else _f32(dh, "dh")
        dc = None if dc is None else _f32(dc, "dc")
        dgates = torch.empty(B, 4 * H, device=dev, dtype=torch.float32)
        dc_prev = torch.empty(B, H, device=dev, dtype=torch.float32)
        d_ovr = None
        if has_ovr:
            d_ovr = torch.empty(B, H, device=dev, dtype=torch.float32)
            check(lib().blm_lstm_cell_ovr_bwd(ptr(dh), ptr(dc), ptr(c_prev), ptr(c), ptr(ga), gidx, ptr(dgates), ptr(d_ovr),
                                              ptr(dc_prev), B, H, stream()), "blm_lstm_cell_ovr_bwd")
        else:
            check(lib().blm_lstm_cell_bwd(ptr(dh), ptr(dc), ptr(c_prev), ptr(c), ptr(ga), ptr(dgates), ptr(dc_prev), B, H,
                                          stream()), "blm_lstm_cell_bwd")
        return dgates, dgates, dc_prev, d_ovr, None


def lstm_cell(xw, hw, c_prev, gate_ovr=None, gate_idx=-1):
    return _LSTMCellStep.apply(xw, hw, c_prev, gate_ovr, gate_idx)


class _GPMix(torch.autograd.Function):
    """out = sum_i act_i(z) * coef4[i]  with the fixed slot order tanh, sigmoid, relu, gelu."""

    @staticmethod
    def forward(ctx, z, coef4):
        z, coef4 = _f32(z, "z"), _f32(coef4, "coef4")
        N = z.shape[-1]
        M = z.numel() // N
        out = torch.empty_like(z)
        L.require_gfx950()
        check(lib().blm_gp_mix_fwd(ptr(z), ptr(coef4), ptr(out), M, N, stream()), "blm_gp_mix_fwd")
        ctx.save_for_backward(z, coef4)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, coef4 = ctx.saved_tensors
        dout = _f32(dout, "dout")
        N = z.shape[-1]
        M = z.numel() // N
        dz = torch.empty_like(z)
        check(lib().blm_gp_mix_bwd(ptr(dout), ptr(z), ptr(coef4), ptr(dz), M, N, stream()), "blm_gp_mix_bwd")
        dcoef = torch.zeros_like(coef4)
        check(lib().blm_gp_coef_grad(ptr(dout), ptr(z), ptr(dcoef), M, N, stream()), "blm_gp_coef_grad")
        return dz, dcoef


def gp_mix(z, coef4):
    return _GPMix.apply(z, coef4)


class _AddRowVec(torch.autograd.Function):
    """h (B,H) + v (H,) broadcast over rows (VNN noise on the hidden state, model.py:2571-2577)."""

    @staticmethod
    def forward(ctx, h, v):
        out = _f32(h, "h").clone()
        v = _f32(v, "v")
        B, H = out.shape
        L.require_gfx950()
        check(lib().blm_add_rowvec(ptr(out), ptr(v), B, H, stream()), "blm_add_rowvec")
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32(g, "g")
        B, H = g.shape
        dv = torch.empty(H, device=g.device, dtype=torch.float32)
        _colsum_into(g, B, H, dv, accumulate=False)
        return g, dv


def add_rowvec(h, v):
    return _AddRowVec.apply(h, v)


# ----------------------------------------------------------------------------
# optimiser:  clip_grad_norm_ + SGD(momentum)   (train.py:419-420,466)
# ----------------------------------------------------------------------------
class PtrTable:
    """Device arrays of pointers/sizes for the multi-tensor kernels (built once per tensor list)."""

    def __init__(self, params, grads, bufs):
        dev = params[0].device
        self.n = len(params)
        self.params = torch.tensor([p.data_ptr() for p in params], dtype=torch.int64, device=dev)
        self.grads = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64, device=dev)
        self.bufs = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device=dev)
        self.sizes = torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws = torch.zeros(int(lib().blm_sqnorm_ws_floats(self.n)), dtype=torch.float32, device=dev)
        self._keep = (params, grads, bufs)


def clip_sgd(table, clip, lr, momentum, first, grad_scale=1.0, weight_decay=0.0):
    """-> device scalar holding the squared global gradient norm (before grad_scale).  ``weight_decay``:
    torch.optim.SGD's L2 term, added after the clip (train_search_bayes.py:391-392)."""
    L.require_gfx950()
    table.sq.zero_()
    st = stream()
    check(lib().blm_sqnorm_multi(ptr(table.grads), ptr(table.sizes), table.n, ptr(table.sq), ptr(table.ws), st),
          "blm_sqnorm_multi")
    if weight_decay:
        check(lib().blm_clip_sgd_multi_wd(ptr(table.params), ptr(table.grads), ptr(table.bufs), ptr(table.sizes), table.n,
                                          ptr(table.sq), float(clip), float(lr), float(momentum), 1 if first else 0,
                                          float(grad_scale), float(weight_decay), st), "blm_clip_sgd_multi_wd")
    else:
        check(lib().blm_clip_sgd_multi(ptr(table.params), ptr(table.grads), ptr(table.bufs), ptr(table.sizes), table.n,
                                       ptr(table.sq), float(clip), float(lr), float(momentum), 1 if first else 0,
                                       float(grad_scale), st), "blm_clip_sgd_multi")
    return table.sq


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """In-place torch.optim.Adam update of one tensor (architect.py:33); ``step`` counts from 1."""
    L.require_gfx950()
    check(lib().blm_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                              float(eps), float(weight_decay), int(step), stream()), "blm_adam_step")


# ----------------------------------------------------------------------------
# architecture search (model_search_bayes.py): branch mixes with gradients for the mixing weights
# ----------------------------------------------------------------------------
def _reduce_partials(partial, k):
    """(n, k) per-block partial sums -> (k,) on the device."""
    out = torch.empty(k, device=partial.device, dtype=torch.float32)
    _colsum_into(partial, partial.numel() // k, k, out, accumulate=False)
    return out


def _drop_args(drop, B):
    if drop is not None and drop.on:
        return float(drop.p), C.byref(drop.rng()), int(drop.col_offset), int(drop.global_cols or B)
    return 0.0, None, 0, 0


class _Mix2(torch.autograd.Function):
    """out = (p[0] a + p[1] b) * keep over (rows, B, N) activations, probs (2,) on the device
    (BayesTransSearchEncoderLayer, model_search_bayes.py:77-78)."""

    @staticmethod
    def forward(ctx, a, b, probs, drop):
        a, b, probs = _f32(a, "a"), _f32(b, "b"), _f32(probs, "probs")
        N, B = a.shape[-1], a.shape[-2]
        rows = a.numel() // (B * N)
        out = torch.empty_like(a)
        dp, drng, dco, dgc = _drop_args(drop, B)
        L.require_gfx950()
        check(lib().blm_mix2_fwd(ptr(a), ptr(b), ptr(probs), ptr(out), rows, B, N, dp, drng, dco, dgc, stream()), "blm_mix2_fwd")
        ctx.save_for_backward(a, b, probs)
        ctx.meta = (drop, rows, B, N)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, probs = ctx.saved_tensors
        drop, rows, B, N = ctx.meta
        dout = _f32(dout, "dout")
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        partial = torch.empty(int(lib().blm_mix2_partials(rows, B, N)), device=a.device, dtype=torch.float32)
        dp, drng, dco, dgc = _drop_args(drop, B)
        check(lib().blm_mix2_bwd(ptr(dout), ptr(a), ptr(b), ptr(probs), None, ptr(da), ptr(db), ptr(partial), rows, B, N, dp,
                                 drng, dco, dgc, stream()), "blm_mix2_bwd")
        dprobs = _reduce_partials(partial, 2) if ctx.needs_input_grad[2] else None
        return da, db, dprobs, None


def mix2(a, b, probs, drop=NO_DROP):
    return _Mix2.apply(a, b, probs, drop)


class _SearchFFN(torch.autograd.Function):
    """y = lin2(drop(p[0] * GELU(x W1^T + b1) + p[1] * sum_i act_i(x Wg^T + bg) coef[i]))
    GaussTransSearchEncoderLayer FFN (model_search_bayes.py:234-236).  ``probs`` (2,) lives on the device and
    gets its gradient; wg/bg/coef may be leaf parameters or sampled (non-leaf) tensors.  The weight-gradient
    GEMMs are skipped for tensors that do not require grad (the architect step differentiates w.r.t. the
    architecture logits only, architect.py:66-75)."""

    @staticmethod
    def forward(ctx, x, w1, b1, wg, bg, coef, probs, w2, b2, drop):
        x, probs = _f32(x, "x"), _f32(probs, "probs")
        F_, D = w1.shape
        N2 = w2.shape[0]
        M = x.numel() // D
        B = x.shape[-2]
        rows = M // B
        dev = x.device
        need_bwd = any(ctx.needs_input_grad)
        a1 = torch.empty(M, F_, device=dev, dtype=torch.float32) if need_bwd else None
        zg = torch.empty(M, F_, device=dev, dtype=torch.float32) if need_bwd else None
        h1 = torch.empty(M, F_, device=dev, dtype=torch.float32)
        hg = torch.empty(M, F_, device=dev, dtype=torch.float32)
        gemm(L.GEMM_NT, x, w1, h1, M, F_, D, D, D, F_, epilogue=L.EPI_BIAS_GELU, bias=b1, aux=a1)
        gemm(L.GEMM_NT, x, wg, hg, M, F_, D, D, D, F_, epilogue=L.EPI_GP_MIX, bias=bg, aux=zg, coef=coef)
        s = torch.empty(M, F_, device=dev, dtype=torch.float32)
        dp, drng, dco, dgc = _drop_args(drop, B)
        check(lib().blm_mix2_fwd(ptr(h1), ptr(hg), ptr(probs), ptr(s), rows, B, F_, dp, drng, dco, dgc, stream()),
              "blm_mix2_fwd")
        y = torch.empty(*x.shape[:-1], N2, device=dev, dtype=torch.float32)
        gemm(L.GEMM_NT, s, w2, y, M, N2, F_, F_, F_, N2, epilogue=L.EPI_BIAS, bias=b2)
        ctx.save_for_backward(x, a1, h1, zg, hg, s, probs)
        ctx.p = (w1, b1, wg, bg, coef, w2, b2, drop, B)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a1, h1, zg, hg, s, probs = ctx.saved_tensors
        w1, b1, wg, bg, coef, w2, b2, drop, B = ctx.p
        dy = _f32(dy, "dy")
        F_, D = w1.shape
        N2 = w2.shape[0]
        M = x.numel() // D
        rows = M // B
        dev = x.device
        st = stream()
        ds = torch.empty(M, F_, device=dev, dtype=torch.float32)
        gemm(L.GEMM_NN, dy, w2, ds, M, F_, N2, N2, F_, F_)
        partial = torch.empty(int(lib().blm_mix2_partials(rows, B, F_)), device=dev, dtype=torch.float32)
        dz1 = torch.empty(M, F_, device=dev, dtype=torch.float32)
        dzg = torch.empty(M, F_, device=dev, dtype=torch.float32)
        dp, drng, dco, dgc = _drop_args(drop, B)
        # one pass: dz1 = p0 g GELU'(z1), dzg = p1 g mixture'(zg), the two mixing-weight partials, and -- only when
        # the coefficients want their gradient -- dhg = p1 g for blm_gp_coef_grad
        dhg = torch.empty(M, F_, device=dev, dtype=torch.float32) if coef.requires_grad else None
        check(lib().blm_mix2_gp_bwd(ptr(ds), ptr(h1), ptr(hg), ptr(probs), ptr(a1), ptr(zg), ptr(coef), ptr(dz1), ptr(dzg),
                                    ptr(dhg), ptr(partial), rows, B, F_, dp, drng, dco, dgc, st), "blm_mix2_gp_bwd")
        dprobs = _reduce_partials(partial, 2) if ctx.needs_input_grad[6] else None
        dcoef = None
        if coef.requires_grad:
            buf, _, dcoef = _wgrad_target(coef)
            if dcoef is not None:
                buf.zero_()
            check(lib().blm_gp_coef_grad(ptr(dhg), ptr(zg), ptr(buf), M, F_, st), "blm_gp_coef_grad")
            del dhg
        if w2.requires_grad:
            gemm(L.GEMM_TN, dy, s, _grad_buf(w2), N2, F_, M, N2, F_, F_, accumulate=True,
                 colsum_a=_grad_buf(b2) if b2.requires_grad else None)
        elif b2.requires_grad:
            _colsum_into(dy, M, N2, _grad_buf(b2))
        if w1.requires_grad:
            gemm(L.GEMM_TN, dz1, x, _grad_buf(w1), F_, D, M, F_, D, D, accumulate=True,
                 colsum_a=_grad_buf(b1) if b1.requires_grad else None)
        elif b1.requires_grad:
            _colsum_into(dz1, M, F_, _grad_buf(b1))
        dwg = dbg = None
        if wg.requires_grad:
            buf, acc, dwg = _wgrad_target(wg)
            gemm(L.GEMM_TN, dzg, x, buf, F_, D, M, F_, D, D, accumulate=acc)
        if bg.requires_grad:
            buf, acc, dbg = _wgrad_target(bg)
            _colsum_into(dzg, M, F_, buf, accumulate=acc)
        _notify(w2, b2, w1, b1, wg, bg, coef)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(L.GEMM_NN, dz1, w1, dx, M, D, F_, F_, D, D)
            gemm(L.GEMM_NN, dzg, wg, dx, M, D, F_, F_, D, D, accumulate=True)
        return dx, None, None, dwg, dbg, dcoef, dprobs, None, None, None


def search_ffn(x, w1, b1, wg, bg, coef, probs, w2, b2, drop=NO_DROP):
    return _SearchFFN.apply(x, w1, b1, wg, bg, coef, probs, w2, b2, drop)


class _LSTMSearchLayer(torch.autograd.Function):
    """One BayesLSTMSearchCell over a window (model_search_bayes.py:661-710).  w8_ih (8H,I), w8_hh (8H,H),
    bias8 (8H): the standard gate block [i f g o] stacked on the four `Bayes` gate maps; probs (4,2) on the
    device.  Per step: one (B,8H) recurrent GEMM + the pointwise search cell; the input products of all
    steps, both weight gradients and dx are single GEMMs over the whole window."""

    @staticmethod
    def forward(ctx, x, h0, c0, w8_ih, w8_hh, bias8, probs):
        x, h0, c0, probs = _f32(x, "x"), _f32(h0, "h0"), _f32(c0, "c0"), _f32(probs, "probs")
        w8_ih, w8_hh, bias8 = _f32(w8_ih, "w8_ih"), _f32(w8_hh, "w8_hh"), _f32(bias8, "bias8")
        T, B, I = x.shape
        H = h0.shape[-1]
        dev = x.device
        st = stream()
        need_bwd = any(ctx.needs_input_grad)
        xw = torch.empty(T, B, 8 * H, device=dev, dtype=torch.float32)
        gemm(L.GEMM_NT, x, w8_ih, xw, T * B, 8 * H, I, I, I, 8 * H, epilogue=L.EPI_BIAS, bias=bias8)
        hs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        cs = torch.empty(T + 1, B, H, device=dev, dtype=torch.float32)
        hs[0].copy_(h0)
        cs[0].copy_(c0)
        acts = torch.empty(T, B, 8 * H, device=dev, dtype=torch.float32) if need_bwd else None
        if H % 32 == 0:  # one launch per step: recurrent product over the stacked weight + the search cell
            xw_p, hs_p, cs_p, w_p, pr_p = _P(xw), _P(hs), _P(cs), ptr(w8_hh), ptr(probs)
            ac_p = _P(acts) if need_bwd else None
            step_fwd = lib().blm_lstm_search_step_fwd
            for t in range(T):
                check(step_fwd(xw_p[t], w_p, hs_p[t], cs_p[t], pr_p, hs_p[t + 1], cs_p[t + 1], ac_p[t] if need_bwd else None, B, H,
                               st), "blm_lstm_search_step_fwd")
        else:
            hw = torch.empty(B, 8 * H, device=dev, dtype=torch.float32)
            for t in range(T):
                gemm(L.GEMM_NT, hs[t], w8_hh, hw, B, 8 * H, H, H, H, 8 * H)
                check(lib().blm_lstm_search_cell_fwd(ptr(xw[t]), ptr(hw), ptr(cs[t]), ptr(probs), ptr(hs[t + 1]),
                                                     ptr(cs[t + 1]), ptr(acts[t]) if need_bwd else None, B, H, st),
                      "blm_lstm_search_cell_fwd")
        ctx.save_for_backward(x, hs, cs, acts, w8_ih, w8_hh, probs)
        ctx.dims = (T, B, I, H)
        return hs[1:], hs[T], cs[T]

    @staticmethod
    def backward(ctx, dy, dhT, dcT):
        x, hs, cs, acts, w8_ih, w8_hh, probs = ctx.saved_tensors
        T, B, I, H = ctx.dims
        dev = x.device
        st = stream()
        dy = _f32(dy.contiguous(), "dy")
        npart = int(lib().blm_lstm_search_cell_partials(B, H))
        part = torch.empty(T, npart, device=dev, dtype=torch.float32)
        dz = torch.empty(T, B, 8 * H, device=dev, dtype=torch.float32)
        dcb = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]
        dhr = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]
        dh_rec = _f32(dhT.contiguous(), "dhT") if dhT is not None else None
        dc = _f32(dcT.contiguous(), "dcT") if dcT is not None else None
        # H % 16 == 0: one launch per step -- the skinny recurrent dgrad dh = dz8 . W8 on the LSTM step kernel (fixed
        # summation order, no split-K memset + atomics) against the weight transposed once per window, with the
        # search cell's backward of the previous step fused behind it; otherwise cell kernel + blm_gemm per step
        fused = H % 16 == 0
        part2 = None
        if fused:
            w_t = torch.empty(H, 8 * H, device=dev, dtype=torch.float32)
            check(lib().blm_transpose(ptr(w8_hh), ptr(w_t), 8 * H, H, st), "blm_transpose")
            part = torch.empty(1, npart, device=dev, dtype=torch.float32)
            check(lib().blm_lstm_search_cell_bwd(ptr(dy[T - 1]), ptr(dh_rec), ptr(dc), ptr(cs[T - 1]), ptr(cs[T]),
                                                 ptr(acts[T - 1]), ptr(probs), ptr(dz[T - 1]), ptr(dcb[0]), ptr(part[0]), B, H, st),
                  "blm_lstm_search_cell_bwd")
            nstep = int(lib().blm_lstm_search_step_partials(B, H))
            part2 = torch.empty(max(T - 1, 1), nstep, device=dev, dtype=torch.float32)
            k = 0
            dz_p, dy_p, cs_p, ac_p, p2_p = _P(dz), _P(dy), _P(cs), _P(acts), _P(part2)
            dcb_p, wt_p, pr_p = (ptr(dcb[0]), ptr(dcb[1])), ptr(w_t), ptr(probs)
            step_bwd = lib().blm_lstm_search_step_bwd
            for t in range(T - 1, 0, -1):
                check(step_bwd(dz_p[t], wt_p, dy_p[t - 1], dcb_p[k], cs_p[t - 1], cs_p[t], ac_p[t - 1], pr_p, dz_p[t - 1],
                               dcb_p[k ^ 1], p2_p[t - 1], B, H, st), "blm_lstm_search_step_bwd")
                k ^= 1
            check(lib().blm_lstm_step_dh(ptr(dz[0]), ptr(w_t), ptr(dhr[0]), B, H, 8 * H, st), "blm_lstm_step_dh")
            dh_rec, dc = dhr[0], dcb[k]
            if T == 1:
                part2 = None
        else:
            for t in range(T - 1, -1, -1):
                k = t & 1
                check(lib().blm_lstm_search_cell_bwd(ptr(dy[t]), ptr(dh_rec), ptr(dc), ptr(cs[t]), ptr(cs[t + 1]), ptr(acts[t]),
                                                     ptr(probs), ptr(dz[t]), ptr(dcb[k]), ptr(part[t]), B, H, st),
                      "blm_lstm_search_cell_bwd")
                dc = dcb[k]
                gemm(L.GEMM_NN, dz[t], w8_hh, dhr[k], B, H, 8 * H, 8 * H, H, H)
                dh_rec = dhr[k]
        dprobs = None
        if ctx.needs_input_grad[6]:
            dprobs = _reduce_partials(part, 8)
            if part2 is not None:
                dprobs = dprobs + _reduce_partials(part2, 8)
            dprobs = dprobs.view(4, 2)
        dw_ih = dw_hh = db = dx = None
        M = T * B
        if ctx.needs_input_grad[4]:
            dw_hh = torch.empty_like(w8_hh)
            gemm(L.GEMM_TN, dz, hs, dw_hh, 8 * H, H, M, 8 * H, H, H)
        if ctx.needs_input_grad[3]:
            dw_ih = torch.empty_like(w8_ih)
            db = torch.empty(8 * H, device=dev, dtype=torch.float32) if ctx.needs_input_grad[5] else None
            if db is not None:
                db.zero_()
            gemm(L.GEMM_TN, dz, x, dw_ih, 8 * H, I, M, 8 * H, I, I, colsum_a=db)
        elif ctx.needs_input_grad[5]:
            db = torch.empty(8 * H, device=dev, dtype=torch.float32)
            _colsum_into(dz, M, 8 * H, db, accumulate=False)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(L.GEMM_NN, dz, w8_ih, dx, M, I, 8 * H, 8 * H, I, I)
        return dx, dh_rec, dc, dw_ih, dw_hh, db, dprobs


def lstm_search_layer(x, h0, c0, w8_ih, w8_hh, bias8, probs):
    return _LSTMSearchLayer.apply(x, h0, c0, w8_ih, w8_hh, bias8, probs)

"""ctypes binding of libbayeslm_hip.so (include/bayeslm.h).

The product path has no CPU fallback: if the library is missing, was built for
another ABI, or a call fails, this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLM_LIB=/path/to/another/libbayeslm_hip.so selects a different BUILD of the same library (same-box A/B runs of
# tools/ab_lib.sh) without touching the in-tree file; it is still the HIP library or nothing -- there is no fallback
LIB_PATH = os.environ.get("BLM_LIB") or os.path.join(_HERE, "libbayeslm_hip.so")

ABI_VERSION = 1
OK = 0
ERR_INVALID, ERR_ABI, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4  # blm_status (include/bayeslm.h)
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_MUL_DGELU, EPI_BAYES_WGRAD, EPI_GP_MIX, EPI_MUL_DGP_MIX = range(7)
GEMM_ACCUMULATE = 1
STREAM_WEIGHT = 0x10000000   # stream class in the top 4 bits, tensor / site id in the low 28: the classes cannot alias
STREAM_DROPOUT = 0x20000000

c_fp = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("stream", C.c_uint32), ("step", C.c_uint32)]


class Variational(C.Structure):
    _fields_ = [("lgstd", C.c_void_p), ("eps", C.c_void_p), ("row_lo", C.c_int32), ("srows", C.c_int32),
                ("rng", Rng)]


class VarItem(C.Structure):
    """blm_var_item: one tensor of a variational group (include/bayeslm.h)."""
    _fields_ = [("mu", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("v", Variational), ("w_out", C.c_void_p),
                ("kl_weight", C.c_float), ("kl_minus", C.c_float), ("dw", C.c_void_p), ("dmu", C.c_void_p),
                ("dlgstd", C.c_void_p)]


VAR_GROUP_MAX = 16


class GemmArgs(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("op", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("alpha", C.c_float), ("flags", C.c_uint32),
                ("epilogue", C.c_int32), ("bias", C.c_void_p), ("aux", C.c_void_p), ("coef", C.c_void_p),
                ("var_b", Variational), ("C2", C.c_void_p), ("wg_mu", C.c_void_p), ("var_c", Variational),
                ("kl_lambda", C.c_float), ("kl_inv_n", C.c_float), ("drop_p", C.c_float), ("drop_rng", Rng),
                ("drop_B", C.c_int32), ("drop_col_offset", C.c_int32), ("drop_global_cols", C.c_int32),
                ("colsum_a", C.c_void_p)]


class Gpnn2Seq(C.Structure):
    """blm_gpnn2_seq (include/bayeslm.h)."""
    _fields_ = ([("abi_version", C.c_uint32), ("mode", C.c_int32), ("gate", C.c_int32), ("acts", C.c_int32)]
                + [(n, C.c_int32) for n in ("T", "B", "H", "M", "MP", "GP", "nF")]
                + [(n, C.c_void_p) for n in ("xw", "w_hh", "w_hh_t", "FT", "Fp", "cwp", "cwt", "hs", "cs", "ga", "z4", "pre", "feat",
                                             "sact", "gout", "dy", "dh", "dcs2", "dgates", "da", "ds", "df")])


class GemmPlan(C.Structure):
    _fields_ = [("tile", C.c_int32), ("splits", C.c_int32), ("source", C.c_int32), ("model_us", C.c_float)]


# name -> (restype, argtypes); every symbol include/bayeslm.h declares
_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_rngp, _varp = C.POINTER(Rng), C.POINTER(Variational)
SIGNATURES = {
    "blm_abi_version": (C.c_uint32, []),
    "blm_last_error": (C.c_char_p, []),
    "blm_query": (_i, [_i, C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "blm_sample_weight": (_i, [_vp, _i64, _i64, _varp, _vp, _vp, _f, _vp]),
    "blm_sample_weight_bwd": (_i, [_vp, _i64, _i64, _varp, _vp, _vp, _vp]),
    "blm_variational_group_fwd": (_i, [C.POINTER(VarItem), _i, _vp, _vp]),
    "blm_variational_group_bwd": (_i, [C.POINTER(VarItem), _i, _vp, _vp]),
    "blm_philox_normal": (_i, [_vp, _i64, _rngp, _vp]),
    "blm_kl_mean_fwd": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _f, _vp, _vp]),
    "blm_kl_mean_bwd": (_i, [_vp, _i64, _vp, _i64, _i64, _vp, _f, _vp, _i64, _vp, _vp]),
    "blm_gemm": (_i, [C.POINTER(GemmArgs), _vp]),
    "blm_set_gemm_mode": (_i, [_i]),
    "blm_get_gemm_mode": (_i, []),
    "blm_gemm_plan_query": (_i, [C.POINTER(GemmArgs), C.POINTER(GemmPlan)]),
    "blm_gemm_plan_launch": (_i, [C.POINTER(GemmArgs), C.POINTER(GemmPlan)]),
    "blm_gemm_plan_model_us": (_i, [C.POINTER(GemmArgs), _i, _i, C.POINTER(C.c_float)]),
    "blm_gemm_plan_override": (_i, [_i, _i]),
    "blm_gemm_plan_set": (_i, [_i] * 8),
    "blm_gemm_plan_clear": (_i, [_i]),
    "blm_mfma_probe_ws_floats": (_i64, []),
    "blm_mfma_probe": (_i, [_vp, _i, C.POINTER(C.c_double), _vp]),
    "blm_set_option": (_i, [C.c_char_p, _i]),
    "blm_get_option": (_i, [C.c_char_p, C.POINTER(C.c_int)]),
    "blm_gemm_plan_set_cus": (_i, [_i]),
    "blm_gemm_plan_get_cus": (_i, []),
    "blm_gemm_plan_comm_window": (_i, [_f]),
    "blm_gemm_plan_comm_window_left": (_f, []),
    "blm_gemm_plan_set_comm": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "blm_embed_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _f, _f, _rngp, _i, _i, _vp]),
    "blm_embed_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _f, _f, _rngp, _i, _i, _vp]),
    "blm_add_pe_dropout": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_dropout": (_i, [_vp, _vp, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_dropout_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_add_dropout_ln_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _rngp, _i, _i, _vp]),
    "blm_ln_bwd_ws_floats": (_i64, [_i, _i]),
    "blm_add_dropout_ln_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_attn_fwd": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_attn_fwd_rows": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp]),
    "blm_attn_bwd": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_attn_bwd_ws": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _rngp, _i, _i, _vp,
                            _i64, _vp]),
    "blm_attn_bwd_ws_floats": (_i64, [_i, _i, _i, _i]),
    "blm_attn_fwd_keep": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "blm_attn_bwd_keep": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "blm_ce_fwd_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "blm_ce_interp_fwd": (_i, [_vp, _vp, _i64, _f, _vp, _vp, _i, _i, _vp]),
    "blm_linear_nll_ws_floats": (_i64, [_i, _i]),
    "blm_linear_nll": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "blm_linear_nll2_wcat_floats": (_i64, [_i, _i, _i]),
    "blm_linear_nll2_ws_floats": (_i64, [_i, _i, _i, _i]),
    "blm_linear_nll2": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _vp, _i64, _vp, _i64, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "blm_ce_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "blm_gp_coef_grad": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "blm_colsum": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp]),
    "blm_colsum2": (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _vp]),
    "blm_init_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "blm_sqnorm_ws_floats": (_i64, [_i]),
    "blm_sqnorm_multi": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "blm_clip_sgd_multi": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _i, _f, _vp]),
    "blm_lstm_step_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_step_bwd": (_i, [_vp] * 10 + [_i, _i, _vp]),
    "blm_lstm_seq_fwd": (_i, [_vp] * 6 + [_i, _i, _i, _vp]),
    "blm_lstm_seq_bwd": (_i, [_vp] * 7 + [_i, _vp, _i, _i, _i, _i, _i, _vp]),
    "blm_lstm_seq_pair_fwd": (_i, [_vp] * 5 + [_i] + [_vp] * 5 + [_i, _i, _i, _vp]),
    "blm_lstm_step_fwd_gp": (_i, [_vp] * 8 + [_i, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_step_bwd_gp": (_i, [_vp] * 10 + [_i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_transpose": (_i, [_vp, _vp, _i, _i, _vp]),
    "blm_lstm_cell_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_cell_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_cell_bwd2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_cell_ovr_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_lstm_cell_ovr_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_gp_mix_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "blm_gp_mix_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_add_rowvec": (_i, [_vp, _vp, _i, _i, _vp]),
    "blm_axpy": (_i, [_vp, _vp, _i64, _f, _vp]),
    "blm_rows_gather_add": (_i, [_vp, _vp, _vp, _i64, _i, _i64, _vp]),
    "blm_mix2_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_mix2_partials": (_i64, [_i, _i, _i]),
    "blm_mix2_bwd": (_i, [_vp] * 8 + [_i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_mix2_gp_bwd": (_i, [_vp] * 11 + [_i, _i, _i, _f, _rngp, _i, _i, _vp]),
    "blm_lstm_search_cell_fwd": (_i, [_vp] * 7 + [_i, _i, _vp]),
    "blm_lstm_search_cell_partials": (_i64, [_i, _i]),
    "blm_lstm_search_cell_bwd": (_i, [_vp] * 10 + [_i, _i, _vp]),
    "blm_lstm_search_step_fwd": (_i, [_vp] * 8 + [_i, _i, _vp]),
    "blm_lstm_search_step_partials": (_i64, [_i, _i]),
    "blm_lstm_search_step_bwd": (_i, [_vp] * 11 + [_i, _i, _vp]),
    "blm_lstm_step_dh": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "blm_lstm_step_dh_ld": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "blm_lstm_step_dh_act": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _i, _f, _i, _vp]),
    "blm_gpnn2_actsum_fwd": (_i, [_vp, _vp, _i64, _i, _i, _i, _f, _i, _vp]),
    "blm_gpnn2_actsum_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _f, _i, _vp]),
    "blm_add_cols": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _vp]),
    "blm_gpnn2_sample_steps": (_i, [_vp, _vp, _vp, _rngp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "blm_gpnn2_freq_grad": (_i, [_vp, _vp, _vp, _rngp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "blm_lstm_gpnn2_seq_fwd": (_i, [C.POINTER(Gpnn2Seq), _vp]),
    "blm_lstm_gpnn2_seq_bwd": (_i, [C.POINTER(Gpnn2Seq), _vp]),
    "blm_lstm_cell_ovr_bwd2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "blm_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp]),
    "blm_clip_sgd_multi_wd": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _i, _f, _f, _vp]),
}

_lib = None
ROCTX_RANGES = [0]  # ranges pushed so far (BLM_ROCTX=1)


def _wrap_with_roctx(l):
    """BLM_ROCTX=1: every C-ABI call of this binding runs inside a roctx range named after the entry point
    (roctxRangePushA / roctxRangePop of librocprofiler-sdk-roctx, else libroctx64), so `rocprofv3 --marker-trace --kernel-trace` shows which call each kernel
    dispatch belongs to.  Off by default: a range costs two more foreign calls per entry point."""
    try:  # rocprofv3 (rocprofiler-sdk) intercepts this one; the roctracer library is the fallback for older tools
        rx = C.CDLL("librocprofiler-sdk-roctx.so")
    except OSError:
        rx = C.CDLL("libroctx64.so")
    rx.roctxRangePushA.argtypes, rx.roctxRangePushA.restype = [C.c_char_p], C.c_int
    rx.roctxRangePop.argtypes, rx.roctxRangePop.restype = [], C.c_int
    for name in SIGNATURES:
        fn = getattr(l, name)

        def ranged(*a, _fn=fn, _name=name.encode()):
            rx.roctxRangePushA(_name)
            ROCTX_RANGES[0] += 1
            try:
                return _fn(*a)
            finally:
                rx.roctxRangePop()
        setattr(l, name, ranged)


class BayesLMError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it is not built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BayesLMError(
                "libbayeslm_hip.so is not built (%s). Build it with `make -C bayeslms_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.blm_abi_version() != ABI_VERSION:
            raise BayesLMError("libbayeslm_hip.so ABI %d != binding ABI %d" % (l.blm_abi_version(), ABI_VERSION))
        if os.environ.get("BLM_ROCTX", "0") == "1":
            _wrap_with_roctx(l)
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != OK:
        raise BayesLMError("%s failed (status %d): %s" % (what or "libbayeslm_hip call", rc,
                                                          lib().blm_last_error().decode("utf-8", "replace")))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_index = None


def stream():
    """Raw handle of torch's CURRENT stream on this process's GPU (one process per GPU: the device index is read once).
    ``torch.cuda.current_stream()`` costs ~17 us per call (device-index resolution); the raw query is < 1 us, and an
    op makes this call for every kernel it launches."""
    global _dev_index
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return _raw_stream(_dev_index)


def ptr(t):
    """Device address of a tensor (None -> NULL); plain integers (ops._P row addresses) pass through."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


def dev_tensor(t, name="tensor", dtype=torch.float32):
    """Product-path guard: fp32 (or int64) contiguous tensor on the GPU."""
    if not t.is_cuda:
        raise BayesLMError("%s must live on the GPU: bayeslms_amd has no CPU path" % name)
    if _dev_index is not None and t.device.index != _dev_index:
        raise BayesLMError("%s lives on cuda:%d but this process launches on cuda:%d (one process per GPU: call "
                           "torch.cuda.set_device before the first kernel)" % (name, t.device.index, _dev_index))
    if t.dtype != dtype:
        raise BayesLMError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def rng(seed, stream_id, step):
    return Rng(int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF)


_checked_arch = False


def require_gfx950():
    """Called once per process before the first kernel launch."""
    global _checked_arch
    if _checked_arch:
        return
    if not torch.cuda.is_available():
        raise BayesLMError("no GPU visible: bayeslms_amd runs on MI355X (gfx950) only")
    arch = C.create_string_buffer(32)
    ncu, lds = C.c_int(0), C.c_int(0)
    check(lib().blm_query(torch.cuda.current_device(), arch, C.byref(ncu), C.byref(lds)), "blm_query")
    name = arch.value.decode()
    if not name.startswith("gfx950"):
        raise BayesLMError("libbayeslm_hip.so is built for gfx950 only, device is %s" % name)
    _checked_arch = True

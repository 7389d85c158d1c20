#!/usr/bin/env python3
"""What does the epilogue of a short-K GEMM cost?  The FFN up-projection of the headline step (NT 8192 x 4096 x 512, 64x128
tiles) with the epilogue built up piece by piece: plain store, + bias, + GELU, + the derivative written for backward,
+ the dropout mask (Philox, one block per 4 outputs); and the down-projection's input gradient (NN, * dGELU).
usage: gemm_epi_probe.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    dbg = None
    if "life" in os.path.basename(L.LIB_PATH):  # -DBLM_GEMM_LIFE build: the store-mode switch exists
        import ctypes
        dbg = ctypes.CDLL(L.LIB_PATH)
    for mode in ((0, 1, 2) if dbg else (0,)):
        if dbg:
            assert dbg.blm_debug_store_mode_nt(mode) == 0 and dbg.blm_debug_store_mode_nn(mode) == 0
            print("== store mode %d (%s)" % (mode, ("as shipped", "computed, not stored", "every workgroup stores to tile (0, 0)")[mode]))
        run(dev, reps, (12,) if dbg else tuple(int(t) for t in os.environ.get("EPI_TILES", "12,11,22").split(",")))


def run(dev, reps, tiles):
    g = torch.Generator(device=dev).manual_seed(1)
    M, N, K = 8192, 4096, 512
    x, w = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    wt = torch.randn(K, N, device=dev, generator=g)
    b = torch.randn(N, device=dev, generator=g)
    y, aux = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    drop = ops.Drop(p=0.2, seed=5, site=3, step=9)
    B = 64
    fl = 2.0 * M * N * K
    for tile in tiles:
        L.check(L.lib().blm_gemm_plan_override(tile, 1), "override")
        rows = [
            ("NT plain store", lambda: ops.gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, N)),
            ("NT + bias", lambda: ops.gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, N, epilogue=L.EPI_BIAS, bias=b)),
            ("NT + bias + GELU", lambda: ops.gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, N, epilogue=L.EPI_BIAS_GELU, bias=b)),
            ("NT + bias + GELU + derivative", lambda: ops.gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, N, epilogue=L.EPI_BIAS_GELU, bias=b, aux=aux)),
            ("NT + bias + GELU + derivative + dropout", lambda: ops.gemm(L.GEMM_NT, x, w, y, M, N, K, K, K, N, epilogue=L.EPI_BIAS_GELU, bias=b, aux=aux, drop=drop, drop_B=B)),
            ("NN plain store", lambda: ops.gemm(L.GEMM_NN, x, wt, y, M, N, K, K, N, N)),
            ("NN * dGELU", lambda: ops.gemm(L.GEMM_NN, x, wt, y, M, N, K, K, N, N, epilogue=L.EPI_MUL_DGELU, aux=aux)),
        ]
        for name, fn in rows:
            us = timed(fn, reps)
            print("tile %d  %-42s %7.1f us  %5.1f TF/s" % (tile, name, us, fl / us / 1e6), flush=True)
    L.check(L.lib().blm_gemm_plan_override(0, 0), "override")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the shapes of BASELINE.json configs[2] (one process per
tile override: BLM_GEMM_TILE=11|12|21|22, BLM_GEMM_SPLITK=n).  Prints TFLOP/s per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402

M = 8192
SHAPES = [  # (name, op, M, N, K)
    ("lin2_fwd   NT", L.GEMM_NT, M, 512, 4096), ("lin1_fwd   NT", L.GEMM_NT, M, 4096, 512),
    ("qkv_fwd    NT", L.GEMM_NT, M, 1536, 512), ("o_fwd      NT", L.GEMM_NT, M, 512, 512),
    ("dec_fwd    NT", L.GEMM_NT, M, 33000, 512),
    ("lin2_dgrad NN", L.GEMM_NN, M, 4096, 512), ("lin1_dgrad NN", L.GEMM_NN, M, 512, 4096),
    ("dec_dgrad  NN", L.GEMM_NN, M, 512, 33000), ("o_dgrad    NN", L.GEMM_NN, M, 512, 512),
    ("lin2_wgrad TN", L.GEMM_TN, 512, 4096, M), ("lin1_wgrad TN", L.GEMM_TN, 4096, 512, M),
    ("qkv_wgrad  TN", L.GEMM_TN, 1536, 512, M), ("o_wgrad    TN", L.GEMM_TN, 512, 512, M),
    ("dec_wgrad  TN", L.GEMM_TN, 33000, 512, M),
]
M2 = 2240  # BASELINE configs[1] (LSTM 2x1024, T 35 x B 64): CFG=2 selects these
SHAPES2 = [
    ("c2dec_fwd   NT", L.GEMM_NT, M2, 33000, 1024), ("c2dec_dgrad NN", L.GEMM_NN, M2, 1024, 33000),
    ("c2dec_wgrad TN", L.GEMM_TN, 33000, 1024, M2), ("c2in_fwd    NT", L.GEMM_NT, M2, 4096, 1024),
    ("c2in_dgrad  NN", L.GEMM_NN, M2, 1024, 4096), ("c2in_wgrad  TN", L.GEMM_TN, 4096, 1024, M2),
]
if os.environ.get("CFG") == "2":
    SHAPES = SHAPES2


def main():
    dev = torch.device("cuda:0")
    reps = int(os.environ.get("REPS", "20"))
    only = os.environ.get("ONLY")
    print("BLM_GEMM_TILE=%s BLM_GEMM_SPLITK=%s" % (os.environ.get("BLM_GEMM_TILE", "auto"), os.environ.get("BLM_GEMM_SPLITK", "auto")))
    for name, op, m, n, k in SHAPES:
        if only and not any(o in name for o in only.split(",")):
            continue
        if op == L.GEMM_NT:
            A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), k, k
        elif op == L.GEMM_NN:
            A, B, lda, ldb = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), k, n
        else:
            A, B, lda, ldb = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), m, n
        C = torch.zeros(m, n, device=dev)
        acc = op == L.GEMM_TN
        for _ in range(6):
            ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(op, A, B, C, m, n, k, lda, ldb, n, accumulate=acc)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-16s M=%5d N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s" % (name, m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
        del A, B, C


if __name__ == "__main__":
    main()

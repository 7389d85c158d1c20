#!/usr/bin/env python3
"""What the fused epilogue of the FFN's first GEMM costs (8192 x 4096 x 512, NT): plain, + bias, + GELU, + the saved
derivative (second 134 MB output), + dropout.  Same launch each time, only the epilogue differs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m, n, k = 8192, 4096, 512
    A, B = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) * 0.05
    C, Z = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    b = torch.randn(n, device=dev)
    drop = ops.Drop(p=0.2, seed=7, site=3, step=1)
    cases = [("plain", dict()), ("bias", dict(epilogue=L.EPI_BIAS, bias=b)),
             ("bias+gelu", dict(epilogue=L.EPI_BIAS_GELU, bias=b)),
             ("bias+gelu+aux", dict(epilogue=L.EPI_BIAS_GELU, bias=b, aux=Z)),
             ("bias+gelu+drop", dict(epilogue=L.EPI_BIAS_GELU, bias=b, drop=drop, drop_B=64)),
             ("bias+gelu+aux+drop", dict(epilogue=L.EPI_BIAS_GELU, bias=b, aux=Z, drop=drop, drop_B=64))]
    reps = 30
    for name, kw in cases:
        for _ in range(5):
            ops.gemm(L.GEMM_NT, A, B, C, m, n, k, k, k, n, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(L.GEMM_NT, A, B, C, m, n, k, k, k, n, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-20s %7.1f us  %6.1f TFLOP/s" % (name, ms * 1e3, 2.0 * m * n * k / ms / 1e9))


if __name__ == "__main__":
    main()

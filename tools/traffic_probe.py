#!/usr/bin/env python3
"""Launch sequence for the HBM-traffic study of the roofline GEMM (Bayesian FFN linear2 forward, NT 8192 x 512 x K).
Run it under `rocprofv3 --pmc <counters> --kernel-trace` (tools/traffic_probe.sh does, one pass per counter group and
library build); it forces the tile per launch (blm_gemm_plan_override), flushes the L2s between launches and writes the
label of every GEMM dispatch, in order, to $PROBE_LABELS so that tools/traffic_report.py can join them with the
counter rows by dispatch order."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, N = 8192, 512
    labels = []
    flush = torch.empty(96 << 20, device=dev)  # 384 MB: every L2 line and the Infinity Cache turned over
    reps = int(os.environ.get("REPS", "3"))
    for K in (512, 1024, 2048, 4096, 8192):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev)
        Cm = torch.empty(M, N, device=dev)
        for tile in (21, 22, 12, 11, 28):
            L.check(L.lib().blm_gemm_plan_override(tile, 1), "override")
            for r in range(reps):
                flush.fill_(float(r))
                ops.gemm(L.GEMM_NT, A, B, Cm, M, N, K, K, K, N)
                labels.append({"K": K, "tile": tile, "rep": r, "M": M, "N": N})
        torch.cuda.synchronize()
    L.check(L.lib().blm_gemm_plan_override(0, 0), "override")
    out = os.environ.get("PROBE_LABELS")
    if out:
        json.dump(labels, open(out, "w"))


if __name__ == "__main__":
    main()

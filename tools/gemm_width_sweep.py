#!/usr/bin/env python3
"""Does the planner pick well for model widths it has no table entries for?  Stand-alone time of every product of a Transformer layer at
d_model D, ff 4 D (M = 8192 rows: T 128 x B 64), under the planner's own plan and under every forced (tile, K slices) candidate.
usage: gemm_width_sweep.py [D ...]   -> profiles/r05_gemm_width_sweep.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402
from bayeslms_amd._lib import check, lib  # noqa: E402

M = 8192
dev = torch.device("cuda:0")
widths = [int(x) for x in sys.argv[1:]] or [384, 768]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for D in widths:
    F_ = 4 * D
    shapes = [("qkv fwd NT", L.GEMM_NT, M, 3 * D, D), ("o fwd NT", L.GEMM_NT, M, D, D), ("ffn1 fwd NT", L.GEMM_NT, M, F_, D),
              ("ffn2 fwd NT", L.GEMM_NT, M, D, F_), ("ffn1 dgrad NN", L.GEMM_NN, M, D, F_), ("ffn2 dgrad NN", L.GEMM_NN, M, F_, D),
              ("qkv dgrad NN", L.GEMM_NN, M, D, 3 * D), ("ffn1 wgrad TN", L.GEMM_TN, F_, D, M), ("ffn2 wgrad TN", L.GEMM_TN, D, F_, M),
              ("qkv wgrad TN", L.GEMM_TN, 3 * D, D, M), ("o wgrad TN", L.GEMM_TN, D, D, M)]
    tot_auto = tot_best = 0.0
    for name, op, m, n, k in shapes:
        if op == L.GEMM_NT:
            A, Bm, Cc, lda, ldb = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev), k, k
        elif op == L.GEMM_NN:
            A, Bm, Cc, lda, ldb = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev), k, n
        else:
            A, Bm, Cc, lda, ldb = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.zeros(m, n, device=dev), m, n
        acc = op == L.GEMM_TN
        run = lambda: ops.gemm(op, A, Bm, Cc, m, n, k, lda, ldb, n, accumulate=acc)  # noqa: E731
        check(lib().blm_gemm_plan_override(0, 0), "override")
        t_auto = timed(run)
        best = (t_auto, "planner")
        for tile in (11, 12, 21, 22, 28):
            for sp in ((1, 2, 4) if op == L.GEMM_TN else (1,)):
                check(lib().blm_gemm_plan_override(tile, sp), "override")
                try:
                    t = timed(run, 6)
                except Exception:  # noqa: BLE001
                    continue
                if t < best[0]:
                    best = (t, "tile %d x %d slices" % (tile, sp))
        check(lib().blm_gemm_plan_override(0, 0), "override")
        fl = 2.0 * m * n * k
        tot_auto += t_auto
        tot_best += best[0]
        print("D %4d %-14s %5d x %5d x %5d: planner %7.1f us (%.2f of peak), best %7.1f us (%s)%s"
              % (D, name, m, n, k, t_auto, fl / t_auto / 1e6 / 157.3, best[0], best[1], "   <-- %.0f %% left" % (100 * (t_auto / best[0] - 1)) if t_auto > 1.05 * best[0] else ""), flush=True)
    print("D %4d: layer products under the planner %.1f us, best candidates %.1f us (%.1f %%)" % (D, tot_auto, tot_best, 100 * (tot_auto / tot_best - 1)), flush=True)

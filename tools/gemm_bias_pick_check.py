#!/usr/bin/env python3
"""How good is the cost model's pick for bias-epilogue products now that they may take K slices?  Stand-alone timing of every
candidate (5 tiles x slice counts) on a small grid of NT shapes with a bias, against the plan the model chooses (plan table off).
Prints chosen / best time per shape and the worst ratios."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402
import gemm_tune as G  # noqa: E402


def t_us(fn, reps=20):
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


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    L.check(L.lib().blm_gemm_plan_clear(0), "clear")  # cost model only
    rows = []
    for M in (64, 256, 300, 700, 1024, 2560, 3200):
        for N in (512, 1536, 4096):
            for K in (512, 1024, 4096):
                A, B, bv = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g), torch.randn(N, device=dev, generator=g)
                C = torch.empty(M, N, device=dev)
                fn = lambda: ops.gemm(L.GEMM_NT, A, B, C, M, N, K, K, K, N, epilogue=L.EPI_BIAS, bias=bv)  # noqa: E731
                G.override(0, 0)
                chosen = G.query(L.GEMM_NT, M, N, K, L.EPI_BIAS, False)[:2]
                t_chosen = t_us(fn)
                best, t_best, seen = None, 1e30, set()
                for t in G.TILES:
                    for s in (1, 2, 3, 4, 8):
                        G.override(t, s)
                        lab = G.query(L.GEMM_NT, M, N, K, L.EPI_BIAS, False)[:2]
                        if lab in seen:
                            continue
                        seen.add(lab)
                        u = t_us(fn, 10)
                        if u < t_best:
                            best, t_best = lab, u
                G.override(0, 0)
                rows.append((t_best / t_chosen, M, N, K, chosen, t_chosen, best, t_best))
                print("NT %5d x %5d x %5d + bias: model %d/%d %.1f us | best %d/%d %.1f us | %.2f" % (M, N, K, chosen[0], chosen[1], t_chosen, best[0], best[1], t_best, t_best / t_chosen), flush=True)
    L.check(L.lib().blm_gemm_plan_clear(1), "clear")
    rows.sort()
    print("worst five:", [(round(r[0], 2), r[1], r[2], r[3], r[4], r[6]) for r in rows[:5]])
    print("mean best/chosen %.3f, shapes under 0.85: %d of %d" % (sum(r[0] for r in rows) / len(rows), sum(r[0] < 0.85 for r in rows), len(rows)))


if __name__ == "__main__":
    main()

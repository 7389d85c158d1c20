#!/usr/bin/env python3
"""Randomised check of blm_gemm against fp64 over the whole launch heuristic: N shapes drawn from the ranges the tile /
split rules branch on (tiny to 33000-wide, K from 1 to 33000, aligned and odd), all three layouts, accumulate on/off,
bias epilogue.  usage: gemm_fuzz.py [count] [seed]
       gemm_fuzz.py plans [count] [seed]   the same shapes, each under a RANDOM plan forced through blm_gemm_plan_override:
                           every tile (11 / 12 / 21 / 22 / 28) x uniform slices (1 ... 16) or tail slicing (-2 ... -32) -- what the
                           planner could ever hand to a launch, legal or clamped
       gemm_fuzz.py perf   PERFORMANCE regression check of the launch planner (csrc/gemm_plan.hip, cost model only: plan
                           table off): on a log-spaced M x N x K grid (64 ... 33000, all three layouts) every candidate tile x
                           slice count is timed and the planner's own choice must reach >= 0.6 of the best candidate's rate
                           on every shape of 30 us and more (tools/gemm_tune.py --grid --coarse; the full 2160-shape sweep
                           the model was fitted to: --grid alone)."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "perf":
        import subprocess
        sys.exit(subprocess.call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune.py"), "--grid",
                                  "--coarse", "--min-frac", "0.6"]))
    plans = len(sys.argv) > 1 and sys.argv[1] == "plans"
    if plans:
        del sys.argv[1]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    dims = [1, 3, 17, 32, 63, 64, 65, 100, 127, 128, 129, 200, 256, 384, 500, 512, 640, 1000, 1024, 1536, 2048, 2240, 3000, 4096,
            8192, 33000]
    ks = [1, 2, 5, 31, 32, 33, 64, 100, 256, 500, 512, 1000, 1024, 1536, 2048, 2240, 4096, 8192, 33000]
    worst = 0.0
    for it in range(n):
        while True:
            M, N, K = rnd.choice(dims), rnd.choice(dims), rnd.choice(ks)
            if M * N <= 70e6 and M * K <= 70e6 and N * K <= 70e6 and 2.0 * M * N * K <= 3e12:
                break
        if rnd.random() < 0.3:  # nudge off the aligned grid
            M, N, K = max(1, M + rnd.randint(-3, 3)), max(1, N + rnd.randint(-3, 3)), max(1, K + rnd.randint(-3, 3))
        op = rnd.choice([L.GEMM_NT, L.GEMM_NN, L.GEMM_TN])
        acc, bias = rnd.random() < 0.5, rnd.random() < 0.3
        g = torch.Generator(device=dev).manual_seed(it)
        rn = lambda r, c: torch.randn(r, c, device=dev, generator=g)  # noqa: E731
        if op == L.GEMM_NT:
            A, B = rn(M, K), rn(N, K)
            ref = A.double() @ B.double().t()
        elif op == L.GEMM_NN:
            A, B = rn(M, K), rn(K, N)
            ref = A.double() @ B.double()
        else:
            A, B = rn(K, M), rn(K, N)
            ref = A.double().t() @ B.double()
        C = rn(M, N)
        bv = torch.randn(N, device=dev, generator=g) if bias else None
        want = ref + (C.double() if acc else 0) + (bv.double() if bias else 0)
        plan = (0, 0)
        if plans:
            plan = (rnd.choice([11, 12, 21, 22, 28]), rnd.choice([1, 2, 3, 4, 6, 8, 16, -2, -3, -4, -8, -16, -32]))
            L.check(L.lib().blm_gemm_plan_override(*plan), "override")
        ops.gemm(op, A, B, C, M, N, K, A.stride(0), B.stride(0), N, accumulate=acc,
                 epilogue=L.EPI_BIAS if bias else L.EPI_NONE, bias=bv)
        if plans:
            L.check(L.lib().blm_gemm_plan_override(0, 0), "override")
        err = float((C.double() - want).norm() / want.norm().clamp_min(1e-30))
        worst = max(worst, err)
        if not (err < 2e-5):
            print("FAIL", ("NT", "NN", "TN")[op], M, N, K, "acc" if acc else "", "bias" if bias else "", "plan %d/%d" % plan, err)
            sys.exit(1)
        del A, B, C, ref, want
    print("gemm_fuzz: %d shapes ok, worst relative error %.2e" % (n, worst))


if __name__ == "__main__":
    main()

"""Where the HOST's time goes in a training step of one of tools/gemm_tune.py's workloads (cProfile over N steps; the device is
not waited for inside the profiled region)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gemm_tune as G  # noqa: E402


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    step, _ = G.build(name, torch.device("cuda:0"))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime")
    print("host profile of %s, %d steps (times are totals over all steps; divide by %d)" % (name, n, n))
    st.print_stats(28)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What do the HIP-event brackets of bench.py's kernel timer cost the step they sit in?  The same training steps with and without
ops.KernelTimer (tagged launches only, as bench.py's headline uses it), interleaved.  usage: timer_cost.py cfg3|cfg1|..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gemm_tune as G  # noqa: E402
from bayeslms_amd import ops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    step, _ = G.build(name, torch.device("cuda:0"))
    for _ in range(8):
        step()

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    a, b = [], []
    for _ in range(4):
        a.append(run())
        tm = ops.KernelTimer()
        ops.set_kernel_timer(tm)
        b.append(run())
        ops.set_kernel_timer(None)
        nb = len(tm.records) / n
    print("%s: without the timer %s ms/step; with it %s ms/step (%.0f brackets per step)"
          % (name, " ".join("%.3f" % v for v in a), " ".join("%.3f" % v for v in b), nb))


if __name__ == "__main__":
    main()

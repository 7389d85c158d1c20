#!/usr/bin/env python3
"""Runs N steps of one of tools/gemm_tune.py's workloads (cfg3, gauss, cfg2, recipe_tlm, recipe_lstm, cfg1, eval_*):
the thing to put behind `rocprofv3 --kernel-trace --stats -- python3 tools/run_workload.py recipe_tlm 10`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import gemm_tune as G  # noqa: E402


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    step, tokens = G.build(name, dev)
    for _ in range(3):
        step()
    if os.environ.get("NOGC"):  # A/B: the cyclic collector off / everything allocated so far frozen
        import gc
        gc.freeze() if os.environ["NOGC"] == "freeze" else gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = time.perf_counter() - t0  # the host is done enqueueing: close to the total = the host, not the GPU, sets the pace
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%s: %.3f ms/step, %.1f tokens/s (host done enqueueing after %.3f ms/step)" % (name, 1e3 * el / n, tokens * n / el, 1e3 * host / n))


if __name__ == "__main__":
    main()

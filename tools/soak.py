#!/usr/bin/env python3
"""Stability run: N steps of a tools/gemm_tune.py workload; prints step time and allocator state at intervals and fails on a
non-finite loss or on growing reserved memory.  usage: soak.py cfg3 1500"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import gemm_tune as G  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device("cuda:0")
    step, tokens = G.build(name, dev)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    r0 = torch.cuda.memory_reserved()
    t0 = time.perf_counter()
    every = max(1, n // 6)
    for i in range(1, n + 1):
        step()
        if i % every == 0:
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print("%s step %5d: %.3f ms/step so far, reserved %.2f GB (start %.2f GB), allocator retries %d"
                  % (name, i, 1e3 * el / i, torch.cuda.memory_reserved() / 2**30, r0 / 2**30, torch.cuda.memory_stats()["num_alloc_retries"]), flush=True)
    torch.cuda.synchronize()
    grown = torch.cuda.memory_reserved() - r0
    bad = [k for k, p in enumerate(G._last_model_params()) if not torch.isfinite(p).all()] if hasattr(G, "_last_model_params") else []
    print("%s: %d steps, reserved memory grew by %.1f MB, non-finite parameter tensors: %d" % (name, n, grown / 2**20, len(bad)))
    sys.exit(1 if (bad or grown > (256 << 20)) else 0)


if __name__ == "__main__":
    main()

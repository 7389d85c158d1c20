#!/usr/bin/env python3
"""bench.py's architecture-search window (Architect.step + network step) on its own, for rocprofv3.  usage: search_workload.py lstm|tlm [windows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    r = bench.search_leg(kind, torch.device("cuda:0"), steps=n, warm=3)
    print({k: r[k] for k in ("value", "unit", "ms_per_window")})


if __name__ == "__main__":
    main()

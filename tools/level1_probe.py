#!/usr/bin/env python3
"""bench.level1_leg alone (INTEGRATION.md level 1 priced: the headline model under the reference's own loop shape), the thing to
put behind `rocprofv3 --kernel-trace --stats -- python3 tools/level1_probe.py`."""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    if os.environ.get("SLAB", "1") == "0":  # A/B of ops._GradSlab: a zeros_like per missing gradient, as before round 5
        from bayeslms_amd import ops
        ops.set_grad_slab(False)
    print(bench.level1_leg(dev, SimpleNamespace(steps=steps, warmup=5), 20.0))

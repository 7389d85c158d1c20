#!/usr/bin/env python3
"""Which Python lines issue the small torch-native launches (fill / add / cat / copy) inside the architecture-search window?
torch.profiler with stacks over bench.search_leg; prints, per aten op of interest, the call sites by count.
usage: search_fill_trace.py lstm|tlm"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
    dev = torch.device("cuda:0")
    bench.search_leg(kind, dev, steps=2, warm=2)
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        bench.search_leg(kind, dev, steps=2, warm=0)
    want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::cat", "aten::copy_", "aten::clone", "aten::mul", "aten::stack")
    sites = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        if ev.name in want:
            st = [f for f in (ev.stack or []) if "bayeslms_amd" in f or "bench.py" in f or "autograd" in f]
            key = (st[0] if st else "(no python frame: autograd engine)").strip()[-110:]
            shp = str(ev.input_shapes[:1])[:40] if ev.input_shapes else ""
            sites[ev.name][(key, shp)] += 1
    for name in want:
        if sites[name]:
            print("==", name, sum(sites[name].values()))
            for (k, shp), c in sites[name].most_common(14):
                print("   %4d  %s  %s" % (c, k, shp))


if __name__ == "__main__":
    main()

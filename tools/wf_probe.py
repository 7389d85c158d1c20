import sys, os, time
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
import gemm_tune as G
from bayeslms_amd import ops
dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    step, tokens = G.build(name, dev)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    tm = ops.KernelTimer(); ops.set_kernel_timer(tm)
    for _ in range(5): step()
    s = tm.summary(); ops.set_kernel_timer(None)
    print("%s: %.3f ms/step (%.0f tok/s), host enqueue %.3f ms/step" % (name, 1e3*el/n, tokens*n/el, 1e3*host/n))
    for k, v in sorted(s.items(), key=lambda kv: -kv[1]["avg_ms"]*kv[1]["n"])[:8]:
        print("    %-40s %.3f ms x %d" % (k, v["avg_ms"], v["n"]))

#!/usr/bin/env python3
"""Per-step times of the headline training step (tools/gemm_tune.py workload cfg3) as the GEMM mode is switched f32 -> bf16x6 ->
f32 -> bf16x3 -> bf16x6 -> f32, each step bracketed on its own, with the clock / power / temperature rocm-smi reports after each
series.  Written when one bench run reported the opt-in bf16x6 step at 28.7 ms (16.5 on every other box; the fp32 step of the same
run was at its usual 19.97): does the mode settle, and does it hold its rate?  (It does: 16.6-16.7 ms from the fourth step on, 840 W
against 770 W in fp32.)  -> profiles/r05_gemm_mode_step_series.txt"""
import os, sys, time, subprocess
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import torch
import gemm_tune as G
from bayeslms_amd import ops
dev = torch.device("cuda:0")
step, tokens = G.build("cfg3", dev)
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "Power" in l or "junction" in l.lower()) and "GPU[0]" in l]
        return " | ".join(keep)[:400]
    except Exception as e:
        return repr(e)
def series(mode, n):
    ops.set_gemm_mode(mode)
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(mode, " ".join("%.1f" % t for t in ts), flush=True)
    print("   ", smi(), flush=True)
print(smi())
series("f32", 15)
series("bf16x6", 60)
series("f32", 15)
series("bf16x3", 30)
series("bf16x6", 60)
series("f32", 15)

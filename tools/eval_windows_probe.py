#!/usr/bin/env python3
"""evaluate() of the headline Transformer (12 windows of 20 x 128 held-out tokens, bench._eval_leg) against the number of windows run as
one batch (BLM_EVAL_WINDOWS): one MI355X, round 5: 1 -> 1.085 M tokens/s, 2 -> 1.14 M, 3 -> 1.18 M, 4 -> 1.17 M, 6 -> 1.197 M, 12 -> 1.19 M."""
import os, sys, time
sys.path.insert(0, "."); 
import torch, bench
from bayeslms_amd import engine, model as M
dev = torch.device("cuda:0")
torch.manual_seed(1111)
m = M.BayesTransformerModel(bench.V, bench.D_MODEL, bench.NHEAD, bench.D_FF, bench.NLAYERS, bench.DROPOUT, True, "FFN").to(dev)
for w in ("1", "2", "3", "4", "6", "12"):
    os.environ["BLM_EVAL_WINDOWS"] = w
    r = [bench._eval_leg(m, 128, dev, engine, bench.V, bench.tlm_flops_per_token(128, train=False))["value"] for _ in range(3)]
    print("BLM_EVAL_WINDOWS", w, r, flush=True)

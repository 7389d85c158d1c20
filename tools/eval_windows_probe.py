#!/usr/bin/env python3
"""evaluate() against the number of windows per call (BLM_EVAL_WINDOWS), bench._eval_leg's 12 windows of 20 columns:
the headline Transformer (windows as one wider batch; one MI355X, round 5: 1 -> 1.085 M tokens/s, 3 -> 1.18 M, 6 -> 1.197 M) and the
configs[1] Bayesian LSTM (consecutive windows as one longer window, state carried).  usage: eval_windows_probe.py [tlm|lstm]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bayeslms_amd import engine, model as M  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "tlm"
torch.manual_seed(1111)
if which == "tlm":
    m = M.BayesTransformerModel(bench.V, bench.D_MODEL, bench.NHEAD, bench.D_FF, bench.NLAYERS, bench.DROPOUT, True, "FFN").to(dev)
    seq, fl = 128, bench.tlm_flops_per_token(128, train=False)
else:
    m = M.BayesRNNModel("LSTM", bench.V, 1024, 1024, 2, 0.2, True, 3).to(dev)
    seq, fl = 35, bench.lstm_flops_per_token(bench.V, train=False)
for w in ("1", "2", "3", "4", "6", "12", "0"):
    os.environ["BLM_EVAL_WINDOWS"] = w
    r = [bench._eval_leg(m, seq, dev, engine, bench.V, fl) for _ in range(3)]
    print(which, "BLM_EVAL_WINDOWS", w, [x["value"] for x in r], "loss", r[0]["loss"], flush=True)

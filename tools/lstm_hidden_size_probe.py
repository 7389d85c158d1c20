#!/usr/bin/env python3
"""What does a hidden size that leaves the fused LSTM step kernels cost?  Training tokens/s of the 2-layer LSTM LM (engine.Trainer
step, V 10000, batch 20, seq_len 35, tied) for E = H in the classic word-language-model sizes (200, 650, 1500 -- none a multiple
of 32) and their neighbours that are.  -> profiles/r05_lstm_hidden_size_probe.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bayeslms_amd import engine, model as M, ops  # noqa: E402

dev = torch.device("cuda:0")
sizes = [int(x) for x in sys.argv[1:]] or [200, 224, 256, 650, 672, 1024, 1500, 1536]
for H in sizes:
    torch.manual_seed(1)
    m = M.RNNModel("LSTM", 10000, H, H, 2, 0.2, True).to(dev)
    fl = bench.lstm_flops_per_token(10000, E=H, H=H)
    r, _ = bench._train_leg(m, None, 35, 20, 1.0, 12, 4, dev, engine, ops, vocab=10000, flops_per_token=fl)
    print("E = H = %4d (H %% 32 = %2d): %8.0f tokens/s, %6.3f ms/step, %.3f of the fp32 MFMA peak" % (H, H % 32, r["value"], r["ms_per_step"], r["step_roofline"]["frac"]), flush=True)
    del m
    torch.cuda.empty_cache()

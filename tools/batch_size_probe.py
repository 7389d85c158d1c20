#!/usr/bin/env python3
"""Training tokens/s against the batch size: the headline Transformer at T 128, B 16 ... 256 and the configs[1] Bayesian LSTM at T 35,
B 8 ... 256 (last column: fraction of the fp32 MFMA peak).  -> profiles/r05_batch_size_probe.txt"""
import os, sys
sys.path.insert(0, ".")
import torch, bench
from bayeslms_amd import engine, model as M, ops
dev = torch.device("cuda:0")
for B in (16, 32, 64, 128, 256):
    torch.manual_seed(1)
    m = M.BayesTransformerModel(33000, 512, 8, 4096, 6, 0.2, True, "FFN").to(dev)
    kl = lambda mm: mm.transformerlayers[0].linear2.kl_divergence()
    kl.fusable = True
    r, _ = bench._train_leg(m, kl, 128, B, 0.1, 6, 3, dev, engine, ops, vocab=33000, flops_per_token=bench.tlm_flops_per_token(128))
    print("Transformer (headline) T 128 B %3d: %8.0f tokens/s, %7.3f ms/step, %.3f" % (B, r["value"], r["ms_per_step"], r["step_roofline"]["frac"]), flush=True)
    del m; torch.cuda.empty_cache()
for B in (8, 20, 64, 128, 256):
    torch.manual_seed(1)
    m = M.BayesRNNModel("LSTM", 33000, 1024, 1024, 2, 0.2, True, 3).to(dev)
    r, _ = bench._train_leg(m, (lambda mm: mm.rnn.kl_divergence()), 35, B, 1.0, 6, 3, dev, engine, ops, vocab=33000, flops_per_token=bench.lstm_flops_per_token(33000))
    print("Bayesian LSTM (configs[1]) T 35 B %3d: %8.0f tokens/s, %7.3f ms/step, %.3f" % (B, r["value"], r["ms_per_step"], r["step_roofline"]["frac"]), flush=True)
    del m; torch.cuda.empty_cache()

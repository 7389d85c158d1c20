#!/usr/bin/env python3
"""The Bayesian / GP / Variational LSTM language models at 650 against 672 hidden units (V 10000, batch 20, seq_len 35): what an odd
hidden size costs each family.  -> profiles/r05_lstm_hidden_size_probe.txt"""
import os, sys
sys.path.insert(0, ".")
import torch, bench
from bayeslms_amd import engine, model as M, ops, train as TR
from types import SimpleNamespace
dev = torch.device("cuda:0")
for name, build, ns in (
    ("Bayesian LSTM gate 3", lambda H: M.BayesRNNModel("LSTM", 10000, H, H, 2, 0.2, True, 3), dict(uncertainty="Bayesian", L_bayes_pos=3)),
    ("GP-LSTM '33'", lambda H: M.GaussRNNModel("LSTM", 10000, H, H, 2, 0.2, True, "33"), dict(uncertainty="Gaussian", L_gauss_pos="33")),
    ("Variational LSTM '11'", lambda H: M.VariationalRNNModel("LSTM", 10000, H, H, 2, 0.2, True, "11"), dict(uncertainty="Variational", L_v_pos="11"))):
    for H in (650, 672):
        torch.manual_seed(1)
        m = build(H).to(dev)
        a = SimpleNamespace(**{**dict(model="LSTM", uncertainty="none", T_bayes_pos="none", L_bayes_pos=0, T_gauss_pos=0, L_gauss_pos="00", L_v_pos="00", T_v_pos=0), **ns})
        r, _ = bench._train_leg(m, TR.kl_selector(a), 35, 20, 1.0, 8, 3, dev, engine, ops, vocab=10000, flops_per_token=bench.lstm_flops_per_token(10000, E=H, H=H))
        print("%-22s E = H = %4d: %8.0f tokens/s, %6.3f ms/step" % (name, H, r["value"], r["ms_per_step"]), flush=True)
        del m

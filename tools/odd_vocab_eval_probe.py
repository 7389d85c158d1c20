#!/usr/bin/env python3
"""evaluate() of the headline Transformer and the 2 x 1024 LSTM at V = 33000 against 33278 (wikitext-2: not a multiple of 4 words, so
the decoder cannot be fused with the cross entropy and the logits are stored -- with padded rows): 1.19 M / 735 k tokens/s either way."""
import os, sys
sys.path.insert(0, ".")
import torch, bench
from bayeslms_amd import engine, model as M
dev = torch.device("cuda:0")
for V in (33000, 33278):
    torch.manual_seed(1)
    m = M.BayesTransformerModel(V, bench.D_MODEL, bench.NHEAD, bench.D_FF, bench.NLAYERS, bench.DROPOUT, True, "FFN").to(dev)
    r = [bench._eval_leg(m, 128, dev, engine, V, bench.tlm_flops_per_token(128, V_=V, train=False))["value"] for _ in range(2)]
    print("evaluate() Transformer V %d: %s tokens/s" % (V, r), flush=True)
    del m
    m = M.RNNModel("LSTM", V, 1024, 1024, 2, 0.2, True).to(dev)
    r = [bench._eval_leg(m, 35, dev, engine, V, bench.lstm_flops_per_token(V, train=False))["value"] for _ in range(2)]
    print("evaluate() LSTM V %d: %s tokens/s" % (V, r), flush=True)
    del m

#!/usr/bin/env python3
"""Where do shapes other than BASELINE's leave the fast paths?  Training tokens/s (engine.Trainer step) and the fraction of the fp32 MFMA
peak of (a) the headline Transformer family at other widths and (b) the same models with vocabularies that are not a multiple of 4
(wikitext-2 has 33278 words).  usage: shape_cliff_probe.py [tlm|vocab|all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bayeslms_amd import engine, model as M, ops  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"


def tlm(V, d, heads, ff, L=6, T=128, B=64, tag=""):
    torch.manual_seed(1)
    m = M.BayesTransformerModel(V, d, heads, ff, L, 0.2, True, "FFN").to(dev)
    fl = bench.tlm_flops_per_token(T, V_=V, L_=L, d=d, ff=ff)
    kl = lambda mm: mm.transformerlayers[0].linear2.kl_divergence()  # noqa: E731
    kl.fusable = True
    r, _ = bench._train_leg(m, kl, T, B, 0.1, 8, 3, dev, engine, ops, vocab=V, flops_per_token=fl)
    print("Transformer V %5d d %4d heads %2d (x %3d) ff %4d L %d, T %d B %d%s: %8.0f tokens/s, %7.3f ms/step, %.3f of the fp32 MFMA peak"
          % (V, d, heads, d // heads, ff, L, T, B, tag, r["value"], r["ms_per_step"], r["step_roofline"]["frac"]), flush=True)
    del m
    torch.cuda.empty_cache()


def lstm(V, H, T=35, B=64):
    torch.manual_seed(1)
    m = M.RNNModel("LSTM", V, H, H, 2, 0.2, True).to(dev)
    fl = bench.lstm_flops_per_token(V, E=H, H=H)
    r, _ = bench._train_leg(m, None, T, B, 1.0, 8, 3, dev, engine, ops, vocab=V, flops_per_token=fl)
    print("LSTM V %5d E = H = %4d, T %d B %d: %8.0f tokens/s, %7.3f ms/step, %.3f of the fp32 MFMA peak"
          % (V, H, T, B, r["value"], r["ms_per_step"], r["step_roofline"]["frac"]), flush=True)
    del m
    torch.cuda.empty_cache()


if which in ("tlm", "all"):
    tlm(33000, 512, 8, 4096)
    tlm(33000, 200, 2, 200, L=2, T=35, B=20, tag=" (train.py's defaults)")
    tlm(33000, 256, 4, 1024)
    tlm(33000, 384, 6, 1536)
    tlm(33000, 768, 12, 3072)
    tlm(33000, 1024, 16, 4096)
    tlm(33000, 1000, 8, 4000)
if which == "ln":  # the widths whose LayerNorm left the register kernels before round 5 (768) / still does (384)
    tlm(33000, 768, 12, 3072)
    tlm(33000, 384, 6, 1536)
    tlm(33000, 1536, 24, 6144, L=2)
if which in ("vocab", "all"):
    for V in (33000, 33278, 33001, 10001):
        tlm(V, 512, 8, 4096)
    for V in (33000, 33278, 10001):
        lstm(V, 1024)

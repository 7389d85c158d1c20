#!/usr/bin/env python3
"""What does a head size other than 64 cost?  Causal attention forward / backward through the C ABI at T 128, B 64, dropout 0.2 for
(heads, head_dim) = (8, 64), (4, 128), (16, 32), (8, 96), (2, 256).  -> profiles/r05_attention_head_dim_probe.txt"""
import os, sys, ctypes
sys.path.insert(0, ".")
import torch
from bayeslms_amd._lib import Rng, check, lib, ptr, stream
T, B = 128, 64
L = lib()
for nh, hd in ((8, 64), (4, 128), (16, 32), (8, 96), (2, 256)):
    d = nh * hd
    qkv = torch.randn(T, B, 3 * d, device="cuda")
    out = torch.empty(T, B, d, device="cuda"); lse = torch.empty(B * nh, T, device="cuda")
    dout = torch.randn(T, B, d, device="cuda"); dqkv = torch.empty_like(qkv)
    rng = Rng(1234, 0x20000000, 1)
    def fwd():
        check(L.blm_attn_fwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(lse), T, B, nh, hd, 0.2, ctypes.byref(rng), 0, B, stream()))
    def bwd():
        check(L.blm_attn_bwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(dqkv) + 4 * d, ptr(dqkv) + 8 * d, 3 * d, T, B, nh, hd, 0.2, ctypes.byref(rng), 0, B, stream()))
    res = []
    for f in (fwd, bwd):
        try:
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.append("%.1f us" % (e0.elapsed_time(e1) * 100))
        except Exception as e:
            res.append("refused: " + str(e)[:60])
    fl = 4.0 * T * T * hd * B * nh
    print("heads %2d x head_dim %3d (d_model %d): fwd %s, bwd %s   (fwd matrix work %.2f GFLOP)" % (nh, hd, d, res[0], res[1], fl / 1e9), flush=True)

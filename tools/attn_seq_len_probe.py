#!/usr/bin/env python3
"""Causal attention forward / backward (one-launch backward with the dS workspace) through the C ABI at 8192 tokens, 8 heads x 64,
dropout 0.2, against the sequence length (T 64 ... 2048): T <= 128 on the one-tile kernels, beyond on the chunked (flash-style) ones.
One MI355X, round 5: forward 32 us (T 128) ... 276 us (T 2048, 62 TFLOP/s of causal work), backward 74 ... 1340 us; T 129 costs 117 / 341 us."""
import os, sys, ctypes
sys.path.insert(0, ".")
import torch
from bayeslms_amd._lib import Rng, check, lib, ptr, stream
L = lib()
nh, hd = 8, 64
d = nh * hd
for T, B in ((64, 128), (128, 64), (129, 64), (256, 32), (512, 16), (1024, 8), (2048, 4)):
    qkv = torch.randn(T, B, 3 * d, device="cuda")
    out = torch.empty(T, B, d, device="cuda"); lse = torch.empty(B * nh, T, device="cuda")
    dout = torch.randn(T, B, d, device="cuda"); dqkv = torch.empty_like(qkv)
    rng = Rng(1234, 0x20000000, 1)
    nws = int(L.blm_attn_bwd_ws_floats(T, B, nh, hd)); ws = torch.empty(max(nws, 1), device="cuda")
    def fwd():
        check(L.blm_attn_fwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(lse), T, B, nh, hd, 0.2, ctypes.byref(rng), 0, B, stream()))
    def bwd():
        check(L.blm_attn_bwd_ws(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(dqkv) + 4 * d, ptr(dqkv) + 8 * d, 3 * d, T, B, nh, hd, 0.2, ctypes.byref(rng), 0, B, ptr(ws), nws, stream()))
    res = []
    for f in (fwd, bwd):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    fl = 2.0 * T * T * hd * B * nh  # causal half of 4 T^2 hd
    print("T %4d B %3d (8192 tokens, 8 x 64): fwd %7.1f us (%5.1f TFLOP/s causal), bwd %7.1f us (%5.1f)" % (T, B, res[0], fl / res[0] / 1e6, res[1], 2.5 * fl / res[1] / 1e6), flush=True)

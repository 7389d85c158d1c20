"""Times the causal attention kernels through the C ABI at the cfg3 shape (T 128, B 64, 8 heads x 64), with and without
dropout, back-to-back launches: forward, backward without workspace (two recomputations of the probabilities) and with
the dS workspace (blm_attn_bwd_ws: one launch)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import Rng, check, lib, ptr, stream  # noqa: E402


def main():
    T, B, nh, hd = int(os.environ.get("T", "128")), int(os.environ.get("B", "64")), 8, 64
    d = nh * hd
    dev = "cuda"
    qkv = torch.randn(T, B, 3 * d, device=dev)
    out = torch.empty(T, B, d, device=dev)
    lse = torch.empty(B * nh, T, device=dev)
    dout = torch.randn(T, B, d, device=dev)
    dqkv = torch.empty_like(qkv)
    L = lib()
    nws = int(L.blm_attn_bwd_ws_floats(T, B, nh, hd))
    ws = torch.empty(max(nws, 1), device=dev)
    for pdrop in (0.0, 0.2):
        rng = Rng(1234, 0x20000000, 1)

        def fwd():
            check(L.blm_attn_fwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(lse), T, B, nh, hd,
                                 pdrop, ctypes.byref(rng), 0, B, stream()))

        def bwd():
            check(L.blm_attn_bwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(dout), ptr(lse),
                                 ptr(dqkv), ptr(dqkv) + 4 * d, ptr(dqkv) + 8 * d, 3 * d, T, B, nh, hd, pdrop,
                                 ctypes.byref(rng), 0, B, stream()))

        def bwd_ws():
            check(L.blm_attn_bwd_ws(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(dout), ptr(lse),
                                    ptr(dqkv), ptr(dqkv) + 4 * d, ptr(dqkv) + 8 * d, 3 * d, T, B, nh, hd, pdrop,
                                    ctypes.byref(rng), 0, B, ptr(ws), nws, stream()))
        for name, f in (("fwd", fwd), ("bwd (no workspace)", bwd), ("bwd (dS workspace)", bwd_ws)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            print(f"attention {name} p={pdrop}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")


if __name__ == "__main__":
    main()

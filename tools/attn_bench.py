"""Times the causal attention kernels (forward, backward) at the cfg3 shape, with and without dropout."""
import sys
import torch
sys.path.insert(0, ".")
from bayeslms_amd._lib import lib, check, ptr, stream

def main():
    import os
    T, B, nh, hd = int(os.environ.get("T", "128")), int(os.environ.get("B", "64")), 8, 64
    d = nh * hd
    dev = "cuda"
    qkv = torch.randn(T, B, 3 * d, device=dev)
    out = torch.empty(T, B, d, device=dev)
    lse = torch.empty(B * nh, T, device=dev)
    dout = torch.randn(T, B, d, device=dev)
    dqkv = torch.empty_like(qkv)
    L = lib()
    from bayeslms_amd._lib import Rng
    import ctypes
    for pdrop in (0.0, 0.2):
        rng = Rng(1234, 0x2000, 1)
        def fwd():
            check(L.blm_attn_fwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(lse), T, B, nh, hd,
                                 pdrop, ctypes.byref(rng), 0, B, stream()))
        def bwd():
            check(L.blm_attn_bwd(ptr(qkv), ptr(qkv) + 4 * d, ptr(qkv) + 8 * d, 3 * d, ptr(out), ptr(dout), ptr(lse),
                                 ptr(dqkv), ptr(dqkv) + 4 * d, ptr(dqkv) + 8 * d, 3 * d, T, B, nh, hd, pdrop,
                                 ctypes.byref(rng), 0, B, stream()))
        for name, f in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print(f"attention {name} p={pdrop}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")

if __name__ == "__main__":
    main()

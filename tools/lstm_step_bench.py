"""us per launch of the fused LSTM step kernels at the cfg2 shape (B=64, H=1024) as a DEPENDENT chain on one stream
(step t's h / dgates feed step t+1), i.e. what a time loop pays.  BLM_LSTM_RING=1|2|4 caps the load ring (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402


def main():
    B, H = int(os.environ.get("B", "64")), int(os.environ.get("H", "1024"))
    dev = "cuda"
    torch.manual_seed(0)
    xw = torch.randn(B, 4 * H, device=dev)
    w = torch.randn(4 * H, H, device=dev) * 0.03
    wt = w.t().contiguous()
    hs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    cs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    ga = torch.rand(B, 4 * H, device=dev)
    dgs = [torch.randn(B, 4 * H, device=dev) * 0.01 for _ in range(2)]
    dcs = [torch.randn(B, H, device=dev) * 0.01 for _ in range(2)]
    dy = torch.randn(B, H, device=dev) * 0.01
    L = lib()
    n = 400

    def fwd():
        for i in range(n):
            check(L.blm_lstm_step_fwd(ptr(xw), ptr(w), ptr(hs[i & 1]), ptr(cs[i & 1]), ptr(hs[1 - (i & 1)]),
                                      ptr(cs[1 - (i & 1)]), ptr(ga), None, B, H, stream()))

    def bwd():
        for i in range(n):
            check(L.blm_lstm_step_bwd(ptr(dgs[i & 1]), ptr(wt), ptr(dy), ptr(dcs[i & 1]), ptr(cs[0]), ptr(cs[1]), ptr(ga),
                                      ptr(dgs[1 - (i & 1)]), ptr(dcs[1 - (i & 1)]), None, B, H, stream()))
    # two independent recurrences on two streams (what the layer wavefront of ops.lstm_stack does)
    hs2 = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    cs2 = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    ga2 = torch.rand(B, 4 * H, device=dev)
    w2 = torch.randn(4 * H, H, device=dev) * 0.03
    side = torch.cuda.Stream()

    def fwd2():
        side.wait_stream(torch.cuda.current_stream())
        for i in range(n):
            check(L.blm_lstm_step_fwd(ptr(xw), ptr(w), ptr(hs[i & 1]), ptr(cs[i & 1]), ptr(hs[1 - (i & 1)]),
                                      ptr(cs[1 - (i & 1)]), ptr(ga), None, B, H, stream()))
            with torch.cuda.stream(side):
                check(L.blm_lstm_step_fwd(ptr(xw), ptr(w2), ptr(hs2[i & 1]), ptr(cs2[i & 1]), ptr(hs2[1 - (i & 1)]),
                                          ptr(cs2[1 - (i & 1)]), ptr(ga2), None, B, H, stream()))
        torch.cuda.current_stream().wait_stream(side)
    dgs2 = [torch.randn(B, 4 * H, device=dev) * 0.01 for _ in range(2)]
    dcs2 = [torch.randn(B, H, device=dev) * 0.01 for _ in range(2)]
    wt2 = w2.t().contiguous()

    def bwd2():
        side.wait_stream(torch.cuda.current_stream())
        for i in range(n):
            check(L.blm_lstm_step_bwd(ptr(dgs[i & 1]), ptr(wt), ptr(dy), ptr(dcs[i & 1]), ptr(cs[0]), ptr(cs[1]), ptr(ga),
                                      ptr(dgs[1 - (i & 1)]), ptr(dcs[1 - (i & 1)]), None, B, H, stream()))
            with torch.cuda.stream(side):
                check(L.blm_lstm_step_bwd(ptr(dgs2[i & 1]), ptr(wt2), ptr(dy), ptr(dcs2[i & 1]), ptr(cs2[0]), ptr(cs2[1]), ptr(ga2),
                                          ptr(dgs2[1 - (i & 1)]), ptr(dcs2[1 - (i & 1)]), None, B, H, stream()))
        torch.cuda.current_stream().wait_stream(side)
    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd), ("fwd2", fwd2), ("bwd2", bwd2)):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1000)
        out[name] = best
    print("lstm_step B=%d H=%d waves=%s: fwd %.2f us/step, bwd %.2f us/step; two chains on two streams: fwd %.2f / bwd %.2f us per step PAIR"
          % (B, H, "4", out["fwd"], out["bwd"], out["fwd2"], out["bwd2"]))


if __name__ == "__main__":
    main()

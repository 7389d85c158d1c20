"""The backward LSTM step with the contraction split over workgroups (blm_lstm_step_bwd_ks: 32 x 32 dh tiles, S K slices, the last
arriver of a tile adds the partials in slice order and runs the cell backward) against the 16 x 16 kernel (blm_lstm_step_bwd):
same results (vs fp64 too), then us per launch as a DEPENDENT chain on one stream.  usage: [B=64] [H=1024] lstm_bwd_ks_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402


def main():
    B, H = int(os.environ.get("B", "64")), int(os.environ.get("H", "1024"))
    G = 4 * H
    dev = "cuda"
    torch.manual_seed(0)
    L = lib()
    wt = (torch.randn(4 * H, H, device=dev) * 0.03).t().contiguous()
    cs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    ga = torch.rand(B, G, device=dev)
    dgs = [torch.randn(B, G, device=dev) * 0.01 for _ in range(2)]
    dcs = [torch.randn(B, H, device=dev) * 0.01 for _ in range(2)]
    dy = torch.randn(B, H, device=dev) * 0.01
    ws = torch.zeros(int(L.blm_lstm_step_bwd_ks_ws_floats(B, H, G)), device=dev)
    assert ws.numel() > 0, "shape not taken by the split kernel"
    # parity: one step, both kernels, against fp64
    o16, c16, h16 = torch.empty(B, G, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    oks, cks, hks = torch.empty(B, G, device=dev), torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    check(L.blm_lstm_step_bwd(ptr(dgs[0]), ptr(wt), ptr(dy), ptr(dcs[0]), ptr(cs[0]), ptr(cs[1]), ptr(ga), ptr(o16), ptr(c16), ptr(h16), B, H, stream()))
    for _ in range(3):  # three launches on one workspace: the counters re-arm themselves
        check(L.blm_lstm_step_bwd_ks(ptr(dgs[0]), ptr(wt), ptr(dy), ptr(dcs[0]), ptr(cs[0]), ptr(cs[1]), ptr(ga), ptr(oks), ptr(cks), ptr(hks),
                                     ptr(ws), B, H, G, stream()))
    torch.cuda.synchronize()
    dh64 = dgs[0].double() @ wt.double().t()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))  # noqa: E731
    ntiles = (H // 32) * ((B + 31) // 32)
    print({"B": B, "H": H, "dh vs fp64: 16x16": rel(h16, dh64), "split": rel(hks, dh64), "dgates split vs 16x16": rel(oks, o16),
           "dc split vs 16x16": rel(cks, c16), "counters re-armed": bool((ws[:ntiles].view(torch.int32) == 0).all())})
    runs = [hks.clone()]
    for _ in range(2):
        check(L.blm_lstm_step_bwd_ks(ptr(dgs[0]), ptr(wt), ptr(dy), ptr(dcs[0]), ptr(cs[0]), ptr(cs[1]), ptr(ga), ptr(oks), ptr(cks), ptr(hks),
                                     ptr(ws), B, H, G, stream()))
        torch.cuda.synchronize()
        runs.append(hks.clone())
    print({"bitwise equal over launches": all(torch.equal(r, runs[0]) for r in runs)})
    n = 400

    def chain(split):
        for i in range(n):
            a = (ptr(dgs[i & 1]), ptr(wt), ptr(dy), ptr(dcs[i & 1]), ptr(cs[0]), ptr(cs[1]), ptr(ga), ptr(dgs[1 - (i & 1)]), ptr(dcs[1 - (i & 1)]), None)
            if split:
                check(L.blm_lstm_step_bwd_ks(*a, ptr(ws), B, H, G, stream()))
            else:
                check(L.blm_lstm_step_bwd(*a, B, H, stream()))
    out = {}
    for name, split in (("16x16", False), ("split", True)):
        chain(split)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            chain(split)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        out[name] = round(best, 2)
    print({"B": B, "H": H, "us_per_step": out})


if __name__ == "__main__":
    main()

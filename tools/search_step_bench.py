"""us per launch of the architecture-search LSTM step kernels (blm_lstm_search_step_fwd / _bwd: two gate sets per step,
model_search_bayes.py:686-710) as a DEPENDENT chain on one stream, at B (default 64) x H (default 1024).  Run it under
BLM_LSTM_PIPE=0|1 to compare the pipelined (one workgroup per CU) and the plain (two per CU) forms; B=32 shows one round of the
forward grid.  usage: [B=64] [H=1024] search_step_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402


def main():
    B, H = int(os.environ.get("B", "64")), int(os.environ.get("H", "1024"))
    dev = "cuda"
    torch.manual_seed(0)
    L = lib()
    n = 400
    xw = torch.randn(B, 8 * H, device=dev)
    w8 = torch.randn(8 * H, H, device=dev) * 0.03
    w8t = w8.t().contiguous()
    probs = torch.softmax(torch.randn(4, 2, device=dev), -1).contiguous()
    hs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    cs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    acts = torch.rand(B, 8 * H, device=dev)
    dz = [torch.randn(B, 8 * H, device=dev) * 0.01 for _ in range(2)]
    dc = [torch.randn(B, H, device=dev) * 0.01 for _ in range(2)]
    dy = torch.randn(B, H, device=dev) * 0.01
    part = torch.empty(int(L.blm_lstm_search_step_partials(B, H)), device=dev)

    def fwd():
        for i in range(n):
            check(L.blm_lstm_search_step_fwd(ptr(xw), ptr(w8), ptr(hs[i & 1]), ptr(cs[i & 1]), ptr(probs), ptr(hs[1 - (i & 1)]),
                                             ptr(cs[1 - (i & 1)]), ptr(acts), B, H, stream()))

    def bwd():
        for i in range(n):
            check(L.blm_lstm_search_step_bwd(ptr(dz[i & 1]), ptr(w8t), ptr(dy), ptr(dc[i & 1]), ptr(cs[0]), ptr(cs[1]), ptr(acts),
                                             ptr(probs), ptr(dz[1 - (i & 1)]), ptr(dc[1 - (i & 1)]), ptr(part), B, H, stream()))
    out = {}
    for name, fn in (("search_step_fwd", fwd), ("search_step_bwd", bwd)):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / n)
        out[name] = round(best, 2)
    floor = 2.0 * B * 8 * H * H / 157.3e12 * 1e6
    print({"B": B, "H": H, "pipe": os.environ.get("BLM_LSTM_PIPE", "1"), **out, "mfma_floor_us": round(floor, 2)})


if __name__ == "__main__":
    main()

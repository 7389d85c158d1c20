#!/usr/bin/env python3
"""Can the tied decoder's weight-gradient product hide the LSTM backward recurrence?  The decoder wgrad (TN, 33000 x 1024 over
M = T*B rows) does not feed the recurrent layers' backward, and the fused step kernels keep the matrix pipe busy for 0.19 of
their span (profiles/r05_pmc_lstm_step_kernels.txt) -- so on paper the chain could run UNDER the product.  Measured here at the
configs[1] shape (B 64, H 1024, T 35, V 33000): the backward chain of one layer alone, the product alone, both on one stream
(today), and the chain on a high-priority stream beside the product on a second stream, for every tile the planner has.
usage: chain_gemm_overlap_probe.py [B] [T] [V]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 35
    V = int(sys.argv[3]) if len(sys.argv) > 3 else 33000
    H = 1024
    M = T * B
    dev = "cuda"
    torch.manual_seed(0)
    w = torch.randn(4 * H, H, device=dev) * 0.03
    wt = w.t().contiguous()
    cs = [torch.randn(B, H, device=dev) * 0.1 for _ in range(2)]
    ga = torch.rand(B, 4 * H, device=dev)
    dgs = [torch.randn(B, 4 * H, device=dev) * 0.01 for _ in range(2)]
    dcs = [torch.randn(B, H, device=dev) * 0.01 for _ in range(2)]
    dy = torch.randn(B, H, device=dev) * 0.01
    dlog = torch.randn(M, V, device=dev) * 0.01
    hseq = torch.randn(M, H, device=dev) * 0.1
    dw = torch.zeros(V, H, device=dev)
    lb = lib()
    hi = torch.cuda.Stream(priority=-1)
    lo = torch.cuda.Stream(priority=0)
    nchain = 2 * T  # two layers' backward recurrences, back to back

    def chain():
        for i in range(nchain):
            check(lb.blm_lstm_step_bwd(ptr(dgs[i & 1]), ptr(wt), ptr(dy), ptr(dcs[i & 1]), ptr(cs[0]), ptr(cs[1]), ptr(ga),
                                       ptr(dgs[1 - (i & 1)]), ptr(dcs[1 - (i & 1)]), None, B, H, stream()))

    def wgrad():
        ops.gemm(L.GEMM_TN, dlog, hseq, dw, V, H, M, V, H, H, accumulate=True)

    def serial():
        wgrad()
        chain()

    def beside():
        cur = torch.cuda.current_stream()
        hi.wait_stream(cur)
        lo.wait_stream(cur)
        with torch.cuda.stream(lo):
            wgrad()
        with torch.cuda.stream(hi):
            chain()
        cur.wait_stream(hi)
        cur.wait_stream(lo)

    def timed(fn, reps=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1))
        return best

    print("B %d T %d V %d: backward chain of %d steps alone %.1f us" % (B, T, V, nchain, timed(chain)), flush=True)
    print("%-6s %-7s %10s %10s %10s %8s" % ("tile", "slices", "wgrad", "serial", "beside", "saved"))
    for tile, splits in ((0, 0), (11, 1), (12, 1), (21, 1), (22, 1), (28, 1), (11, 2), (22, 2), (28, 2)):
        check(lb.blm_gemm_plan_override(tile, splits), "override")
        try:
            t_w, t_s, t_b = timed(wgrad), timed(serial), timed(beside)
        except Exception as e:  # noqa: BLE001  (a tile that is illegal for this shape)
            print("%-6d %-7d refused: %s" % (tile, splits, str(e)[:80]), flush=True)
            continue
        print("%-6d %-7d %10.1f %10.1f %10.1f %8.1f" % (tile, splits, t_w, t_s, t_b, t_s - t_b), flush=True)
    check(lb.blm_gemm_plan_override(0, 0), "override")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Does a HIP graph shorten the LSTM time loop?  T dependent step launches (blm_lstm_seq_fwd) enqueued on the stream as
today, against the same launches captured once and replayed as one graph.  usage: lstm_graph_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H = 1024
    for T, B in ((35, 64), (100, 32), (35, 20)):
        xw = torch.randn(T, B, 4 * H, device=dev) * 0.1
        w = torch.randn(4 * H, H, device=dev) * 0.03
        hs = torch.zeros(T + 1, B, H, device=dev)
        cs = torch.zeros(T + 1, B, H, device=dev)
        acts = torch.empty(T, B, 4 * H, device=dev)

        def seq():
            L.check(L.lib().blm_lstm_seq_fwd(ops.ptr(xw), ops.ptr(w), ops.ptr(hs), ops.ptr(cs), ops.ptr(acts), None, T, B, H, ops.stream()), "seq")

        def timed(fn, reps=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / reps
        t_stream = timed(seq)
        ref = hs.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            seq()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            seq()
        hs.zero_()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(hs[1:], ref[1:])
        t_graph = timed(g.replay)
        print("T %3d B %2d: stream %.1f us = %.2f us/step | graph replay %.1f us = %.2f us/step | same result: %s"
              % (T, B, t_stream, t_stream / T, t_graph, t_graph / T, same), flush=True)


if __name__ == "__main__":
    main()

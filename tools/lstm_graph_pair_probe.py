#!/usr/bin/env python3
"""Two independent recurrences on two streams, launched from the host step by step (as ops.lstm_stack2 does) against the same
launches captured ONCE into a HIP graph (fork / join inside the capture) and replayed: what does the pair cost when the host is
out of the way?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H = 1024
    side = torch.cuda.Stream()
    for T, B in ((36, 20), (36, 32), (36, 64), (96, 32)):
        def mk():
            return (torch.randn(T, B, 4 * H, device=dev) * 0.1, torch.randn(4 * H, H, device=dev) * 0.03, torch.zeros(T + 1, B, H, device=dev),
                    torch.zeros(T + 1, B, H, device=dev), torch.empty(T, B, 4 * H, device=dev))
        a, b = mk(), mk()

        def seq(s, t0, n):
            xw, w, hs, cs, acts = s
            L.check(L.lib().blm_lstm_seq_fwd(xw.data_ptr() + t0 * B * 16 * H, ops.ptr(w), hs.data_ptr() + t0 * B * 4 * H, cs.data_ptr() + t0 * B * 4 * H,
                                             acts.data_ptr() + t0 * B * 16 * H, None, n, B, H, ops.stream()), "seq")

        def two(chunk=12):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            for t0 in range(0, T, chunk):
                seq(a, t0, chunk)
                with torch.cuda.stream(side):
                    seq(b, t0, chunk)
            main.wait_stream(side)

        def timed(fn, reps=20):
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
        t_stream = timed(two)
        ref = b[2].clone()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            two()
        torch.cuda.current_stream().wait_stream(cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            two()
        b[2].zero_()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(b[2][1:], ref[1:])
        t_graph = timed(g.replay)
        print("T %3d B %2d, two chains: host-launched %.1f us = %.2f us per step PAIR | one graph replay %.1f us = %.2f us per step PAIR | same: %s"
              % (T, B, t_stream, t_stream / T, t_graph, t_graph / T, same), flush=True)


if __name__ == "__main__":
    main()

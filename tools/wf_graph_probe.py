#!/usr/bin/env python3
"""Upper bound of what a host-free issue of the layer wavefront would buy: ops.lstm_stack2's forward (no grad) as the host issues
it today against the same launches captured into one HIP graph and replayed (three streams, events and all)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    H = 1024
    for T, B in ((35, 20), (100, 32), (35, 64)):
        torch.manual_seed(0)
        x = torch.randn(T, B, H, device=dev) * 0.1
        h0 = torch.zeros(2, B, H, device=dev)
        c0 = torch.zeros(2, B, H, device=dev)
        lay = [tuple(t.to(dev) for t in (torch.randn(4 * H, H) * 0.03, torch.randn(4 * H, H) * 0.03, torch.zeros(4 * H), torch.zeros(4 * H))) for _ in range(2)]
        ops.set_lstm_wavefront(True)

        def fwd():
            with torch.no_grad():
                return ops.lstm_stack2(x, h0, c0, lay[0], lay[1])[0]

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
        t_host = timed(fwd)
        ref = fwd().clone()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            fwd()
        torch.cuda.current_stream().wait_stream(cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = fwd()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(y, ref)
        t_graph = timed(g.replay)
        print("T %3d B %2d cmax %s: forward wavefront (incl. layer 1's input GEMM) host-issued %.1f us | graph replay %.1f us | same: %s"
              % (T, B, os.environ.get("BLM_TMP_CMAX", "16"), t_host, t_graph, same), flush=True)


if __name__ == "__main__":
    main()

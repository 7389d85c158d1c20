#!/usr/bin/env python3
"""Is there time to win by running a linear layer's two backward products (dgrad NN, wgrad TN -- independent, both read
dY) side by side instead of one after the other?  For each pair of the headline step: both launches on one stream
(today), and the same two launches on two streams (an upper bound of what a grouped launch could overlap: the tail of one
grid under the ramp of the other).  usage: gemm_pair_probe.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402

PAIRS = [  # name, M (tokens), N_out, K_in of the forward y = x W^T
    ("o_net 512->512", 8192, 512, 512),
    ("qkv 512->1536", 8192, 1536, 512),
    ("ffn1 512->4096", 8192, 4096, 512),
    ("ffn2 4096->512", 8192, 512, 4096),
    ("o_net M=3200", 3200, 512, 512),
    ("qkv M=3200", 3200, 1536, 512),
    ("ffn2 M=3200", 3200, 512, 4096),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    print("%-18s %10s %10s %10s %10s   (us; dgrad, wgrad alone; both on one stream; both on two streams)" % ("pair", "dgrad", "wgrad", "serial", "side by side"))
    for name, M, N, K in PAIRS:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g)
        dy = torch.randn(M, N, device=dev, generator=g)
        dx = torch.zeros(M, K, device=dev)
        dw = torch.zeros(N, K, device=dev)

        def dgrad():
            ops.gemm(L.GEMM_NN, dy, w, dx, M, K, N, N, K, K, accumulate=True)

        def wgrad():
            ops.gemm(L.GEMM_TN, dy, x, dw, N, K, M, N, K, K, accumulate=True)

        def timed(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / reps

        def both():
            dgrad()
            wgrad()

        def side():
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                wgrad()
            dgrad()
            cur.wait_stream(s2)

        t_d, t_w, t_b, t_s = timed(dgrad), timed(wgrad), timed(both), timed(side)
        print("%-18s %10.1f %10.1f %10.1f %10.1f" % (name, t_d, t_w, t_b, t_s), flush=True)


if __name__ == "__main__":
    main()

"""Do the step kernels of two independent recurrences really run side by side on two streams?  The launches come from the
library's own loop (blm_lstm_seq_fwd: CHUNK steps per call, as ops.lstm_stack2 issues them), so the host cost per launch is the
C one, not a ctypes call's.  Prints us per step of ONE chain, and us per step PAIR of two chains on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402


def main():
    H = int(os.environ.get("H", "1024"))
    chunk = int(os.environ.get("CHUNK", "12"))
    rounds = int(os.environ.get("ROUNDS", "30"))
    dev = "cuda"
    L = lib()
    side = torch.cuda.Stream()
    for B in [int(b) for b in os.environ.get("BS", "8,20,32,64").split(",")]:
        torch.manual_seed(0)
        T = chunk * rounds

        def mk():
            return dict(xw=torch.randn(T, B, 4 * H, device=dev), w=torch.randn(4 * H, H, device=dev) * 0.03,
                        hs=torch.zeros(T + 1, B, H, device=dev), cs=torch.zeros(T + 1, B, H, device=dev),
                        ga=torch.empty(T, B, 4 * H, device=dev))
        a, b = mk(), mk()
        bh, bg = B * H * 4, B * 4 * H * 4

        def seq(s, t0, n):
            check(L.blm_lstm_seq_fwd(s["xw"].data_ptr() + t0 * bg, ptr(s["w"]), s["hs"].data_ptr() + t0 * bh, s["cs"].data_ptr() + t0 * bh,
                                     s["ga"].data_ptr() + t0 * bg, None, n, B, H, stream()))

        def one():
            for r in range(rounds):
                seq(a, r * chunk, chunk)

        def two():
            side.wait_stream(torch.cuda.current_stream())
            for r in range(rounds):
                seq(a, r * chunk, chunk)
                with torch.cuda.stream(side):
                    seq(b, r * chunk, chunk)
            torch.cuda.current_stream().wait_stream(side)
        out = {}
        for name, fn in (("one", one), ("two", two)):
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / T * 1000)
            out[name] = best
        print("B=%d H=%d, %d-step calls: one chain %.2f us/step; two chains on two streams %.2f us per step PAIR (%.2f x one)"
              % (B, H, chunk, out["one"], out["two"], out["two"] / out["one"]))


if __name__ == "__main__":
    main()

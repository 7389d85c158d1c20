"""us per launch of blm_lstm_step_fwd (cfg2 shape B=64 H=1024), back-to-back on one stream."""
import sys
import torch
sys.path.insert(0, ".")
from bayeslms_amd._lib import lib, check, ptr, stream

def main():
    B, H = 64, 1024
    dev = "cuda"
    xw = torch.randn(B, 4 * H, device=dev)
    w = torch.randn(4 * H, H, device=dev) * 0.03
    hp, cp = torch.randn(B, H, device=dev), torch.randn(B, H, device=dev)
    hs = [torch.empty(B, H, device=dev) for _ in range(2)]
    c, ga = torch.empty(B, H, device=dev), torch.empty(B, 4 * H, device=dev)
    L = lib()
    n = 200
    def run():
        for i in range(n):
            check(L.blm_lstm_step_fwd(ptr(xw), ptr(w), ptr(hp), ptr(cp), ptr(hs[i & 1]), ptr(c), ptr(ga), None, B, H, stream()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    print(f"lstm_step_fwd B={B} H={H}: {e0.elapsed_time(e1) / n * 1000:.2f} us/launch")

if __name__ == "__main__":
    main()

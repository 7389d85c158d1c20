"""Host cost of the library's launch loops: wall time of blm_lstm_seq_fwd (N launches from one call) WITHOUT waiting for the device."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayeslms_amd._lib import check, lib, ptr, stream  # noqa: E402
B, H, T = 20, 1024, 360
dev = "cuda"
L = lib()
xw = torch.randn(T, B, 4 * H, device=dev); w = torch.randn(4 * H, H, device=dev) * 0.03
hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev); ga = torch.empty(T, B, 4 * H, device=dev)
for n in (12, 36, 360):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(T // n):
        check(L.blm_lstm_seq_fwd(xw.data_ptr() + r * n * B * 4 * H * 4, ptr(w), hs.data_ptr() + r * n * B * H * 4, cs.data_ptr() + r * n * B * H * 4,
                                 ga.data_ptr() + r * n * B * 4 * H * 4, None, n, B, H, stream()))
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("seq_fwd in calls of %3d steps: host %.2f us per launch, device done after %.2f us per step" % (n, 1e6 * host / T, 1e6 * tot / T))

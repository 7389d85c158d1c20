#!/usr/bin/env python3
"""Do the workgroups that share a CU progress at the same rate?  (HBM-traffic study, VERDICT r2 #4.)
Library built with -DBLM_GEMM_PROF (make -C bayeslms_amd/csrc EXTRA=-DBLM_GEMM_PROF OBJDIR=../../build/obj_prof
LIB=../libbayeslm_hip_prof.so; BLM_LIB selects it): every workgroup of the roofline GEMM (NT 8192 x 512 x K) stamps its
entry and the end of its K loop with the 100 MHz wall clock and its CU.  Printed per tile: the kernel's span, and for the
CUs that hold two or more workgroups the K-loop end of the FIRST and of the LAST of them as a fraction of the span."""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(L.LIB_PATH)
    M, N = 8192, 512
    for K in (1024, 4096, 8192):
        A, B, Cm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        for tile in (22, 21, 11):
            L.check(L.lib().blm_gemm_plan_override(tile, 1), "override")
            for _ in range(3):
                ops.gemm(L.GEMM_NT, A, B, Cm, M, N, K, K, K, N)
            torch.cuda.synchronize()
            nwg = {22: 256, 21: 512, 11: 1024}[tile]
            buf = (ctypes.c_longlong * (4 * nwg))()
            assert lib.blm_debug_wg_life_nt(buf, nwg) == 0
            st = [buf[4 * i] for i in range(nwg)]
            en = [buf[4 * i + 1] for i in range(nwg)]
            cu = [((buf[4 * i + 2] >> 32) & 15, (buf[4 * i + 2] >> 13) & 7, (buf[4 * i + 2] >> 12) & 1, (buf[4 * i + 2] >> 8) & 15) for i in range(nwg)]
            t0, t1 = min(st), max(en)
            span = (t1 - t0) * 0.01
            per = defaultdict(list)
            for i in range(nwg):
                per[cu[i]].append(((st[i] - t0) / (t1 - t0), (en[i] - t0) / (t1 - t0)))
            counts = defaultdict(int)
            for v in per.values():
                counts[len(v)] += 1
            firsts = sorted(min(e for _, e in v) for v in per.values() if len(v) > 1)
            lasts = sorted(max(e for _, e in v) for v in per.values() if len(v) > 1)
            starts = sorted(s for v in per.values() for s, _ in v)
            med = lambda x: x[len(x) // 2] if x else float("nan")  # noqa: E731
            print("K %5d tile %d: span %.1f us, %d CUs used, workgroups per CU %s; entry: median %.3f max %.3f of the span; "
                  "K loop ends, CUs with >= 2 workgroups: first done at median %.3f (min %.3f), last at median %.3f"
                  % (K, tile, span, len(per), dict(sorted(counts.items())), med(starts), starts[-1], med(firsts),
                     firsts[0] if firsts else float("nan"), med(lasts)), flush=True)
    L.check(L.lib().blm_gemm_plan_override(0, 0), "override")


if __name__ == "__main__":
    main()

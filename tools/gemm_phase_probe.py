#!/usr/bin/env python3
"""How much of a short-K GEMM's time does the matrix pipe of a CU sit idle because every workgroup on it is in its
epilogue (or not yet in its K loop)?  Library built with -DBLM_GEMM_LIFE (make -C bayeslms_amd/csrc EXTRA=-DBLM_GEMM_LIFE OBJDIR=../../build/obj_life LIB=../libbayeslm_hip_life.so); every workgroup
stamps entry, end of K loop and end of epilogue.  Per launch: span, and over the CUs the mean fraction of the span during
which NO resident workgroup is inside its K loop, split into head (before the first K loop starts), interior and tail."""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bayeslms_amd import _lib as L, ops  # noqa: E402

CASES = [  # name, op, M, N, K, epilogue, tiles to try
    ("ffn1 fwd  NT 8192x4096x512 +bias+GELU", "NT", 8192, 4096, 512, L.EPI_BIAS_GELU, (12, 11, 22)),
    ("ffn2 dgrad NN 8192x4096x512 *dGELU", "NN", 8192, 4096, 512, L.EPI_MUL_DGELU, (12, 11, 22)),
    ("qkv fwd   NT 8192x1536x512 +bias", "NT", 8192, 1536, 512, L.EPI_BIAS, (12, 11)),
    ("o_net fwd NT 8192x512x512 +bias", "NT", 8192, 512, 512, L.EPI_BIAS, (11,)),
    ("ffn2 fwd  NT 8192x512x4096 +bias", "NT", 8192, 512, 4096, L.EPI_BIAS, (28,)),
]
TILE_DIMS = {11: (64, 64), 12: (64, 128), 21: (128, 64), 22: (128, 128), 28: (128, 128)}


def main():
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(L.LIB_PATH)
    g = torch.Generator(device=dev).manual_seed(3)
    for name, op, M, N, K, epi, tiles in CASES:
        if op == "NT":
            A, B, code, lda, ldb = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g), L.GEMM_NT, K, K
        else:
            A, B, code, lda, ldb = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g), L.GEMM_NN, K, N
        Cm = torch.empty(M, N, device=dev)
        aux = torch.randn(M, N, device=dev, generator=g) if epi in (L.EPI_BIAS_GELU, L.EPI_MUL_DGELU) else None
        bias = torch.randn(N, device=dev, generator=g) if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU) else None
        for tile in tiles:
            bm, bn = TILE_DIMS[tile]
            nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
            if nwg > 8192:
                continue
            L.check(L.lib().blm_gemm_plan_override(tile, 1), "override")
            for _ in range(3):
                ops.gemm(code, A, B, Cm, M, N, K, lda, ldb, N, epilogue=epi, bias=bias, aux=aux)
            torch.cuda.synchronize()
            buf = (ctypes.c_longlong * (4 * nwg))()
            assert getattr(lib, "blm_debug_wg_life_" + op.lower())(buf, nwg) == 0
            st = [buf[4 * i] for i in range(nwg)]
            ke = [buf[4 * i + 1] for i in range(nwg)]
            en = [buf[4 * i + 3] for i in range(nwg)]
            cu = [((buf[4 * i + 2] >> 32) & 15, (buf[4 * i + 2] >> 13) & 7, (buf[4 * i + 2] >> 12) & 1, (buf[4 * i + 2] >> 8) & 15) for i in range(nwg)]
            t0, t1 = min(st), max(en)
            span = float(t1 - t0)
            per = defaultdict(list)
            for i in range(nwg):
                per[cu[i]].append((st[i], ke[i], en[i]))
            head = inner = tail = 0.0
            kdur = sum(k - s for s, k, _ in zip(st, ke, en)) / nwg * 0.01
            edur = sum(e - k for _, k, e in zip(st, ke, en)) / nwg * 0.01
            for v in per.values():
                iv = sorted((s, k) for s, k, _ in v)
                cur_s, cur_e = iv[0]
                head += (cur_s - t0)
                busy_end = cur_e
                for s, k in iv[1:]:
                    if s > busy_end:
                        inner += s - busy_end
                    busy_end = max(busy_end, k)
                tail += t1 - busy_end
            n = len(per)
            print("%-40s tile %d: %5d workgroups on %d CUs, span %.1f us; a workgroup: K loop %.1f us, epilogue %.1f us; "
                  "no K loop running on the CU: head %.3f + interior %.3f + tail %.3f = %.3f of the span"
                  % (name, tile, nwg, n, span * 0.01, kdur, edur, head / n / span, inner / n / span, tail / n / span,
                     (head + inner + tail) / n / span), flush=True)
            if os.environ.get("PHASE_TIMELINE"):
                v = sorted(per[sorted(per)[len(per) // 2]])
                print("      one CU: " + " ".join("[%.1f %.1f %.1f]" % ((a - t0) * 0.01, (b - t0) * 0.01, (c - t0) * 0.01) for a, b, c in v[:18]))
    L.check(L.lib().blm_gemm_plan_override(0, 0), "override")


if __name__ == "__main__":
    main()

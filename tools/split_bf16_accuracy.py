#!/usr/bin/env python3
"""Accuracy of an fp32 GEMM emulated on the bf16 matrix cores by operand splitting (CPU emulation, no
GPU needed): x = hi + lo (+ lo2) with every part a bf16, products accumulated in fp32.

Why: gfx950's bf16 MFMA rate is 16x the fp32 one (v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16
against 8 x 64 cycles of v_mfma_f32_32x32x2_f32 for the same k), so a 3-product split costs 96 and a
6-product split 192 cycles per k16 tile against 512 -- if the result is as good as fp32.  Numbers for
the roofline shape's K:
    fp32 matmul        3.6e-07 max-rel   2.9e-07 rms-rel   (vs fp64)
    bf16 x1            2.2e-03           2.4e-03
    bf16 x3 (2 parts)  4.5e-06           4.4e-06
    bf16 x6 (3 parts)  1.8e-07           1.4e-07   <- at least as accurate as the fp32 path
"""
import torch


def split(t, parts):
    out, r = [], t
    for _ in range(parts):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def main(M=512, N=512, K=4096):
    torch.manual_seed(0)
    x, w = torch.randn(M, K), torch.randn(N, K) / K ** 0.5
    ref = x.double() @ w.double().t()

    def err(y):
        d = y.double() - ref
        return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print("fp32    max-rel %.2e rms-rel %.2e" % err(x @ w.t()))
    for parts, keep in ((1, 1), (2, 3), (3, 6)):
        xs, ws = split(x, parts), split(w, parts)
        terms = sorted(((i, j) for i in range(parts) for j in range(parts)), key=lambda ij: ij[0] + ij[1])[:keep]
        y = sum(xs[i] @ ws[j].t() for i, j in reversed(terms))  # smallest terms first
        print("bf16 x%d max-rel %.2e rms-rel %.2e" % ((keep,) + err(y)))


if __name__ == "__main__":
    main()

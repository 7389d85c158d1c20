#!/usr/bin/env python3
"""Fits the constants of the GEMM cost model (bayeslms_amd/csrc/gemm_plan.hip) to the stand-alone tile x split sweep
written by `tools/gemm_tune.py --grid` (gpurun_out/gemm_grid.jsonl) and reports how good the model's PICK is: for every
shape of the grid, measured time of the plan the fitted model would choose / measured time of the best candidate.
Runs on the CPU (numpy + scipy); prints the constants as the C++ initialiser of ModelK."""
import json
import sys

import numpy as np
from scipy.optimize import least_squares

TILES = {11: (1, 1, 5), 12: (1, 2, 3), 21: (2, 1, 3), 22: (2, 2, 2), 28: (2, 2, 2)}  # 28: 128x128 on eight waves
TIDX = {11: 0, 12: 1, 21: 2, 22: 3, 28: 4}
NTILE = 5


def unpack(x):
    p = {}
    n = NTILE
    p["einf"] = x[0:3 * n].reshape(3, n)
    o = 3 * n
    p["a"] = x[o:o + n]
    p["t0"] = x[o + n:o + 2 * n]
    p["t0r"] = x[o + 2 * n:o + 3 * n]
    o += 3 * n
    p["launch"], p["atomic"], p["memset_us"], p["memset_bpus"], p["hbm"] = x[o:o + 5]
    p["t0o"] = x[o + 5:o + 5 + n]
    return p


N5 = NTILE
X0 = np.concatenate([np.full(3 * N5, 0.9), np.full(N5, 0.3), np.full(N5, 3.0), np.full(N5, 1.0), [2.0, 1.6e6, 2.0, 3.0e6, 4.0e6], np.full(N5, 0.5)])
LO = np.concatenate([np.full(3 * N5, 0.3), np.full(N5, 0.0), np.full(N5, 0.0), np.full(N5, 0.0), [0.5, 1e5, 0.0, 5e5, 1e6], np.full(N5, 0.0)])
HI = np.concatenate([np.full(3 * N5, 1.2), np.full(N5, 3.0), np.full(N5, 30.0), np.full(N5, 30.0), [8.0, 2e7, 20.0, 2e7, 8e6], np.full(N5, 10.0)])
CYC = 2400.0  # nominal cycles per microsecond; the efficiencies absorb the clock under load


def model(p, op, M, N, K, acc, tile, S):
    """vectorised over numpy arrays (tile as index 0..3)"""
    wtm = np.array([1, 1, 2, 2, 2])[tile]
    wtn = np.array([1, 2, 1, 2, 2])[tile]
    occ = np.array([5, 3, 3, 2, 2])[tile]
    BM, BN = 64 * wtm, 64 * wtn
    tiles = np.ceil(M / BM) * np.ceil(N / BN)
    G = tiles * S
    kt = np.ceil(np.ceil(K / S) / 32)
    tk = kt * 1024.0 * wtm * wtn / CYC
    slots = 256 * occ
    full = np.floor(G / slots)
    rem = G - full * slots
    o_r = np.ceil(rem / 256)
    einf = p["einf"][op, tile]
    a = p["a"][tile]

    def e(o):
        return einf * o / (o + a)
    t_full = occ * tk / e(occ) + p["t0o"][tile] * occ
    us = p["launch"] + np.where(full > 0, p["t0"][tile] + t_full + (full - 1) * (p["t0r"][tile] + t_full), 0.0)
    us = us + np.where(rem > 0, np.where(full > 0, p["t0r"][tile], p["t0"][tile]) + o_r * tk / e(np.maximum(o_r, 1)) + p["t0o"][tile] * o_r, 0.0)
    bytes_ = 4.0 * (M * K + N * K + M * N * np.where(acc > 0, 2.0, 1.0))
    mem = p["launch"] + bytes_ / p["hbm"]
    us = (us ** 3 + mem ** 3) ** (1.0 / 3.0)  # soft maximum: near the streaming bound the plan still matters
    us = us + np.where(S > 1, S * M * N * 4.0 / p["atomic"] + np.where(acc > 0, 0.0, p["memset_us"] + M * N * 4.0 / p["memset_bpus"]), 0.0)
    return us


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_grid.jsonl"
    rows = [json.loads(ln) for ln in open(path)]
    op, M, N, K, acc, tile, S, us, sid = [], [], [], [], [], [], [], [], []
    for i, r in enumerate(rows):
        for lab, v in r["us"].items():
            t, s = lab.split("/")
            if int(t) == 28 and r["K"] % 32 != 0:
                continue
            op.append(r["op"]); M.append(r["M"]); N.append(r["N"]); K.append(r["K"]); acc.append(r["acc"])
            tile.append(TIDX[int(t)]); S.append(int(s)); us.append(v); sid.append(i)
    op, tile, sid = np.array(op), np.array(tile), np.array(sid)
    M, N, K, acc, S, us = (np.array(v, dtype=np.float64) for v in (M, N, K, acc, S, us))
    # weight: what matters is ranking near the optimum -> emphasise candidates within 2x of their shape's best
    best = np.zeros(len(rows))
    for i in range(len(rows)):
        best[i] = us[sid == i].min() if i % 1 == 0 else 0
    w = np.where(us <= 1.5 * best[sid], 1.0, 0.3)

    def resid(x):
        return w * np.log(model(unpack(x), op, M, N, K, acc, tile, S) / us)
    sol = least_squares(resid, X0, bounds=(LO, HI), loss="soft_l1", f_scale=0.1, max_nfev=200)
    p = unpack(sol.x)
    pred = model(p, op, M, N, K, acc, tile, S)
    err = np.abs(np.log(pred / us))
    print("fit: %d points, median |log err| %.3f, 90th pct %.3f" % (len(us), np.median(err), np.percentile(err, 90)))
    # quality of the pick
    fracs = []
    worst = []
    for i, r in enumerate(rows):
        m = sid == i
        j = np.argmin(pred[m] * (1.0 + 0.002 * S[m]))
        f = us[m].min() / us[m][j]
        fracs.append(f)
        worst.append((f, r["op"], r["M"], r["N"], r["K"]))
    fracs = np.array(fracs)
    print("pick quality (best time / picked time): min %.3f, 1st pct %.3f, 10th pct %.3f, median %.3f, mean %.3f"
          % (fracs.min(), np.percentile(fracs, 1), np.percentile(fracs, 10), np.median(fracs), fracs.mean()))
    for f in sorted(worst)[:8]:
        print("   worst:", f)
    big = np.array([us[sid == i].min() >= 30.0 for i in range(len(rows))])
    print("shapes whose best candidate takes >= 30 us (%d): min %.3f, 1st pct %.3f, 10th pct %.3f, mean %.3f"
          % (big.sum(), fracs[big].min(), np.percentile(fracs[big], 1), np.percentile(fracs[big], 10), fracs[big].mean()))
    for f in sorted(w_ for w_, b in zip(worst, big) if b)[:8]:
        print("   worst >= 30 us:", f)
    np.set_printoptions(precision=4, suppress=True)
    print("ModelK g_model = {\n    %.1f,\n    {%s},\n    {%s},\n    {%s},\n    {%s},\n    %.3f,\n    %.4g,\n    %.4g, %.3f,\n    %.4g,\n    {%s},\n};"
          % (CYC, ", ".join("{" + ", ".join("%.4f" % v for v in row) + "}" for row in p["einf"]),
             ", ".join("%.4f" % v for v in p["a"]), ", ".join("%.3f" % v for v in p["t0"]), ", ".join("%.3f" % v for v in p["t0r"]),
             p["launch"], p["atomic"], p["memset_bpus"], p["memset_us"], p["hbm"], ", ".join("%.3f" % v for v in p["t0o"])))


if __name__ == "__main__":
    main()

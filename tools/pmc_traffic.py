"""HBM bytes per launch of the roofline kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
collected separately), corrected as MI355X_MICROARCH.md prescribes for gfx950:
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Usage:
  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
The kernel is the Bayesian FFN linear2 forward GEMM (NT 8192 x 512 x 4096); its instantiation and grid are taken from the
library's planner (round 3: 128x64 tiles, 512 workgroups); the six linear2 forwards of a step (one sampled, five with bias)
share the shape.  KERNEL=... GRID=... in the environment select another instantiation."""
import csv
import json
import os
import sys

def _roofline_launch():
    """Kernel instantiation and grid of the roofline launch (NT 8192 x 512 x 4096, plain epilogue) as the library's own
    planner picks them (blm_gemm_plan_query: host code, no GPU)."""
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bayeslms_amd import _lib as L
    a = L.GemmArgs()
    a.abi_version, a.op, a.M, a.N, a.K, a.lda, a.ldb, a.ldc = L.ABI_VERSION, L.GEMM_NT, 8192, 512, 4096, 4096, 4096, 512
    pl = L.GemmPlan()
    L.check(L.lib().blm_gemm_plan_query(C.byref(a), C.byref(pl)), "blm_gemm_plan_query")
    # tile code -> (WTM, WTN, wave-grid columns): 28 is the 128 x 128 tile on eight waves of 64 x 32
    wtm, wtn, wgn = {11: (1, 1, 2), 12: (1, 2, 2), 21: (2, 1, 2), 22: (2, 2, 2), 28: (2, 1, 4)}[pl.tile]
    bm, bn = 64 * wtm, 32 * wtn * wgn
    nwg = ((8192 + bm - 1) // bm) * ((512 + bn - 1) // bn) * pl.splits
    return "void blm::gemm_f32_kernel<0, %d, %d, false, true, 0, %d>" % (wtm, wtn, wgn), str(128 * wgn * nwg)


KERNEL, GRID = os.environ.get("KERNEL"), os.environ.get("GRID")  # another instantiation (A/B of two tiles)
if not KERNEL or not GRID:
    KERNEL, GRID = _roofline_launch()
MIN_US = float(os.environ.get("MIN_US", "150"))  # the o_net forward (8192 x 512 x 512, ~40 us) shares kernel AND grid with the
# roofline launch (8192 x 512 x 4096, ~250 us): the two are told apart by their duration


def _is(r):
    return (r["Kernel_Name"].startswith(KERNEL) and r["Grid_Size"] == GRID
            and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 >= MIN_US)


def avg(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if _is(r) and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)


def mfma_util(path, out):
    """Third pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE): matrix-pipe utilisation of the same launches."""
    rows = [r for r in csv.DictReader(open(path)) if _is(r)]
    busy = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES"]
    act = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    b, a = sum(busy) / len(busy), sum(act) / len(act)
    res = {"kernel": KERNEL + ", ...>", "grid_threads": int(GRID), "launches": len(busy), "SQ_VALU_MFMA_BUSY_CYCLES": b,
           "GRBM_GUI_ACTIVE_sum_over_8_XCDs": a, "mfma_pipe_utilisation": b / ((a / 8) * 256 * 4),
           "note": "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE/8) * 256 CUs * 4 SIMDs); expected busy cycles for "
                   "34.36 GFLOP at 64 cycles per v_mfma_f32_32x32x2_f32 = 5.369e+08"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


def main():
    if sys.argv[1] == "mfma":
        return mfma_util(sys.argv[2], sys.argv[3])
    f, nf = avg(sys.argv[1], "FETCH_SIZE")
    try:
        w, nw = avg(sys.argv[2], "WRITE_SIZE")
    except ZeroDivisionError:  # the WRITE pass belongs to another tile's run: WRITE_SIZE of this launch is exactly Y on every tile
        w, nw = 8192 * 512 * 4 / 1024.0, nf
    M, N, K = 8192, 512, 4096
    out = {"kernel": KERNEL, "grid_threads": int(GRID), "launches": min(nf, nw), "FETCH_SIZE_KB_avg": f,
           "WRITE_SIZE_KB_avg": w, "traffic_bytes_per_launch": (2 * f + w) * 1024,
           "rule": "MI355X_MICROARCH.md HBM: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950; separate --pmc passes",
           "algorithmic_bytes": 4 * (M * K + N * K + M * N)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()

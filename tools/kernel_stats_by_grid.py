"""Per (kernel, grid size) duration summary of a rocprofv3 --kernel-trace CSV.  The GEMM template is
one symbol for many shapes, so the plain --stats table averages them together; this table keeps the
launches of one shape apart (the roofline kernel of bench.py -- Bayesian FFN linear2 forward,
M=8192 N=512 K=4096 -- is gemm_f32_kernel<0, 2, 1, false, true> with 131072 threads = 512 tiles of 128x64).
Usage: python tools/kernel_stats_by_grid.py <kernel_trace.csv> <out.csv>"""
import collections
import csv
import sys


def main():
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        agg[(r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for _, v in rows)
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridThreads", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for (name, grid), v in rows:
            w.writerow([name, grid, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 2), min(v), max(v)])


if __name__ == "__main__":
    main()

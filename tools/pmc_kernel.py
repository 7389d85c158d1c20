#!/usr/bin/env python3
"""Mean of every collected counter over the launches of one kernel (name substring + total grid threads, optional minimum
duration in us) in a rocprofv3 --pmc counter_collection.csv.  usage: pmc_kernel.py FILE.csv "kernel substring" GRID [min_us]"""
import csv
import sys
from collections import defaultdict


def main():
    path, sub, grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
    min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]:
            continue
        g = int(r.get("Grid_Size", 0) or 0) or int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        if g != grid:
            continue
        if min_us and "Start_Timestamp" in r and (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) < 1e3 * min_us:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
    for k in sorted(acc):
        print("%-34s %16.1f  (mean of %d launches)" % (k, acc[k] / n[k], n[k]))


if __name__ == "__main__":
    main()

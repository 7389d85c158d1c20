#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 kernel trace (steps are delimited by the clip+SGD launch):
runs of the same kernel are folded, with start offset, summed duration and the idle gaps in front of them.
usage: trace_step.py <kernel_trace.csv> [min_us]"""
import csv
import re
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "clip_sgd" in r["Kernel_Name"]]
    step = rows[idx[-2] + 1:idx[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    span = (int(step[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
    print("step: %.1f us, %d kernels, busy %.1f us" % (span, len(step), busy))

    def short(n):
        return re.sub(r"\(.*", "", n.replace("at::native::", ""))[:72]
    out, prev = [], None
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append((short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
    i = 0
    while i < len(out):
        j = i
        while j + 1 < len(out) and out[j + 1][0] == out[i][0]:
            j += 1
        d, g = sum(o[2] for o in out[i:j + 1]), sum(o[3] for o in out[i:j + 1])
        if d + g >= floor:
            print("%8.1f  %-72s x%3d  %8.1f us  gaps %6.1f" % (out[i][1], out[i][0], j - i + 1, d, g))
        i = j + 1


if __name__ == "__main__":
    main()

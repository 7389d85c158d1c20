#!/usr/bin/env python3
"""GPU busy fraction of a traced run: union of the kernel intervals / span, over the last `frac` of the trace (steady state).
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o x -- python3 tools/run_workload.py cfg1 30; gpu_busy.py DIR [frac]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
    t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * (1.0 - frac)
    iv = [x for x in iv if x[0] >= t_lo]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    gaps.sort()
    print("%d kernels over %.2f ms: GPU busy %.3f; %d gaps, median %.2f us, 90th pct %.2f us, sum of gaps > 5 us: %.2f ms"
          % (len(iv), span / 1e6, busy / span, len(gaps), gaps[len(gaps) // 2] / 1e3 if gaps else 0, gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0,
             sum(g for g in gaps if g > 5000) / 1e6))


if __name__ == "__main__":
    main()

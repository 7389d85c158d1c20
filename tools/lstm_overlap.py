"""How much of the two-stream layer wavefront's step kernels really run side by side?  Reads a rocprofv3 --kernel-trace csv and,
for the launches whose name contains argv[2] (default lstm_step_fwd_kernel), prints per queue: launches, mean duration, mean gap
to the previous launch of the same queue; and over all of them: sum of durations, length of the union of their busy intervals,
time during which two of them are in flight."""
import csv
import sys
from collections import defaultdict


def main():
    f = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "lstm_step_fwd_kernel"
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    by_q = defaultdict(list)
    iv = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        by_q[r.get("Queue_Id", "?")].append((s, e))
        iv.append((s, e))
    for q, v in sorted(by_q.items()):
        v.sort()
        d = [e - s for s, e in v]
        gaps = [v[i][0] - v[i - 1][1] for i in range(1, len(v)) if 0 <= v[i][0] - v[i - 1][1] < 20000]
        print("queue %s: %d launches, mean duration %.2f us, mean gap to the previous one %.2f us (%d gaps under 20 us)"
              % (q, len(v), sum(d) / len(d) / 1e3, (sum(gaps) / max(len(gaps), 1)) / 1e3, len(gaps)))
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last, busy1, busy2 = 0, None, 0, 0
    for t, k in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += k
        last = t
    tot = sum(e - s for s, e in iv)
    print("%s: %d launches, sum of durations %.1f us, union of busy time %.1f us, two or more in flight %.1f us (%.0f %% of the union)"
          % (pat, len(iv), tot / 1e3, busy1 / 1e3, busy2 / 1e3, 100.0 * busy2 / max(busy1, 1)))


if __name__ == "__main__":
    main()

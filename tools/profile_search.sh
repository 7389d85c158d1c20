#!/bin/bash
# Architecture-search path profile set (run on the GPU box from the repo root): kernel trace + stats of both search windows
# (tools/search_workload.py = bench.search_leg) and separate PMC passes over the LSTM search window's fused step kernels and the
# Transformer window's mix2 kernels.  Output: gpurun_out/prof_search_$1/
set -e
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_search_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for k in lstm tlm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$k -o wl -- python3 tools/search_workload.py $k 8 > $OUT/trace_$k.log 2>&1
  cp $(find $OUT/trace_$k -name "*kernel_stats.csv" | head -1) $OUT/search_${k}_kernel_stats.csv
done
for k in lstm tlm; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${k}_mfma -o wl -- python3 tools/search_workload.py $k 3 > $OUT/${k}_mfma.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/${k}_wait -o wl -- python3 tools/search_workload.py $k 3 > $OUT/${k}_wait.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${k}_fetch -o wl -- python3 tools/search_workload.py $k 3 > $OUT/${k}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${k}_write -o wl -- python3 tools/search_workload.py $k 3 > $OUT/${k}_write.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def means(d, sub):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    if not f:
        return {}, 0
    for r in csv.DictReader(open(f[0])):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, max(n.values()) if n else 0
with open(out + "/pmc_search_kernels.txt", "w") as g:
    g.write("# rocprofv3 --pmc (separate passes) over tools/search_workload.py: per-launch means\n")
    for wl, subs in (("lstm", ("lstm_step_fwd_kernel", "lstm_step_bwd_kernel")), ("tlm", ("mix2_bwd_kernel", "mix2_fwd_kernel"))):
        for sub in subs:
            c, n = {}, 0
            for d in ("mfma", "wait", "fetch", "write"):
                m, k = means(wl + "_" + d, sub)
                c.update(m); n = max(n, k)
            g.write("== %s window: %s (%d launches)\n" % (wl, sub, n))
            for k in sorted(c):
                g.write("%-34s %16.1f\n" % (k, c[k]))
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                g.write("%-34s %16.1f  ((2 FETCH + WRITE) x 1024, MI355X_MICROARCH.md)\n" % ("hbm_bytes_per_launch", (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
                g.write("%-34s %16.4f\n" % ("mfma_busy_frac_of_simd_cycles", c["SQ_VALU_MFMA_BUSY_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8.0) * 1024)))
            if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
                g.write("%-34s %16.4f\n" % ("wait_frac_of_wave_cycles", c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete
ls $OUT

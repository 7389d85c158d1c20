#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the headline bench, and three
# separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/prof_$1/
set -e
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --gpus 1 --no-cpu-baseline --no-opt-in --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $B --steps 4 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $B --steps 4 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o bench -- $B --steps 4 --warmup 1 > $OUT/mfma.log 2>&1
# summaries for profiles/: kernel stats, stats by (kernel, grid), traffic / matrix-pipe utilisation of the roofline kernel
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1); M=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/pmc_sampled_gemm_fwd.json
python3 tools/pmc_traffic.py mfma $M $OUT/pmc_sampled_gemm_fwd_mfma_util.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_cfg3_kernel_stats.csv
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # launches of one kernel and grid but different K (e.g. o_net 8192x512x512 and linear2 8192x512x4096 on 64x64 tiles) are
    # told apart by their duration: one row per power-of-two duration class
    agg[(r["Kernel_Name"][:110], str(grid), len(bin(max(d, 1))))].append(d)
rows = sorted(((sum(v), k, len(v)) for k, v in agg.items()), reverse=True)
with open(out + "/bench_cfg3_kernel_stats_by_grid.csv", "w") as g:
    g.write("kernel,grid_threads,calls,total_us,avg_us\n")
    for tot, (k, grid, _), n in rows[:70]:
        g.write('"%s",%s,%d,%.1f,%.2f\n' % (k, grid, n, tot / 1e3, tot / n / 1e3))
PY
# the same launch forced onto the 4-wave tiles 64x64, 128x64 and 128x128, in situ: traffic / time trade-off of the tile choice
for t in 11 21 22; do
  export BLM_GEMM_PLAN_SET="0,8192,512,4096,0,0,$t,1;0,8192,512,4096,1,0,$t,1"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_t$t -o bench -- $B --steps 4 --warmup 1 > $OUT/fetch_t$t.log 2>&1
  unset BLM_GEMM_PLAN_SET
  if [ $t = 11 ]; then K="void blm::gemm_f32_kernel<0, 1, 1, false, true, 0, 2>"; G=262144; elif [ $t = 21 ]; then K="void blm::gemm_f32_kernel<0, 2, 1, false, true, 0, 2>"; G=131072; else K="void blm::gemm_f32_kernel<0, 2, 2, false, true, 0, 2>"; G=65536; fi
  KERNEL="$K" GRID=$G python3 tools/pmc_traffic.py $(find $OUT/fetch_t$t -name "*counter_collection.csv" | head -1) $W $OUT/pmc_sampled_gemm_fwd_tile${t}_in_situ.json || true
done
# ---- the LSTM configurations (BASELINE configs[1] = "cfg2", configs[0] = "cfg1"): kernel stats of the training step and the
# counters of the fused step kernels (separate --pmc passes)
for w in cfg2 cfg1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o wl -- python3 tools/run_workload.py $w 12 > $OUT/trace_$w.log 2>&1
  cp $(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1) $OUT/bench_${w}_kernel_stats.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/lstm_mfma -o wl -- python3 tools/run_workload.py cfg2 4 > $OUT/lstm_mfma.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/lstm_wait -o wl -- python3 tools/run_workload.py cfg2 4 > $OUT/lstm_wait.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/lstm_fetch -o wl -- python3 tools/run_workload.py cfg2 4 > $OUT/lstm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/lstm_write -o wl -- python3 tools/run_workload.py cfg2 4 > $OUT/lstm_write.log 2>&1
{
  echo "# rocprofv3 --pmc (separate passes) over tools/run_workload.py cfg2: per-launch means of the fused LSTM step kernels"
  echo "# (B 64, H 1024: 256 workgroups x 256 threads = 65536 threads forward, 64 x 4 x 256 = 65536 backward)"
  for k in lstm_step_fwd_kernel lstm_step_bwd_kernel; do
    echo "== $k"
    for d in lstm_mfma lstm_wait lstm_fetch lstm_write; do
      python3 tools/pmc_kernel.py $(find $OUT/$d -name "*counter_collection.csv" | head -1) $k 65536
    done
  done
} > $OUT/pmc_lstm_step_kernels.txt 2>&1 || true
# ---- attention (T 128, 512 heads): counters of the forward and of the one-launch backward inside the headline step
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/wait -o bench -- $B --steps 4 --warmup 1 > $OUT/wait.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def means(d, sub):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    if not f:
        return {}
    for r in csv.DictReader(open(f[0])):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}
for tag, sub in (("fwd", "attn_fwd_mfma_kernel"), ("bwd", "attn_bwd_dkv_mfma_kernel")):
    c = {}
    for d in ("fetch", "write", "mfma", "wait"):
        c.update(means(d, sub))
    res = {"kernel": sub, "shape": "T 128, B 64, 8 heads of 64 (512 heads per launch), dropout 0.2, inside the cfg3 step", "counters_per_launch": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024  # MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64-byte halves
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        res["mfma_busy_frac_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / ((c["GRBM_GUI_ACTIVE"] / 8.0) * 256 * 4)  # GRBM_GUI_ACTIVE sums the 8 XCDs (tools/pmc_traffic.py)
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        res["wait_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    json.dump(res, open(out + "/pmc_attention_%s.json" % tag, "w"), indent=1)
PY
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +3M -delete
ls $OUT

#!/bin/bash
# HBM-traffic study of the roofline GEMM: per library build (current / first tile through registers /
# one-tile-per-trip loop) one rocprofv3 --pmc pass per counter group over tools/traffic_probe.py's launch sequence.
# Output: gpurun_out/traffic/<build>/<group>/...counter_collection.csv + labels.json; tools/traffic_report.py joins them.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
# builds: cur = the library as built; uneven = the same with BLM_GEMM_EVEN=0 (no even spreading of one-round grids);
# ft / rolled = variant builds (make EXTRA=-DBLM_GEMM_FIRST_TILE_REG ... / -DBLM_GEMM_ONE_TILE_TRIP ..., see csrc/Makefile)
declare -A LIBS=( [cur]="" [uneven]="" [ft]="$PWD/bayeslms_amd/libbayeslm_hip_ft.so" [rolled]="$PWD/bayeslms_amd/libbayeslm_hip_rolled.so" )
BUILDS=${BUILDS:-"cur uneven"}
declare -A CG=( [fetch]="FETCH_SIZE" [sizes]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
  [hit]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" [tcp]="TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum"
  [sqc]="SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_ICACHE_MISSES SQC_ICACHE_REQ" [dram]="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum"
  [write]="WRITE_SIZE" [xcd]="TCC_EA0_RDREQ" )
for b in $BUILDS; do
  for g in fetch sizes hit tcp sqc dram write xcd; do
    d=$OUT/$b/$g; mkdir -p $d
    if [ -n "${LIBS[$b]}" ]; then export BLM_LIB="${LIBS[$b]}"; else unset BLM_LIB; fi
    if [ $b = uneven ]; then export BLM_GEMM_EVEN=0; else unset BLM_GEMM_EVEN; fi
    PROBE_LABELS=$d/labels.json timeout -k 10 300 rocprofv3 --pmc ${CG[$g]} --kernel-trace --output-format csv -d $d -o probe -- python3 tools/traffic_probe.py > $d/log.txt 2>&1 || echo "pass $b/$g failed: see $d/log.txt"
    echo "$b/$g done: $(find $d -name '*counter_collection.csv' | wc -l) csv"
  done
done

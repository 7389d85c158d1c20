#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel trace + stats of the headline bench, and three
# separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/prof_$1/
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --gpus 1 --no-cpu-baseline --no-opt-in --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $B --steps 4 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $B --steps 4 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o bench -- $B --steps 4 --warmup 1 > $OUT/mfma.log 2>&1
find $OUT -name "*.csv" | head -20

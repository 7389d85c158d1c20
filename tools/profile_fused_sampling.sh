#!/bin/bash
# N1 evidence (run on the GPU box): VALU vs MFMA work of the Bayesian FFN linear2 forward GEMM with eps generated inside the
# B-tile loader (--fused-sampling 1) against the default (one materialisation pass + plain GEMM).  Output: gpurun_out/prof_n1/
set -e
OUT=$PWD/gpurun_out/prof_n1
mkdir -p $OUT
export TMPDIR=/tmp
for F in 0 1; do
  B="python3 bench.py --gpus 1 --no-cpu-baseline --no-opt-in --no-extra --fused-sampling $F --steps 4 --warmup 1"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $OUT/insts$F -o bench -- $B > $OUT/insts$F.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/busy$F -o bench -- $B > $OUT/busy$F.log 2>&1
done
ls $OUT/*/

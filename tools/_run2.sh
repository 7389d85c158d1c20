set -x
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r3_counters.txt 2>&1 || true
grep -c . gpurun_out/r3_counters.txt
timeout -k 10 600 python tools/gemm_tune.py > gpurun_out/r3_tune2.log 2>&1; echo rc=$? >> gpurun_out/r3_tune2.log; grep "^##\|keys," gpurun_out/r3_tune2.log
for b in 32 64; do B=$b python tools/lstm_step_bench.py 2>/dev/null; B=$b BLM_LSTM_WAVES=8 python tools/lstm_step_bench.py 2>/dev/null; done > gpurun_out/r3_lstm_b32.txt; cat gpurun_out/r3_lstm_b32.txt

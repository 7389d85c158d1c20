set -x
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or residual or linear" > gpurun_out/r3_t3.log 2>&1; echo rc=$? >> gpurun_out/r3_t3.log; tail -3 gpurun_out/r3_t3.log
BLM_GEMM_PLAN=legacy python bench.py --no-cpu-baseline --no-opt-in --no-extra > gpurun_out/r3_bench_legacy.json 2> gpurun_out/r3_bench_legacy.err; python -c "import json;d=json.load(open('gpurun_out/r3_bench_legacy.json'));print('legacy',d['value'],d['ms_per_step'],d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-opt-in --no-extra > gpurun_out/r3_bench_model.json 2> gpurun_out/r3_bench_model.err; python -c "import json;d=json.load(open('gpurun_out/r3_bench_model.json'));print('model',d['value'],d['ms_per_step'],d['roofline']['frac'])"
timeout -k 10 600 python tools/gemm_tune.py > gpurun_out/r3_tune1.log 2>&1; echo rc=$? >> gpurun_out/r3_tune1.log; grep "^##\|^==" gpurun_out/r3_tune1.log
timeout -k 10 900 python tools/gemm_tune.py --grid --min-frac 0.0 > gpurun_out/r3_grid1.log 2>&1; echo rc=$? >> gpurun_out/r3_grid1.log; tail -3 gpurun_out/r3_grid1.log

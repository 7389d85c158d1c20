set -x
timeout -k 10 900 python tools/gemm_tune.py --write-inc > gpurun_out/r3_tune3.log 2>&1; echo rc=$? >> gpurun_out/r3_tune3.log; grep "^##\|keys," gpurun_out/r3_tune3.log
make -C bayeslms_amd/csrc -j16 > gpurun_out/r3_make.log 2>&1; tail -2 gpurun_out/r3_make.log
for rep in 1 2; do
BLM_GEMM_PLAN=legacy python bench.py --no-cpu-baseline --no-opt-in --no-extra > gpurun_out/r3_bench_legacy.json 2> gpurun_out/r3_bench_legacy.err; python -c "import json;d=json.load(open('gpurun_out/r3_bench_legacy.json'));print('legacy',d['value'],d['ms_per_step'],d['roofline']['frac'])"
BLM_GEMM_PLAN=model python bench.py --no-cpu-baseline --no-opt-in --no-extra > gpurun_out/r3_bench_modelonly.json 2> gpurun_out/r3_bench_model.err; python -c "import json;d=json.load(open('gpurun_out/r3_bench_modelonly.json'));print('model-only',d['value'],d['ms_per_step'],d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-opt-in --no-extra > gpurun_out/r3_bench_new.json 2> gpurun_out/r3_bench_new.err; python -c "import json;d=json.load(open('gpurun_out/r3_bench_new.json'));print('table+model',d['value'],d['ms_per_step'],d['roofline']['frac'])"
done
for w in legacy model table; do
 if [ $w = table ]; then unset BLM_GEMM_PLAN; else export BLM_GEMM_PLAN=$w; fi
 python tools/step_breakdown.py lstm 2>/dev/null | tail -1
done

for rep in 1 2; do
BLM_GEMM_PLAN=model python bench.py --no-cpu-baseline --no-opt-in --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('model-only', d['value'], d['ms_per_step'], d['roofline']['frac'])"
python bench.py --no-cpu-baseline --no-opt-in --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('table(28)  ', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
timeout -k 10 1000 python tools/gemm_tune.py --grid --min-frac 0.0 > gpurun_out/r3_grid2.log 2>&1; tail -2 gpurun_out/r3_grid2.log

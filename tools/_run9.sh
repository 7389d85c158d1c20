for c in 0 8 20 25 34 50; do echo -n "chunk=$c "; BLM_LSTM_WAVE_CHUNK=$c python tools/run_workload.py recipe_lstm 20 2>/dev/null; done
for w in eval_lstm100 eval_lstm; do for on in 0 1; do echo -n "wavefront=$on "; BLM_LSTM_WAVEFRONT=$on python tools/run_workload.py $w 20 2>/dev/null; done; done
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stack2" 2>&1 | tail -2

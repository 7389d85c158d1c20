#!/bin/bash
# tile x split-K sweep of tools/gemm_bench.py (one process per combination); prints one line per shape
out=gpurun_out/sweep; mkdir -p $out; rm -f $out/*.txt
for t in 11 12 21 22; do for s in 0 2 4 8; do
  if [ $s = 0 ]; then BLM_GEMM_TILE=$t timeout -k 10 120 python tools/gemm_bench.py > $out/t${t}_s$s.txt 2>/dev/null || exit 1
  else BLM_GEMM_TILE=$t BLM_GEMM_SPLITK=$s timeout -k 10 120 python tools/gemm_bench.py > $out/t${t}_s$s.txt 2>/dev/null || exit 1; fi
done; done
timeout -k 10 120 python tools/gemm_bench.py > $out/auto.txt 2>/dev/null
python - <<'PY'
import glob, re, collections
res = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/sweep/*.txt"):
    tag = f.split("/")[-1][:-4]
    for line in open(f):
        m = re.match(r"(\S+\s+\S+)\s+M=.*?(\d+\.\d+) TFLOP/s", line)
        if m: res[m.group(1)][tag] = float(m.group(2))
for shape, d in res.items():
    best = sorted(((v, k) for k, v in d.items() if k != "auto"), reverse=True)[:3]
    print(f"{shape:18s} auto {d.get('auto', 0):6.1f} | best " + "  ".join(f"{k} {v:6.1f}" for v, k in best))
PY

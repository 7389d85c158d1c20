"""GPU: bench.py keeps the driver's contract -- ONE JSON line on stdout with the metric of BASELINE.json, the timed
steps, `roofline` (live HIP-event timing of the sampled GEMM) and, when asked, `cpu_baseline` / `extra_configs`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("BLM_GEMM_MODE", "f32") not in ("", "f32"),
                    reason="the contract line is the fp32 parity mode's; an opt-in GEMM mode relabels dtype / gemm_mode")
def test_bench_prints_one_json_line_with_the_contract_fields():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-opt-in", "--no-extra"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 2 and out["unit"] == "tokens/s"
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["config"]["global_batch"] == 64 and out["config"]["seq_len"] == 128 and out["config"]["gemm_mode"] == "f32"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - 3 * 128 * 64 / (out["ms_per_step"] * 3e-3)) < 0.01 * out["value"]
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and rf["launches"] == 3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.4 < rf["frac"] < 1.0  # north-star floor: 0.40
    assert abs(rf["achieved"] - 2.0 * 8192 * 512 * 4096 / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 0.5
    sr = out["step_roofline"]  # whole step: SURVEY 8(d) model FLOPs per token over the measured step time
    assert sr["model_flops_per_token"] == 294838272 and sr["peak"] == 157.3
    assert abs(sr["achieved"] - 294838272 * out["value"] / 1e12) < 0.01 * sr["achieved"] and abs(sr["frac"] - sr["achieved"] / 157.3) < 1e-3
    assert out["value"] > 2.0e5  # a silent fallback (vendor / eager path) or a broken kernel shows up here

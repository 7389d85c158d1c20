"""GPU: the opt-in split-bf16 GEMM arithmetic (`BLM_GEMM_MODE=bf16x6`: every fp32 operand value as three bf16 parts, six part
products, fp32 accumulate -- DESIGN.md section 7; never part of bench.py's `value` / `roofline`) held to the SAME bars as the
fp32 parity mode: every golden-fixture model test and the six reference `train.py` trajectories, in a child interpreter
that runs with the mode set from its first launch on (the claim used to be a round-2 one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("BLM_GEMM_MODE", "f32") not in ("", "f32"), reason="already running under an opt-in GEMM mode")
def test_fixture_model_tests_and_trajectories_pass_in_bf16x6_mode():
    env = dict(os.environ)
    env["BLM_GEMM_MODE"] = "bf16x6"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_gpu_models.py"), os.path.join(ROOT, "tests", "test_gpu_train_traj.py")],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail.split("\n")[-2], tail

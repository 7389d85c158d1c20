"""CPU: the host side of libbayeslm_hip.so under sanitizers (GPU AddressSanitizer is not available on the pool).
`make -C bayeslms_amd/csrc asan tsan` compiles every translation unit for the host only (launchers, argument checks, the GEMM
planner, the option registry, the error state; kernels become launch stubs without a code object) with AddressSanitizer +
UndefinedBehaviorSanitizer / ThreadSanitizer.  A child interpreter loads that build (BLM_LIB) under LD_PRELOAD of the
sanitizer runtime; any finding aborts the child or shows in its output.  Nothing here goes to a GPU."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayeslms_amd", "csrc")


def _runtime(kind):
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.%s-x86_64.so" % kind)
    if not hits:
        pytest.skip("no %s runtime in this image" % kind)
    return sorted(hits)[-1]


@pytest.fixture(scope="module")
def san_libs():
    r = subprocess.run(["make", "-C", CSRC, "-j4", "asan", "tsan"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    libs = {k: os.path.join(ROOT, "bayeslms_amd", "libbayeslm_hip_%s.so" % k) for k in ("asan", "tsan")}
    assert all(os.path.exists(p) for p in libs.values())
    return libs


def _child(kind, lib, argv, timeout=900):
    env = dict(os.environ)
    env.update({"LD_PRELOAD": _runtime(kind), "BLM_LIB": lib, "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", ""),
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1",
                "TSAN_OPTIONS": "halt_on_error=0 report_signal_unsafe=0 exitcode=66"})
    r = subprocess.run([sys.executable] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    for mark in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "UndefinedBehaviorSanitizer"):
        assert mark not in out, out[-6000:]
    return r, out


def test_planner_and_c_abi_tests_pass_on_the_sanitized_host_build(san_libs):
    """tests/test_gemm_plan_cpu.py + tests/test_cabi_cpu.py with the ASan + UBSan build behind every call."""
    r, out = _child("asan", san_libs["asan"], ["-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                                               os.path.join(ROOT, "tests", "test_gemm_plan_cpu.py"),
                                               os.path.join(ROOT, "tests", "test_cabi_cpu.py")])
    assert r.returncode == 0 and " passed" in out, out[-4000:]


def test_no_entry_point_crashes_or_overflows_on_bad_arguments(san_libs):
    """Every blm_* entry point with NULL / misaligned / negative / zero / large / huge arguments (tests/sanitizer_sweep.py,
    driven by the ctypes prototypes): a status comes back, with a message when it is not BLM_OK -- never a crash, a size
    expression that overflows, or a dereference of a pointer the host has no business reading."""
    r, out = _child("asan", san_libs["asan"], [os.path.join(ROOT, "tests", "sanitizer_sweep.py"), "sweep"])
    assert r.returncode == 0 and "SWEEP_OK" in out, out[-4000:]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_planner_state_survives_four_threads(san_libs, kind):
    """choose_plan's memo, the run-time plan tables, the override, the CU count, the comm window and the option registry
    under three planning threads and one mutating thread (ctypes releases the interpreter lock around every call)."""
    r, out = _child(kind, san_libs[kind], [os.path.join(ROOT, "tests", "sanitizer_sweep.py"), "hammer"])
    assert r.returncode == 0 and "HAMMER_OK" in out, out[-4000:]

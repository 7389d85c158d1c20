import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_c_host(out_dir):
    """examples/c_host/bayes_linear_step.c compiled by gcc against include/bayeslm.h, the in-tree library and the HIP runtime:
    a host with no Python and no torch above the C ABI.  -> path of the executable."""
    import subprocess
    exe = os.path.join(str(out_dir), "bayes_linear_step")
    libdir = os.path.join(ROOT, "bayeslms_amd")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host", "bayes_linear_step.c"),
                           "-L" + libdir, "-lbayeslm_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def load_golden(name, device="cpu"):
    """-> (dict of plain entries as torch tensors, state_dict, grad dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    plain, sd, grad = {}, {}, {}
    for k in z.files:
        v = z[k]
        t = torch.from_numpy(v).to(device) if v.dtype.kind in "fiu" else v
        if k.startswith("sd/"):
            sd[k[3:]] = t
        elif k.startswith("grad/"):
            grad[k[5:]] = t
        elif k.startswith("sd2/"):  # second model of an interpolation fixture
            plain.setdefault("sd2", {})[k[4:]] = t
        else:
            plain[k] = t
    return plain, sd, grad


@pytest.fixture
def golden():
    return load_golden

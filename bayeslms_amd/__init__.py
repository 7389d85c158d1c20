"""bayeslms_amd: MI355X-native (gfx950) engine for the Bayesian LM hot path of AmourWaltz/BayesLMs.

Python host on PyTorch-ROCm (device memory, autograd tape, torch.distributed) over a thin C ABI
(include/bayeslm.h, libbayeslm_hip.so) of hand-written HIP kernels.  No CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import BayesLMError  # noqa: F401

__version__ = "0.1.0"

"""Top-level ``mixlib`` module name, so that the reference's callers (``import mixlib`` in
MixQ/src/mixquant/modules/linear.py:5 and modelopt/torch/export/model_config_utils.py:433) run unchanged against the
MI355X library: every op is re-exported from ``mixq_tensorrt_llm_amd.mixlib`` (ctypes over libmixq_mi355x.so, no
fallback).  Put the repository root on ``sys.path`` (or install the package) and the name resolves."""
from mixq_tensorrt_llm_amd.mixlib import *  # noqa: F401,F403
from mixq_tensorrt_llm_amd.mixlib import __all__  # noqa: F401

"""mixq_tensorrt_llm_amd -- MI355X-native MixQ W8A8O16 linear operator (hand-written HIP for gfx950).

Only what the hot path needs:
  csrc/        HIP kernels + the C ABI (include/mixq.h) -> libmixq_mi355x.so
  _lib         ctypes binding of that library (no fallback: missing library == loud failure)
  plugin       mirror of the reference's plugin.py (MixQLinear, mixgemm, plugin object)
  mixlib       mirror of the reference's mixlib torch-extension op names
  pack         quantize-time producer of the 7 tensors (pack_linear_weights counterpart)
  parallel     row-sharded W + one all-gather of the fp16 output (RCCL) when TP > 1
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"

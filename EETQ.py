"""Top-level ``EETQ`` module name for the three functions the reference's MixQ path imports from the EETQ extension
(``from EETQ import quant_weights, preprocess_weights, w8_a16_gemm``: MixQ/src/mixquant/modules/linear.py:9,
modelopt/torch/export/model_config_utils.py:434; bound in EETQ/csrc/eetpy.cpp:7-17), served by libmixq_mi355x.so.
Nothing else of EETQ is reproduced (rotary embedding, layernorm: not on the MixQ linear path)."""
import torch

from mixq_tensorrt_llm_amd import mixlib as _mixlib
from mixq_tensorrt_llm_amd import pack as _pack

__all__ = ["quant_weights", "preprocess_weights", "w8_a16_gemm"]


def quant_weights(origin_weight, quant_type=torch.int8, return_unprocessed_quantized_tensor=False):
    """symmetric_quantize_last_axis_of_tensor (cutlass_preprocessors.cc:573-660) on a CPU tensor [K, N]: per-column scale
    max|col| / 128, round half away from zero, then the mixed-GEMM interleave.  Returns [processed int8 [K, N] (the
    interleaved, +128-biased bytes in an int8 tensor, as the reference returns them), scales fp16 [N]] and, on request,
    the un-interleaved int8 matrix in front."""
    assert quant_type == torch.int8, "the MixQ path quantises to int8 (linear.py:103, model_config_utils.py:438)"
    processed, scales, unprocessed = _pack.eetq_quant_weights(origin_weight)
    out = [processed.view(torch.int8), scales]
    return [unprocessed] + out if return_unprocessed_quantized_tensor else out


def preprocess_weights(origin_weight, is_int4=False):
    """preprocess_weights_cuda (cutlass_preprocessors.cc:536-545): row-major int8 [K, N] -> interleaved image."""
    assert not is_int4, "int4 weights are not on the MixQ int8_mix path"
    return _mixlib.preprocess_weights(origin_weight).view(torch.int8)


def w8_a16_gemm(input, weight, scale):
    """w8_a16_gemm_forward_cuda (fpA_intB_gemm_wrapper.cu:29-70): fp16 [m, k] x interleaved int8 [k, n] -> fp16 [m, n]."""
    return _mixlib.w8_a16_gemm(input, weight, scale)

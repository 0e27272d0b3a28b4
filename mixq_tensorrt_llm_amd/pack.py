"""Quantize-time producer of the MixQ linear's tensors (the caller on the input side of the operator).

Counterpart of ``pack_linear_weights`` in the reference (modelopt/torch/export/model_config_utils.py:378-472) for ONE
linear layer: given the original fp16 weight ``W [N,K]`` and the per-input-channel activation scales, emit the seven
tensors of SURVEY.md A.1 in their true dtypes.  Host-side, offline; uses torch CPU ops like the reference does.
The EETQ interleave of ``qweight`` is done by the C++ importer in libmixq_mi355x.so (host code, no GPU needed).
"""
import ctypes

import numpy as np
import torch

from . import _lib

NUM_OUTLIERS = 128  # model_config_utils.py:444 fp_features


def select_outlier_columns(act_scales: torch.Tensor, num: int = NUM_OUTLIERS, stable: bool = False) -> torch.Tensor:
    """model_config_utils.py:446-448: indices of the ``num`` largest activation scales, ascending by scale.

    The reference's very call -- ``torch.sort(layer_scales)[1][-128:]`` on a CPU tensor, NOT stable -- so that ``fp_ind``
    (and with it the column order of ``fp_weight``) comes out as the reference stores it: real tables tie inside the top
    128 (act_scales/Llama-2-1b.pt: 2..24 tied values per vector) and a stable sort orders those groups differently
    (tests/golden/model_walk.npz pins the order).  The SET of columns is the same either way unless a tie straddles the
    128th place; ``stable=True`` gives a build-independent order for callers that do not need the reference's."""
    s = act_scales.float().cpu()
    return torch.sort(s, stable=stable)[1][-num:].to(torch.int32)


def weight_scales(W: torch.Tensor) -> torch.Tensor:
    """model_config_utils.py:429-430: fp16(max_k |W[n,k]| / 127), computed BEFORE the outlier columns are zeroed."""
    W = W.cpu()
    return (W.abs().max(dim=1)[0].unsqueeze(1) / 127).to(torch.float16).reshape(W.shape[0])


def quantize_weight(W: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """model_config_utils.py:298-308 (int8_mix): round-half-even(W / s) clamped to [-128, 127]."""
    return (W.cpu() / scales.cpu()[:, None]).round().clamp(-128, 127).to(torch.int8)


def eetq_quant_weights(Wt: torch.Tensor):
    """EETQ ``quant_weights(Wt [K,N], int8)`` = symmetric_quantize (cutlass_preprocessors.cc:573-660):
    per column scale = max|col| / 128 (kept in fp32 for the division, returned as fp16), round half away from zero.
    Returns (interleaved uint8 [K,N], scales fp16 [N], un-interleaved int8 [K,N])."""
    Wt = Wt.cpu().float()
    col_max = Wt.abs().max(dim=0)[0] * (1.0 / 128.0)
    q = Wt / col_max[None, :]
    q = torch.where(q >= 0, torch.floor(q + 0.5), torch.ceil(q - 0.5))  # C round(): half away from zero
    q = torch.nan_to_num(q, nan=0.0).clamp(-128, 127).to(torch.int8).contiguous()
    K, N = q.shape
    out = torch.empty((K, N), dtype=torch.uint8)
    _lib.check(_lib.load().mixq_preprocess_weights_int8(ctypes.c_void_p(out.data_ptr()),
                                                        ctypes.c_void_p(q.data_ptr()), K, N), "preprocess_weights")
    return out, col_max.to(torch.float16), q


def pack_linear_weights(W: torch.Tensor, act_scales: torch.Tensor, num_outliers: int = NUM_OUTLIERS,
                        outlier_weights: str = "fp16") -> dict:
    """One layer of model_config_utils.py:421-466.  Returns numpy arrays keyed like the checkpoint tensors.

    outlier_weights = "fp16": the reference's T-flavour (fp_weight = original fp16 columns, :452).
    outlier_weights = "int8": the P-flavour / fpA_intB mode of BASELINE config 4 -- fp_weight = dequantised int8 columns
    ``q_weight[:, ind] * scale_col`` (MixQ/src/mixquant/modules/linear.py:204), so the outlier path carries exactly the
    information the int8 weights have."""
    assert W.dtype == torch.float16 and W.dim() == 2
    assert outlier_weights in ("fp16", "int8")
    W = W.cpu().clone()
    sW = weight_scales(W)                                     # :429-430
    qweight, eetq_scales, _ = eetq_quant_weights(W.t().contiguous())   # :437-441 (un-zeroed W^T)
    ind = select_outlier_columns(act_scales, num_outliers)    # :446-448
    if outlier_weights == "int8":
        q_cols = quantize_weight(W[:, ind.long()], sW)        # int8 of the un-zeroed outlier columns
        fp_weight = q_cols.to(torch.float16) * sW[:, None]    # fp16 product, one rounding
    else:
        fp_weight = W[:, ind.long()].clone()                  # :452
    W[:, ind.long()] *= 0                                     # :453
    Wq = quantize_weight(W, sW)                               # :460-464
    return dict(weight=Wq.numpy(), weights_scaling_factor=sW.numpy(), fp_weight=fp_weight.numpy(),
                fp_ind=ind.numpy().astype(np.int32), qweight=qweight.numpy(), scales=eetq_scales.numpy())

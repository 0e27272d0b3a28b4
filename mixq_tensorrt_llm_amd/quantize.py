"""Model-level quantize-time walk: an HF-named state dict + an activation-scale table -> the MixQ checkpoint.

Counterpart of the reference's export step for the T-flavour (the caller on the INPUT side of the operator):

    merge_qkv(model_config)            modelopt/torch/export/model_config_utils.py:203-217   q | k | v concatenated on N
    pack_linear_weights(model_config)  modelopt/torch/export/model_config_utils.py:378-472   per decoder layer, three linears
    save_checkpoint                    tensorrt_llm/models/modeling_utils.py:510-520          rank{r}.safetensors + config.json

Which linears, and from which HF modules (layer_utils.py:753-786 keyword tables, model_config_utils.py:409-415):

    TRT-LLM module   HF source (weights)                     act-scale key the REFERENCE reads (:398-401, :423-425)
    attention.qkv    self_attn.{q,k,v}_proj concatenated     model.layers.{i}.self_attn.q_proj
    mlp.gate         mlp.up_proj                             model.layers.{i}.mlp.gate_proj   (same input tensor as up_proj)
    mlp.proj         mlp.down_proj   (K = intermediate)      model.layers.{i}.mlp.up_proj     (length = hidden: SURVEY A.3 quirk #4)

``attention.dense`` (o_proj), ``mlp.fc`` (HF gate_proj) and ``lm_head`` stay fp16 in the reference and are not part of this path.

Quirk #4 is reproduced by default (``fix_quirk4=False``): the K = intermediate down projection selects its 128 outlier columns
from a hidden-size vector, so every index is < hidden.  ``fix_quirk4=True`` reads ``mlp.down_proj``'s own vector instead.
Quirk #5 (the hard-coded ``act_scales/Qwen2-72B.pt``, :391) is not reproduced: the table is an argument.

Host-side and offline (torch CPU + the C++ interleave importer of libmixq_mi355x.so); the per-layer arithmetic is
``pack.pack_linear_weights``.  Pinned by ``tests/golden/model_walk.npz`` (the reference's own ``merge_qkv`` +
``pack_linear_weights`` executed on the same synthetic layers with the real ``act_scales/Llama-2-1b.pt`` keys).
"""
from typing import Dict, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import checkpoint, pack

# model_config_utils.py:398-401 -- names[i] pairs with linear_layers[i] = [attention.qkv, mlp.gate, mlp.proj] (:409-415)
REFERENCE_ACT_SCALE_NAMES = {"attention.qkv": "self_attn.q_proj", "mlp.gate": "mlp.gate_proj", "mlp.proj": "mlp.up_proj"}
# the vector that belongs to each linear's real input (what fix_quirk4 switches mlp.proj to)
OWN_ACT_SCALE_NAMES = {"attention.qkv": "self_attn.q_proj", "mlp.gate": "mlp.up_proj", "mlp.proj": "mlp.down_proj"}
HF_SOURCES = {
    "attention.qkv": ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"),   # QKVConfig.weight, model_config.py:132-138
    "mlp.gate": ("mlp.up_proj",),                                                     # layer_utils.py:779-786 gate_keywords
    "mlp.proj": ("mlp.down_proj",),                                                   # layer_utils.py:766-777 proj_keywords
}


def act_scale_key(layer: int, which: str, fix_quirk4: bool = False) -> str:
    """The key of the activation-scale table the walk reads for (layer, TRT-LLM module) -- model_config_utils.py:423-425."""
    names = dict(REFERENCE_ACT_SCALE_NAMES)
    if fix_quirk4:
        names["mlp.proj"] = OWN_ACT_SCALE_NAMES["mlp.proj"]
    return f"model.layers.{layer}.{names[which]}"


def hf_linear(state_dict: Mapping[str, torch.Tensor], layer: int, which: str,
              prefix: str = "model.layers") -> Tuple[torch.Tensor, Optional[torch.Tensor], Tuple[int, ...]]:
    """(W fp16 [N, K], bias or None, the N of every concatenated part) of one TRT-LLM module from HF-named tensors."""
    ws, bs, parts = [], [], []
    for src in HF_SOURCES[which]:
        w = state_dict[f"{prefix}.{layer}.{src}.weight"]
        assert w.dim() == 2, f"{src}.weight: expected [N, K], got {tuple(w.shape)}"
        ws.append(w.to(torch.float16))
        parts.append(int(w.shape[0]))
        bs.append(state_dict.get(f"{prefix}.{layer}.{src}.bias"))
    assert all(b is None for b in bs) or all(b is not None for b in bs), "K and V should have a bias when Q has one"  # model_config.py:146-150
    W = torch.cat(ws, dim=0) if len(ws) > 1 else ws[0]
    bias = torch.cat([b.to(torch.float16) for b in bs]) if bs[0] is not None else None
    return W, bias, tuple(parts)


def rank_major_rows(parts: Sequence[int], tp_size: int) -> np.ndarray:
    """Row order that puts rank r's slice of EVERY part side by side: [q_0 | k_0 | v_0 | q_1 | k_1 | v_1 | ...].

    The reference splits q, k and v per rank BEFORE it merges them (postprocess_model_config runs ahead of merge_qkv,
    model_config_export.py:356-380), so rank r's fused qkv rows are its own heads' q, k and v.  Packing is row-wise, so
    permuting the rows first and then taking contiguous shards (parallel.shard_packed) gives exactly those tensors."""
    order, base = [], 0
    starts = []
    for n in parts:
        assert n % tp_size == 0, f"part of {n} rows does not split {tp_size} ways"
        starts.append(base)
        base += n
    for r in range(tp_size):
        for s, n in zip(starts, parts):
            per = n // tp_size
            order.append(np.arange(s + r * per, s + (r + 1) * per))
    return np.concatenate(order)


def quantize_layer(state_dict: Mapping[str, torch.Tensor], act_scales: Mapping[str, torch.Tensor], layer: int, which: str, *,
                   tp_size: int = 1, fix_quirk4: bool = False, qkv_layout: str = "contiguous", outlier_weights: str = "fp16",
                   prefix: str = "model.layers") -> Dict[str, np.ndarray]:
    """The seven tensors (+ bias) of one MixQ linear, full N; ``qkv_layout``: see quantize_model."""
    W, bias, parts = hf_linear(state_dict, layer, which, prefix)
    if tp_size > 1 and qkv_layout == "per_rank_heads" and len(parts) > 1:
        order = torch.from_numpy(rank_major_rows(parts, tp_size))
        W = W[order]
        bias = bias[order] if bias is not None else None
    scales = act_scales[act_scale_key(layer, which, fix_quirk4)]
    packed = pack.pack_linear_weights(W, torch.as_tensor(scales), outlier_weights=outlier_weights)
    if bias is not None:
        packed["bias"] = bias.cpu().numpy()
    return packed


def quantize_model(state_dict: Mapping[str, torch.Tensor], act_scales: Union[str, Mapping[str, torch.Tensor]], num_layers: int, *,
                   tp_size: int = 1, out_dir: Optional[str] = None, fix_quirk4: bool = False, qkv_layout: str = "contiguous",
                   outlier_weights: str = "fp16", config: Optional[dict] = None, prefix: str = "model.layers",
                   ) -> Dict[str, Dict[str, np.ndarray]]:
    """Walk the decoder layers like pack_linear_weights (model_config_utils.py:405-466) and, with ``out_dir``, write
    ``config.json`` + ``rank{r}.safetensors`` (checkpoint.save_checkpoint: fp16 carriers under the MixQLinear parameter names).

    state_dict   HF names: ``{prefix}.{i}.self_attn.{q,k,v}_proj.weight`` (+ ``.bias``: Qwen2), ``mlp.up_proj.weight``,
                 ``mlp.down_proj.weight``; fp16 / bf16 / fp32 tensors, [out_features, in_features].
    act_scales   the table (``model.layers.{i}.{name}`` -> fp32 [in_features]) or the path of its ``.pt`` file
                 (the reference loads act_scales/<model>.pt, :391-393).
    tp_size      rows of every W are sharded ``tp_size`` ways at save time (north_star: one all-gather of the fp16 output).
    qkv_layout   "contiguous": rank r owns rows [r N/tp, (r+1) N/tp) of q | k | v, the all-gather returns [q | k | v];
                 "per_rank_heads": rank r owns [q_r | k_r | v_r] -- the reference's own split (rank_major_rows), for a runtime
                 whose attention consumes the local shard without a gather.
    Returns {module prefix: packed tensors in true dtypes, full N} (what save_checkpoint consumes)."""
    assert qkv_layout in ("contiguous", "per_rank_heads"), qkv_layout
    if isinstance(act_scales, str):
        act_scales = torch.load(act_scales, map_location="cpu")
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for i in range(num_layers):
        for which in checkpoint.LAYER_PREFIXES:   # [attention.qkv, mlp.gate, mlp.proj], the reference's order (:409-415)
            layers[checkpoint.layer_prefix(i, which)] = quantize_layer(
                state_dict, act_scales, i, which, tp_size=tp_size, fix_quirk4=fix_quirk4, qkv_layout=qkv_layout,
                outlier_weights=outlier_weights, prefix=prefix)
    if out_dir is not None:
        cfg = dict(config or {})
        cfg.setdefault("num_hidden_layers", num_layers)
        cfg.setdefault("quantization", {}).update({"mixq_fix_quirk4": bool(fix_quirk4), "mixq_qkv_layout": qkv_layout})
        checkpoint.save_checkpoint(out_dir, layers, cfg, tp_size=tp_size)
    return layers

"""TensorRT-LLM-style checkpoint I/O for MixQ linears (the on-disk format either side of the op).

The reference writes ``config.json`` + ``rank{r}.safetensors`` (tensorrt_llm/models/modeling_utils.py:510-520); the
MixQ tensors of a layer are stored under the module path of the ``MixQLinear`` that consumes them
(``transformer.layers.{i}.attention.qkv`` / ``.mlp.gate`` / ``.mlp.proj``, tensorrt_llm/quantization/quantize.py:307-350)
with the parameter names of plugin.py:99-123 and -- like every TensorRT plugin input -- as **fp16 carriers**:

    <prefix>.weight                  fp16 [N, K/2]   = int8  [N, K]      (mixlib.int8_matrix_to_half, model_config_utils.py:466)
    <prefix>.weights_scaling_factor  fp16 [N]
    <prefix>.fp_weight               fp16 [N, 128]
    <prefix>.fp_ind                  fp16 [256]      = int32 [128]       (mixlib.int_to_half, :457-458)
    <prefix>.qweight                 fp16 [K, N/2]   = uint8 [K, N] interleaved (:437-440)
    <prefix>.bias                    optional

Host-side only (safetensors + numpy); no GPU needed.  Row-sharded (TP) checkpoints hold rank r's slice of every
per-output-feature tensor (parallel.shard_packed).
"""
import json
import os
from typing import Dict, Iterable, Optional

import numpy as np
import torch
from safetensors.torch import load_file, save_file

from . import parallel

MIXQ_TENSORS = ("weight", "weights_scaling_factor", "fp_weight", "fp_ind", "qweight")
LAYER_PREFIXES = ("attention.qkv", "mlp.gate", "mlp.proj")  # the three linears the reference converts


def layer_prefix(layer: int, which: str) -> str:
    assert which in LAYER_PREFIXES, which
    return f"transformer.layers.{layer}.{which}"


def to_carriers(packed: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """True-dtype tensors of pack.pack_linear_weights -> the fp16 carrier tensors stored on disk."""
    N, K = packed["weight"].shape

    def view16(a, shape):
        return torch.from_numpy(np.ascontiguousarray(a)).view(torch.float16).reshape(shape).clone()

    out = {
        "weight": view16(packed["weight"].astype(np.int8), (N, K // 2)),
        "weights_scaling_factor": view16(packed["weights_scaling_factor"].astype(np.float16), (N,)),
        "fp_weight": view16(packed["fp_weight"].astype(np.float16), (N, packed["fp_weight"].shape[1])),
        "fp_ind": view16(packed["fp_ind"].astype(np.int32), (2 * packed["fp_ind"].size,)),
        "qweight": view16(packed["qweight"].astype(np.uint8), (K, N // 2)),
    }
    if packed.get("bias") is not None:
        out["bias"] = torch.from_numpy(np.ascontiguousarray(packed["bias"]))
    return out


def from_carriers(t: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
    """Inverse of to_carriers (bit reinterpretation only)."""
    N = t["weights_scaling_factor"].numel()
    K = t["weight"].shape[1] * 2
    out = {
        "weight": t["weight"].contiguous().view(torch.int8).reshape(N, K).numpy(),
        "weights_scaling_factor": t["weights_scaling_factor"].contiguous().numpy(),
        "fp_weight": t["fp_weight"].contiguous().numpy(),
        "fp_ind": t["fp_ind"].contiguous().view(torch.int32).numpy(),
        "qweight": t["qweight"].contiguous().view(torch.uint8).reshape(K, N).numpy(),
    }
    if "bias" in t:
        out["bias"] = t["bias"].numpy()
    return out


def save_checkpoint(out_dir: str, layers: Dict[str, Dict[str, np.ndarray]], config: Optional[dict] = None,
                    tp_size: int = 1) -> None:
    """``layers``: module prefix -> packed tensors (true dtypes, full N).  Writes config.json and one
    rank{r}.safetensors per TP rank (rows of W sharded)."""
    os.makedirs(out_dir, exist_ok=True)
    cfg = dict(config or {})
    cfg.setdefault("quantization", {})
    cfg["quantization"].update({"quant_algo": "int8_mix", "num_outlier_columns": 128})   # QuantAlgo.int8_mix
    cfg.setdefault("mapping", {}).update({"world_size": tp_size, "tp_size": tp_size, "pp_size": 1})
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    for r in range(tp_size):
        tensors = {}
        for prefix, packed in layers.items():
            shard = parallel.shard_packed(packed, tp_size, r) if tp_size > 1 else packed
            for name, ten in to_carriers(shard).items():
                tensors[f"{prefix}.{name}"] = ten
        save_file(tensors, os.path.join(out_dir, f"rank{r}.safetensors"))


def load_checkpoint(ckpt_dir: str, rank: int = 0) -> (dict, Dict[str, Dict[str, torch.Tensor]]):
    """Returns (config, {module prefix: {param name: fp16 carrier tensor}}) for one rank."""
    with open(os.path.join(ckpt_dir, "config.json")) as f:
        cfg = json.load(f)
    flat = load_file(os.path.join(ckpt_dir, f"rank{rank}.safetensors"))
    layers: Dict[str, Dict[str, torch.Tensor]] = {}
    for key, ten in flat.items():
        prefix, name = key.rsplit(".", 1)
        layers.setdefault(prefix, {})[name] = ten
    return cfg, layers


def load_linear(layer, tensors: Dict[str, torch.Tensor]):
    """Install one module's carrier tensors into a plugin.MixQLinear (names match plugin.py:99-123)."""
    dev = layer.weight.device
    had_image = getattr(layer, "weight_image", None) is not None
    if had_image:   # (a registered streaming copy belongs to the tensor being replaced; rebuilt for the new one below, as MixQLinear.load does)
        layer.prepare_decode_batches(False)
    for name in MIXQ_TENSORS:
        want = tuple(getattr(layer, name).shape)
        got = tuple(tensors[name].shape)
        assert want == got, f"{name}: checkpoint {got} vs module {want}"
        setattr(layer, name, tensors[name].to(dev))
    if layer.bias is not None and "bias" in tensors:
        want, got = tuple(layer.bias.shape), tuple(tensors["bias"].shape)
        assert want == got, f"bias: checkpoint {got} vs module {want} (row-sharded layers hold their [N/tp] slice)"
        layer.bias = tensors["bias"].to(dev)
    if had_image:
        layer.prepare_decode_batches(True)
    return layer


def iter_mixq_prefixes(num_layers: int) -> Iterable[str]:
    for i in range(num_layers):
        for which in LAYER_PREFIXES:
            yield layer_prefix(i, which)

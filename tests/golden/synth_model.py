"""Closed-form synthetic decoder layers shared by gen_golden.py (which feeds them to the REFERENCE's merge_qkv +
pack_linear_weights) and tests/test_quantize_walk.py (which feeds them to mixq_tensorrt_llm_amd.quantize).  Integer
arithmetic only, so both sides see bit-identical fp16 weights without storing them: hidden = 4096 and intermediate = 11008
match the real vectors of act_scales/Llama-2-1b.pt; the output widths are small to keep the fixture and the test fast."""
import numpy as np
import torch

HIDDEN, INTER, LAYERS = 4096, 11008, 2
# HF module -> (out_features, in_features); q / k / v of different widths (grouped-query attention shape)
HF_SHAPES = {
    "self_attn.q_proj": (64, HIDDEN), "self_attn.k_proj": (32, HIDDEN), "self_attn.v_proj": (32, HIDDEN),
    "mlp.gate_proj": (64, HIDDEN), "mlp.up_proj": (64, HIDDEN), "mlp.down_proj": (32, INTER),
}


def synth_weight(n: int, k: int, salt: int) -> torch.Tensor:
    """fp16 [n, k]: ((n * 7919 + k * 104729 + salt * 1299709) mod 2003 - 1001) / 1001 * 0.05, a few planted large entries."""
    r = np.arange(n, dtype=np.int64)[:, None]
    c = np.arange(k, dtype=np.int64)[None, :]
    v = ((r * 7919 + c * 104729 + salt * 1299709) % 2003 - 1001).astype(np.float64) / 1001.0 * 0.05
    v[(r * 31 + c * 17 + salt) % 4099 == 0] *= 6.0   # row maxima that are not on the regular grid
    return torch.from_numpy(v.astype(np.float16))


def synth_bias(n: int, salt: int) -> torch.Tensor:
    r = np.arange(n, dtype=np.int64)
    return torch.from_numpy((((r * 613 + salt * 7) % 101 - 50) / 400.0).astype(np.float16))


def state_dict(with_bias: bool = False):
    sd = {}
    for layer in range(LAYERS):
        for j, (name, (n, k)) in enumerate(HF_SHAPES.items()):
            sd[f"model.layers.{layer}.{name}.weight"] = synth_weight(n, k, 10 * layer + j)
            if with_bias and name.startswith("self_attn."):
                sd[f"model.layers.{layer}.{name}.bias"] = synth_bias(n, 10 * layer + j)
    return sd

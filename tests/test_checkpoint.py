"""CPU: TensorRT-LLM-style checkpoint writer / reader for MixQ linears."""
import json
import os

import numpy as np
import torch

from conftest import make_layer

from mixq_tensorrt_llm_amd import checkpoint, pack, parallel


def _layers():
    out = {}
    for i, (n, k) in enumerate([(128, 256), (256, 128)]):
        _, W, act = make_layer(4, n, k, seed=i)
        out[checkpoint.layer_prefix(0, checkpoint.LAYER_PREFIXES[i])] = pack.pack_linear_weights(
            torch.from_numpy(W), torch.from_numpy(act))
    return out


def test_roundtrip_and_carrier_contract(tmp_path):
    layers = _layers()
    checkpoint.save_checkpoint(str(tmp_path), layers, {"architecture": "LlamaForCausalLM"})
    cfg, loaded = checkpoint.load_checkpoint(str(tmp_path))
    assert cfg["quantization"]["quant_algo"] == "int8_mix" and cfg["mapping"]["tp_size"] == 1
    assert json.load(open(os.path.join(tmp_path, "config.json")))["architecture"] == "LlamaForCausalLM"
    for prefix, packed in layers.items():
        t = loaded[prefix]
        N, K = packed["weight"].shape
        # declared (carrier) shapes and dtypes of plugin.py:99-123
        assert all(t[n].dtype == torch.float16 for n in checkpoint.MIXQ_TENSORS)
        assert tuple(t["weight"].shape) == (N, K // 2) and tuple(t["qweight"].shape) == (K, N // 2)
        assert tuple(t["fp_ind"].shape) == (256,) and tuple(t["fp_weight"].shape) == (N, 128)
        back = checkpoint.from_carriers(t)
        for name in ("weight", "fp_ind", "qweight"):
            assert np.array_equal(back[name], packed[name]), name
        for name in ("weights_scaling_factor", "fp_weight"):
            assert np.array_equal(back[name].view(np.uint16), packed[name].view(np.uint16)), name


def test_tp_checkpoint_shards_rows(tmp_path):
    layers = _layers()
    checkpoint.save_checkpoint(str(tmp_path), layers, tp_size=2)
    for r in range(2):
        _, loaded = checkpoint.load_checkpoint(str(tmp_path), rank=r)
        for prefix, packed in layers.items():
            want = parallel.shard_packed(packed, 2, r)
            got = checkpoint.from_carriers(loaded[prefix])
            for name in ("weight", "qweight", "fp_ind"):
                assert np.array_equal(got[name], want[name]), (prefix, name, r)
            assert np.array_equal(got["fp_weight"].view(np.uint16), want["fp_weight"].view(np.uint16))


def test_prefix_names_follow_the_reference_module_tree():
    names = list(checkpoint.iter_mixq_prefixes(2))
    assert names[0] == "transformer.layers.0.attention.qkv" and names[-1] == "transformer.layers.1.mlp.proj"
    assert len(names) == 6

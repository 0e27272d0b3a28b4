"""Model-level packing walk: mixq_tensorrt_llm_amd.quantize against tests/golden/model_walk.npz -- the
REFERENCE's own merge_qkv + pack_linear_weights (modelopt/torch/export/model_config_utils.py:203-217, 378-472) executed by
tests/golden/gen_golden.py on the same closed-form layers with the real act_scales/Llama-2-1b.pt vectors."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import assert_same_outlier_columns
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import synth_model as sm  # noqa: E402

from mixq_tensorrt_llm_amd import checkpoint, parallel, quantize  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "model_walk.npz"))
WHICH = ("attention.qkv", "mlp.gate", "mlp.proj")


def act_table():
    return {k[4:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("act.")}


def test_key_map_is_the_references():
    """The keys the reference's walk READ, in order (recorded by the fixture generator), are the keys this walk asks for."""
    want = [str(k) for k in GOLD["keys_read"]]
    got = [quantize.act_scale_key(i, w) for i in range(sm.LAYERS) for w in WHICH]
    assert got == want
    # quirk #4: the K = 11008 down projection reads the hidden-size up_proj vector
    assert quantize.act_scale_key(0, "mlp.proj") == "model.layers.0.mlp.up_proj"
    assert quantize.act_scale_key(0, "mlp.proj", fix_quirk4=True) == "model.layers.0.mlp.down_proj"


def test_walk_reproduces_the_reference_tensors():
    layers = quantize.quantize_model(sm.state_dict(), act_table(), sm.LAYERS)
    assert list(layers) == [checkpoint.layer_prefix(i, w) for i in range(sm.LAYERS) for w in WHICH]
    for i in range(sm.LAYERS):
        for w in WHICH:
            tag, p = f"L{i}.{w}", layers[checkpoint.layer_prefix(i, w)]
            assert tuple(p["weight"].shape) == tuple(GOLD[f"{tag}.weight_shape"]), tag
            assert_same_outlier_columns(p["fp_ind"], GOLD[f"{tag}.fp_ind"], tag)
            np.testing.assert_array_equal(p["weights_scaling_factor"].view(np.uint16),
                                          GOLD[f"{tag}.weights_scaling_factor"].view(np.uint16), err_msg=tag)
            np.testing.assert_array_equal(p["weight"][0], GOLD[f"{tag}.weight_row0"], err_msg=tag)
            np.testing.assert_array_equal(p["fp_weight"][0].view(np.uint16), GOLD[f"{tag}.fp_weight_row0"].view(np.uint16), err_msg=tag)
            assert hashlib.sha256(np.ascontiguousarray(p["weight"]).tobytes()).hexdigest() == str(GOLD[f"{tag}.weight_sha256"]), tag
            assert hashlib.sha256(np.ascontiguousarray(p["fp_weight"]).tobytes()).hexdigest() == str(GOLD[f"{tag}.fp_weight_sha256"]), tag
    # merged qkv = q | k | v on N (model_config.py:132-138): 64 + 32 + 32 rows
    assert layers[checkpoint.layer_prefix(0, "attention.qkv")]["weight"].shape == (128, sm.HIDDEN)
    # quirk #4 as the reference has it: every outlier index of the K = 11008 projection is < hidden
    assert int(layers[checkpoint.layer_prefix(1, "mlp.proj")]["fp_ind"].max()) < sm.HIDDEN


def test_fix_quirk4_reads_the_down_projections_own_vector():
    acts = act_table()
    p = quantize.quantize_layer(sm.state_dict(), acts, 1, "mlp.proj", fix_quirk4=True)
    want = torch.sort(acts["model.layers.1.mlp.down_proj"])[1][-128:].to(torch.int32).numpy()
    np.testing.assert_array_equal(p["fp_ind"], want)
    assert int(p["fp_ind"].max()) >= sm.HIDDEN   # Llama-2's down_proj outliers do live beyond column 4096
    q = quantize.quantize_layer(sm.state_dict(), acts, 1, "mlp.proj")
    assert not np.array_equal(p["fp_ind"], q["fp_ind"])


@pytest.mark.parametrize("tp", [1, 2])
@pytest.mark.parametrize("layout", ["contiguous", "per_rank_heads"])
def test_checkpoint_round_trip_and_shards(tmp_path, tp, layout):
    sd = sm.state_dict(with_bias=True)
    full = quantize.quantize_model(sd, act_table(), sm.LAYERS, tp_size=tp, out_dir=str(tmp_path), qkv_layout=layout)
    ref = quantize.quantize_model(sd, act_table(), sm.LAYERS)   # unsharded, q | k | v
    for r in range(tp):
        cfg, layers = checkpoint.load_checkpoint(str(tmp_path), rank=r)
        assert cfg["quantization"]["quant_algo"] == "int8_mix" and cfg["mapping"]["tp_size"] == tp
        assert cfg["quantization"]["mixq_qkv_layout"] == layout
        assert sorted(layers) == sorted(checkpoint.iter_mixq_prefixes(sm.LAYERS))
        for prefix, carriers in layers.items():
            got = checkpoint.from_carriers(carriers)
            want = parallel.shard_packed(full[prefix], tp, r) if tp > 1 else full[prefix]
            for name in ("weight", "weights_scaling_factor", "fp_weight", "fp_ind", "qweight"):
                np.testing.assert_array_equal(got[name].view(np.uint8), np.ascontiguousarray(want[name]).view(np.uint8), err_msg=f"{prefix}.{name}")
            if prefix.endswith("attention.qkv"):
                np.testing.assert_array_equal(got["bias"].view(np.uint16), want["bias"].view(np.uint16))
    # per_rank_heads at tp = 2: rank r's rows are [q_r | k_r | v_r] of the unsharded tensors
    if tp == 2 and layout == "per_rank_heads":
        pre = checkpoint.layer_prefix(0, "attention.qkv")
        rows = quantize.rank_major_rows((64, 32, 32), 2)
        np.testing.assert_array_equal(rows[:64], np.r_[0:32, 64:80, 96:112])
        np.testing.assert_array_equal(full[pre]["weight"], ref[pre]["weight"][rows])
        np.testing.assert_array_equal(full[pre]["weights_scaling_factor"].view(np.uint16), ref[pre]["weights_scaling_factor"][rows].view(np.uint16))
        np.testing.assert_array_equal(full[pre]["fp_ind"], ref[pre]["fp_ind"])
    if layout == "contiguous":
        for prefix in full:
            np.testing.assert_array_equal(full[prefix]["weight"], ref[prefix]["weight"])


def test_act_scales_may_be_a_pt_path(tmp_path):
    path = os.path.join(str(tmp_path), "act.pt")
    torch.save(act_table(), path)
    a = quantize.quantize_model(sm.state_dict(), path, 1)
    b = quantize.quantize_model(sm.state_dict(), act_table(), 1)
    for prefix in a:
        np.testing.assert_array_equal(a[prefix]["weight"], b[prefix]["weight"])


def test_bf16_and_fp32_state_dicts_pack_like_their_fp16_rounding():
    """The reference casts every weight to the export dtype first (layer_utils.py:585 `.type(dtype)`): a bf16 / fp32 state dict
    must give the tensors of its fp16-rounded twin."""
    sd16 = sm.state_dict()
    want = quantize.quantize_layer(sd16, act_table(), 0, "mlp.gate")
    for dt in (torch.float32, torch.bfloat16):
        sd = {k: v.to(dt) for k, v in sd16.items()}
        ref = quantize.quantize_layer({k: v.to(torch.float16) for k, v in sd.items()}, act_table(), 0, "mlp.gate")
        got = quantize.quantize_layer(sd, act_table(), 0, "mlp.gate")
        for name in ("weight", "weights_scaling_factor", "fp_weight", "fp_ind", "qweight"):
            np.testing.assert_array_equal(np.ascontiguousarray(got[name]).view(np.uint8), np.ascontiguousarray(ref[name]).view(np.uint8), err_msg=f"{dt} {name}")
        if dt == torch.float32:   # fp16 -> fp32 -> fp16 is the identity
            np.testing.assert_array_equal(got["weight"], want["weight"])


def test_walk_rejects_inconsistent_inputs():
    sd = sm.state_dict(with_bias=True)
    del sd["model.layers.0.self_attn.k_proj.bias"]          # K and V should have a bias when Q has one (model_config.py:146-150)
    with pytest.raises(AssertionError):
        quantize.quantize_layer(sd, act_table(), 0, "attention.qkv")
    with pytest.raises(KeyError):                            # a table without the key the walk reads
        quantize.quantize_layer(sm.state_dict(), {}, 0, "mlp.proj")
    with pytest.raises(AssertionError):                      # 64 | 32 | 32 rows do not split 3 ways
        quantize.rank_major_rows((64, 32, 32), 3)
    with pytest.raises(AssertionError):
        quantize.quantize_model(sm.state_dict(), act_table(), 1, qkv_layout="interleaved")


def test_oracle_team_is_capped_by_the_cpu_quota():
    import oracle
    n = oracle.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert oracle.num_threads() <= n

"""GPU (-m gpu): a checkpoint written by checkpoint.save_checkpoint loads into plugin.MixQLinear and runs."""
import numpy as np
import pytest

from conftest import assert_prefill_parity, make_layer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_checkpoint_to_mixqlinear_forward(tmp_path, oracle):
    from mixq_tensorrt_llm_amd import checkpoint, pack, plugin
    A, W, act = make_layer(40, 256, 512, seed=12)
    packed = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    prefix = checkpoint.layer_prefix(3, "mlp.gate")
    checkpoint.save_checkpoint(str(tmp_path), {prefix: packed})
    _, loaded = checkpoint.load_checkpoint(str(tmp_path))
    layer = checkpoint.load_linear(plugin.MixQLinear(512, 256, device="cuda:0"), loaded[prefix])
    got = layer(torch.from_numpy(A).to("cuda:0")).cpu().numpy()
    assert_prefill_parity(oracle, got, A, packed, "checkpoint -> MixQLinear")


def test_model_walk_checkpoint_layers_run_and_match_the_oracle(tmp_path, oracle):
    """SURVEY §8 row g1 end to end: quantize.quantize_model (HF-named state dict + activation-scale table -> rank0.safetensors),
    then two layers of the WRITTEN checkpoint -- the merged qkv of layer 0 and the K = 11008 down projection of layer 1 (outliers
    picked from the hidden-size vector, quirk #4) -- loaded into MixQLinear and run on the GPU; element-wise against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import synth_model as sm
    from mixq_tensorrt_llm_amd import checkpoint, plugin, quantize
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_walk.npz"))
    acts = {k[4:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("act.")}
    packed = quantize.quantize_model(sm.state_dict(), acts, sm.LAYERS, out_dir=str(tmp_path))
    _, loaded = checkpoint.load_checkpoint(str(tmp_path))
    rng = np.random.default_rng(9)
    for layer_id, which in ((0, "attention.qkv"), (1, "mlp.proj"), (1, "mlp.gate")):
        prefix = checkpoint.layer_prefix(layer_id, which)
        p = packed[prefix]
        N, K = p["weight"].shape
        layer = checkpoint.load_linear(plugin.MixQLinear(K, N, device="cuda:0"), loaded[prefix])
        for M in (40, 3):
            A = rng.standard_normal((M, K)).astype(np.float32)
            A[:, p["fp_ind"]] *= 20
            A = A.astype(np.float16)
            got = layer(torch.from_numpy(A).to("cuda:0")).cpu().numpy()
            if M > 4:
                assert_prefill_parity(oracle, got, A, p, f"{prefix} M={M}")
            else:   # decode path on the interleaved qweight the walk wrote
                from conftest import assert_elementwise, w8a16_slack
                from mixq_tensorrt_llm_amd import _lib
                q_un = np.empty((K, N), np.int8)
                _lib.load().mixq_unprocess_weights_int8(q_un.ctypes.data, np.ascontiguousarray(p["qweight"]).ctypes.data, K, N)
                want = oracle.w8a16_gemv(A, q_un, p["weights_scaling_factor"])
                assert_elementwise(got, want, w8a16_slack(A, q_un, p["weights_scaling_factor"]), f"{prefix} decode")

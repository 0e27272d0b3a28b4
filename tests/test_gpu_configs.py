"""GPU (-m gpu): the five BASELINE.json configs as parity cases (the bench runs config 2/3; these are the others and
the per-GPU shapes of the TP config), each through MixQLinear -> C ABI -> HIP, against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REL_TOL = 1e-3


def rel_err(got, want):
    g, w = got.astype(np.float64), want.astype(np.float64)
    return np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)


def run_layer(A, p, bias=None):
    from mixq_tensorrt_llm_amd import plugin
    M, K = A.shape
    N = p["weight"].shape[0]
    layer = plugin.MixQLinear(K, N, bias=bias is not None, device="cuda:0").load(p)
    if bias is not None:
        layer.bias = torch.from_numpy(bias).to("cuda:0")
    return layer(torch.from_numpy(A).to("cuda:0")).cpu().numpy()


def synth(M, N, K, act, seed):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    A = rng.standard_normal((M, K)).astype(np.float32)
    ind = np.argsort(act, kind="stable")[-128:]
    A[:, ind] *= 20.0
    return A.astype(np.float16), W


def test_config0_single_4096_linear_bs32_real_act_scales(oracle):
    """configs[0]: 4096 x 4096 MixQ linear, bs=32, int8_mix; outlier columns from the reference's own activation
    statistics (act_scales/Llama-2-1b.pt layer-0 q_proj, committed fixture)."""
    act = np.load(os.path.join(GOLDEN, "act_scales_llama.npz"))["scales_0"]
    A, W = synth(32, 4096, 4096, act, 0)
    p = oracle.pack_linear_weights(W, act)
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    got = run_layer(A, p)
    assert rel_err(got, want) < REL_TOL
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(got, ref) < 0.05


@pytest.mark.parametrize("name,N,K", [("qkv", 12288, 4096), ("gate", 11008, 4096), ("proj", 4096, 11008)])
def test_config1_llama2_7b_shapes(oracle, name, N, K):
    """configs[1]/[2]: the three Llama-2-7B shapes at a token count the oracle finishes in seconds (M = 96); mlp.proj
    picks its outliers from a K=4096 vector (indices only in [0, 4096))."""
    rng = np.random.default_rng(N)
    act = np.abs(rng.standard_normal(4096)).astype(np.float32)
    A, W = synth(96, N, K, np.concatenate([act, np.zeros(K - 4096, np.float32)]) if K > 4096 else act, N + K)
    p = oracle.pack_linear_weights(W, act)
    assert p["fp_ind"].max() < 4096
    got = run_layer(A, p)
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    assert rel_err(got, want) < REL_TOL


@pytest.mark.parametrize("name,N,K,bias", [("qkv", 4608, 3584, True), ("gate", 18944, 3584, False),
                                           ("proj", 3584, 18944, False)])
def test_config3_qwen2_7b_fpA_intB_outliers(oracle, name, N, K, bias):
    """configs[3]: Qwen2-7B shapes (qkv has a bias, added after the op like plugin.py:158-160) with the outlier weights
    taken from the DEQUANTISED int8 columns (P-flavour `fpA_intB` mode: q_weight[:, ind] * scale_col,
    MixQ/src/mixquant/modules/linear.py:204) instead of the original fp16 columns."""
    from mixq_tensorrt_llm_amd import pack
    rng = np.random.default_rng(K + N)
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    A, W = synth(64, N, K, act, N)
    p = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act), outlier_weights="int8")
    o = oracle.pack_linear_weights(W, act)
    q_full = oracle.quantize_weight(W, o["weights_scaling_factor"])            # int8 of the UN-zeroed weight
    want_fpw = (q_full[:, o["fp_ind"]].astype(np.float16) * o["weights_scaling_factor"][:, None]).astype(np.float16)
    assert np.array_equal(p["fp_weight"].view(np.uint16), want_fpw.view(np.uint16))
    b = (rng.standard_normal(N) * 0.1).astype(np.float16) if bias else None
    got = run_layer(A, p, b)
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    if bias:
        want = (want.astype(np.float16) + b[None, :]).astype(np.float16)
    assert rel_err(got, want) < REL_TOL


@pytest.mark.parametrize("name,N,K", [("qkv/8", 1280, 8192), ("gate/8", 3584, 8192), ("proj/8", 1024, 28672)])
def test_config4_llama2_70b_tp8_shard_shapes(oracle, name, N, K):
    """configs[4]: the per-GPU row shards of Llama-2-70B at TP=8 -- K = 8192 and 28672."""
    rng = np.random.default_rng(N + K)
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    A, W = synth(48, N, K, act, K)
    p = oracle.pack_linear_weights(W, act)
    got = run_layer(A, p)
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    assert rel_err(got, want) < REL_TOL
    # decode on the same shard (M = 2) through the interleaved qweight
    got2 = run_layer(A[:2], p)
    q_un = oracle.eetq_symmetric_quantize(W.T.copy())[0]
    want2 = oracle.w8a16_gemv(A[:2], q_un, p["weights_scaling_factor"])
    assert rel_err(got2, want2) < REL_TOL

"""GPU (-m gpu): the five BASELINE.json configs as parity cases (the bench runs config 2/3; these are the others and
the per-GPU shapes of the TP config), each through MixQLinear -> C ABI -> HIP, against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_elementwise, prefill_slack, ulp16 as ulp16_of, ulp_histogram, w8a16_slack

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REL_TOL = 1e-3


@pytest.fixture
def lib():
    from mixq_tensorrt_llm_amd import _lib
    return _lib.load()


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
    assert_elementwise(got, want, prefill_slack(parts, A, p), "config 0")
    assert ulp_histogram(got, want)["<=1"] > 0.999
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
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, prefill_slack(parts, A, p), f"config 1 {name}")
    assert ulp_histogram(got, want)["<=1"] > 0.999


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
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    slack = prefill_slack(parts, A, p)
    if bias:   # one more fp16 addition per element after the operator: a 1-ulp difference before it can move its result by
        slack = slack + ulp16_of(want)  # one ulp of the un-biased value
        want = (want.astype(np.float16) + b[None, :]).astype(np.float16)
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, slack, f"config 3 {name}")


@pytest.mark.parametrize("name,N,K", [("qkv/8", 1280, 8192), ("gate/8", 3584, 8192), ("proj/8", 1024, 28672)])
def test_config4_llama2_70b_tp8_shard_shapes(oracle, name, N, K):
    """configs[4]: the per-GPU row shards of Llama-2-70B at TP=8 -- K = 8192 and 28672."""
    rng = np.random.default_rng(N + K)
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    A, W = synth(48, N, K, act, K)
    p = oracle.pack_linear_weights(W, act)
    got = run_layer(A, p)
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, prefill_slack(parts, A, p), f"config 4 {name}")
    # decode on the same shard (M = 2) through the interleaved qweight
    got2 = run_layer(A[:2], p)
    q_un = oracle.eetq_symmetric_quantize(W.T.copy())[0]
    want2 = oracle.w8a16_gemv(A[:2], q_un, p["weights_scaling_factor"])
    assert rel_err(got2, want2) < REL_TOL
    assert_elementwise(got2, want2, w8a16_slack(A[:2], q_un, p["weights_scaling_factor"]), f"config 4 decode {name}")


# every (N, K) BASELINE.json's configs name: Llama-2-7B, Qwen2-7B-Instruct, one GPU's row shard of Llama-2-70B at TP = 8
BASELINE_SHAPES = [("llama2-7b qkv", 12288, 4096), ("llama2-7b gate", 11008, 4096), ("llama2-7b proj", 4096, 11008),
                   ("qwen2-7b qkv", 4608, 3584), ("qwen2-7b gate", 18944, 3584), ("qwen2-7b proj", 3584, 18944),
                   ("llama2-70b/8 qkv", 1280, 8192), ("llama2-70b/8 gate", 3584, 8192), ("llama2-70b/8 proj", 1024, 28672)]


@pytest.mark.parametrize("M", [65536, 16384, 4096])
@pytest.mark.parametrize("name,N,K", BASELINE_SHAPES)
def test_bench_configuration_at_full_size_on_sampled_rows(oracle, lib, name, N, K, M):
    """The BENCH configuration itself (configs[2]: 65536-token chunks of Llama-2-7B) and every other (N, K) of BASELINE.json's
    configs at prefill size (Qwen2-7B and the Llama-2-70B shards met the oracle only at M = 48-64): ONE
    mixq_enqueue at M x N x K on bench.py's own synthetic data -- whole rounds of 256 x 256 tiles and the XCD remap at 65536
    rows; at 16384 / 4096 rows the narrow shards take the 256 x 256 tiles with K split over workgroups (`pp_kernel<SPLITK>`) --
    against the oracle on 64 sampled rows (rows are independent: TsinghuaMixQPlugin.cpp:518-532 never mixes tokens).  On
    those rows: qA, sA and the int32 accumulators of the full-size GEMM bit-exact, the fp16 output within the element-wise
    bound and the north-star 1e-3."""
    import bench
    from mixq_tensorrt_llm_amd import mixlib, plugin
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(99 + N + (M >> 10))
    t = bench.synth_layer(N, K, dev, gen)
    A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
    rng = np.random.default_rng(N + M)
    rows = np.unique(np.concatenate([[0, 1, 255, 256, 257, M // 2 - 1, M // 2, M - 257, M - 256, M - 2, M - 1],
                                     rng.integers(0, M, 53)]))[:64]
    ridx = torch.from_numpy(rows).to(dev)
    # the operator through the plugin object (same call as bench.py: mixq_enqueue on a 65536-row chunk)
    plug = plugin.MixQPlugin.create(M, N, K)
    out = plug.enqueue([A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"],
                        t["weights_scaling_factor"]])
    torch.cuda.synchronize()
    kernel = lib.mixq_debug_last_gemm_kernel().decode()
    # the forms this test is there to cover: the plain 256 x 256 ping-pong kernel at the bench's chunk, and the same tiles with K
    # split over workgroups wherever the plugin carves exchange scratch for the call (16384 rows: Qwen2-7B proj, the 70B qkv shard;
    # 4096 rows: every narrow shard)
    if M == 65536 and N >= 3584 and torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert kernel.startswith("gemm_w8a8o16_pp_kernel (256x256"), kernel
    if M > 128 and lib.mixq_gemm_scratch_size(M, N, K) > 0:
        assert lib.mixq_enqueue_scratch_size(M, N, K) > 0 and "SPLITK" in kernel, kernel
    got = out[ridx].cpu().numpy()
    W8 = t["weight"].view(torch.int8).reshape(N, K)
    # oracle on the sampled rows
    A_s = A[ridx].cpu().numpy()
    want, parts = oracle.linear_prefill(A_s, W8.cpu().numpy(), t["weights_scaling_factor"].cpu().numpy(),
                                        t["fp_weight"].cpu().numpy(), t["ind_i32"].cpu().numpy(), return_parts=True)
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, prefill_slack(parts, A_s, dict(fp_ind=t["ind_i32"].cpu().numpy().astype(np.int64),
                                                                 fp_weight=t["fp_weight"].cpu().numpy())),
                       f"{name} at {M} rows ({kernel})")
    h = ulp_histogram(got, want)
    assert h["<=1"] > 0.995, h
    # integer half of the path at full size: quantiser rows and the int32 accumulators of the 65536-row GEMM, bit for bit
    sA = torch.empty(M, dtype=torch.float16, device=dev)
    qA = mixlib.FindRowScale(A, sA, M, K, 8)
    assert np.array_equal(qA[ridx].cpu().numpy(), parts["qA"])
    assert np.array_equal(sA[ridx].cpu().numpy().view(np.uint16), parts["sA"].view(np.uint16))
    acc = mixlib.gemm(qA, W8, M, N, K)          # int32 [M, N]: the same ping-pong main loop, raw accumulators
    torch.cuda.synchronize()
    assert np.array_equal(acc[ridx].cpu().numpy(), parts["acc"])
    del acc, out
    torch.cuda.empty_cache()


def test_one_million_tokens_in_one_call_on_sampled_rows(oracle):
    """Maximum size: BASELINE configs[2]'s 512 x 2048 = 1 048 576 tokens as ONE mixq_enqueue (the reference sizes its workspace
    with `int` and overflows here; this library's sizes are size_t).  M K = 2^32 and M N = 2^32 exactly on
    4096 x 4096: every row offset of the last rows needs more than 32 bits in the quantiser, the GEMM loads and the stores.
    Oracle on 48 sampled rows incl. the first and last of the call and both sides of the 2^31- and 2^32-byte marks."""
    import bench
    from mixq_tensorrt_llm_amd import plugin
    M, N, K = 1 << 20, 4096, 4096
    dev = torch.device("cuda:0")
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs ~25 GB of device memory")
    gen = torch.Generator(device=dev).manual_seed(4242)
    t = bench.synth_layer(N, K, dev, gen)
    A = torch.empty((M, K), dtype=torch.float16, device=dev)
    for m0 in range(0, M, 1 << 16):                      # (chunked fill: bounds the generator's fp32 temporaries)
        A[m0:m0 + (1 << 16)] = bench.synth_activation(1 << 16, K, t["ind_i32"], dev, gen)
    marks = [0, 1, 255, 256, (1 << 18) - 1, 1 << 18, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, M - 257, M - 256, M - 2, M - 1]
    rows = np.unique(np.concatenate([marks, np.random.default_rng(7).integers(0, M, 35)]))[:48]
    ridx = torch.from_numpy(rows).to(dev)
    plug = plugin.MixQPlugin.create(M, N, K)
    out = plug.enqueue([A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"],
                        t["weights_scaling_factor"]])
    torch.cuda.synchronize()
    assert tuple(out.shape) == (M, N)
    got = out[ridx].cpu().numpy()
    A_s = A[ridx].cpu().numpy()
    W8 = t["weight"].view(torch.int8).reshape(N, K)
    want, parts = oracle.linear_prefill(A_s, W8.cpu().numpy(), t["weights_scaling_factor"].cpu().numpy(),
                                        t["fp_weight"].cpu().numpy(), t["ind_i32"].cpu().numpy(), return_parts=True)
    assert np.isfinite(got).all()
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, prefill_slack(parts, A_s, dict(fp_ind=t["ind_i32"].cpu().numpy().astype(np.int64),
                                                                 fp_weight=t["fp_weight"].cpu().numpy())),
                       "one million tokens in one call")
    # rows are independent: the same rows as a 48-token call of their own give the same bits
    small = plugin.MixQPlugin.create(64, N, K).enqueue([A[ridx].contiguous(), t["weight"], t["weights_scaling_factor"],
                                                        t["fp_weight"], t["fp_ind"], t["qweight"],
                                                        t["weights_scaling_factor"]])
    torch.cuda.synchronize()
    assert rel_err(small.cpu().numpy(), want) < REL_TOL
    del out, A
    torch.cuda.empty_cache()

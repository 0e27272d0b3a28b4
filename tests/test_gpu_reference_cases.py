"""GPU (-m gpu): the reference's OWN eyeball cases as parity tests (VERDICT r5 missing #1 / next #2).

The reference ships no test suite for this path; what it ships are five scripts that print two results side by side.  Each
is restated here with its exact shapes, seed, index set and torch expressions, as an assertion:

  MixQ/src/benchmark/fuse_scale_benchmerk.py:19-53   32 x 12288, seed 0, ind = [2, 5, 8, 9]: ExtractOutliersAndSetToZeros + the torch
                                                     row scale + Int8quantize  ==  FindRowScaleFusedExtracOutliers
  MixQ/src/benchmark/scale_benchmark.py:9-38         sum(torch.max(x.abs(), dim=1)[0] / 127.0 - x_scale) over FindRowScale's scales (== 0)
  MixQ/src/benchmark/layer_benchmark.py:15-69        MixLinear_GEMM.from_linear on 512 x 4096 -> 10240, x = randn / 1.3, weights in {-2, -1, 0}
  EETQ/examples/layers/test_w8a16_gemm.py:19-63      M = 1, N = 13824, K = 5120: quant_weights route == preprocess_weights route, vs torch.matmul
  EETQ/examples/layers/test_qlinear.py:19-36         128 x 1024 -> 4096 nn.Linear: torch.allclose(out, ref, atol=1e-2)

The scripts draw their inputs with the CUDA generator; here the same calls run on the CPU generator with the same seed (the draw differs,
the case -- shape, distribution, seed, index set -- is the reference's).  Every case is also checked against the oracle."""
import numpy as np
import pytest

from conftest import assert_elementwise, ulp16, w8a16_slack

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

DEV = "cuda:0"


def bits(t):
    return np.ascontiguousarray(t.detach().cpu().numpy()).view(np.uint16)


def test_fuse_scale_benchmerk_case(oracle):
    """fuse_scale_benchmerk.py:19-53: the unfused route (run1) and the fused kernel must print the same q and the same outliers."""
    from mixq_tensorrt_llm_amd import mixlib
    from mixq_tensorrt_llm_amd.mixlinear import MixLibCache
    torch.manual_seed(0)
    cache = MixLibCache()
    M, N = 32, 12288
    inputs = torch.randn((M, N), dtype=torch.float16).to(DEV)
    inputs2 = torch.clone(inputs)
    A0 = inputs.cpu().numpy().copy()
    ind = torch.as_tensor([2, 5, 8, 9], dtype=torch.int32, device=DEV)
    # run1 (:22-40), one trip of its loop
    activation_outliers = mixlib.ExtractOutliersAndSetToZeros(ind, inputs)
    x_scale = torch.max(inputs.abs(), dim=1)[0] / 127.0
    q_xquant = mixlib.Int8quantize(inputs, x_scale)
    # :45
    q_xquant_new, activation_outliers_new = mixlib.FindRowScaleFusedExtracOutliers(inputs2, cache.x_scale, ind, len(ind), M, N)
    torch.cuda.synchronize()
    assert torch.equal(q_xquant, q_xquant_new)
    assert np.array_equal(bits(activation_outliers), bits(activation_outliers_new))
    assert np.array_equal(bits(inputs), bits(inputs2)), "both routes zero the same columns"
    # the commented-out check of :55 (scale_benchmark.py:38 runs it): the fused kernel's scales against the torch expression
    assert np.array_equal(bits(x_scale), bits(cache.x_scale[0:M].squeeze(1)))
    assert float(torch.sum(torch.max(inputs.abs(), dim=1)[0].squeeze(0)[0:M] / 127.0 - cache.x_scale[0:M].squeeze(0).T)) == 0.0
    # both against the oracle
    Az = A0.copy()
    fo = oracle.extract_outliers(Az, np.array([2, 5, 8, 9], np.int32), set_zero=True)
    qz, sz = oracle.quant_rows(Az)
    assert np.array_equal(bits(activation_outliers), fo.view(np.uint16))
    assert np.array_equal(q_xquant.cpu().numpy(), qz)
    assert np.array_equal(bits(x_scale), sz.view(np.uint16))
    assert np.array_equal(bits(inputs), Az.view(np.uint16))


def test_scale_benchmark_case(oracle):
    """scale_benchmark.py:9-38: Int8quantize with the torch scale against FindRowScale; the script's last line prints
    sum(torch scale - kernel scale), which must be 0 -- here bit for bit."""
    from mixq_tensorrt_llm_amd import mixlib
    from mixq_tensorrt_llm_amd.mixlinear import MixLibCache
    torch.manual_seed(0)
    cache = MixLibCache()
    M, N = 32, 12288
    inputs = torch.randn((M, N), dtype=torch.float16).to(DEV)
    x_scale = torch.max(inputs.abs(), dim=1)[0] / 127.0
    q_xquant = mixlib.Int8quantize(inputs, x_scale)
    q_xquant_new = mixlib.FindRowScale(inputs, cache.x_scale, M, N)
    torch.cuda.synchronize()
    assert float(torch.sum(torch.max(inputs.abs(), dim=1)[0].squeeze(0)[0:M] / 127.0 - cache.x_scale[0:M].squeeze(0).T)) == 0.0
    assert np.array_equal(bits(x_scale), bits(cache.x_scale[0:M].squeeze(1)))
    assert torch.equal(q_xquant, q_xquant_new)
    qo, so = oracle.quant_rows(inputs.cpu().numpy())
    assert np.array_equal(q_xquant_new.cpu().numpy(), qo) and np.array_equal(bits(x_scale), so.view(np.uint16))


def test_layer_benchmark_case(oracle):
    """layer_benchmark.py:15-69: `mix_mod = MixLinear_GEMM.from_linear(baseline_mod, False, False, cache)` on a 4096 -> 10240 layer whose
    weights are randint(-2, 1) (so {-2, -1, 0}), driven with 512 rows of randn / 1.3 -- the first calls (dynamic outlier detection
    on; no |x| reaches sigma = 6 here, so the state machine ends with an empty set) and the steady state, against the oracle's
    restatement of the same module and against the fp16 layer the script times next to it."""
    from mixq_tensorrt_llm_amd import mixlinear
    torch.manual_seed(0)
    input_size, feature_dim_in, feature_dim_out = 512, 4096, 10240
    cache = mixlinear.MixLibCache(input_size)
    x = (torch.randn((input_size, feature_dim_in), dtype=torch.float16) / 1.3)
    w = torch.randint_like(torch.empty((feature_dim_out, feature_dim_in), dtype=torch.float16), low=-2, high=1).to(torch.float16)
    mix_mod = mixlinear.MixLinear_GEMM.from_linear(w, None, 8, False, cache, dev=DEV)
    A = x.numpy().copy()
    W = w.numpy()
    # quantize-time half (linear.py:113-120) against the oracle's restatement
    sW = (np.abs(W.astype(np.float32)).max(axis=1).astype(np.float16) / np.float16(127)).astype(np.float16)
    assert np.array_equal(bits(mix_mod.scale_col.reshape(-1)), sW.view(np.uint16))
    with np.errstate(all="ignore"):
        Wq = np.rint((W / sW[:, None]).astype(np.float16).astype(np.float32)).astype(np.int8)
    assert np.array_equal(mix_mod.q_weight.cpu().numpy(), Wq)
    outs = []
    for _ in range(3):   # (the script calls the module 20 + 100 times on the same x; calls 1-2 run with add_outliers on)
        outs.append(mix_mod(x.clone().to(DEV), unfused=True).cpu().numpy())
    torch.cuda.synchronize()
    assert mix_mod.ind.shape[0] == 0 and not mix_mod.add_outliers
    assert all(np.array_equal(o.view(np.uint16), outs[0].view(np.uint16)) for o in outs[1:])
    # the oracle: per-token quantisation, exact int32 sums, the dequant epilogue with no addend
    qA, sA = oracle.quant_rows(A)
    acc = oracle.gemm_s8s8s32(qA, Wq)
    want = oracle.dequant_epilogue(acc, sA, sW, None)
    assert np.array_equal(outs[0].view(np.uint16), want.view(np.uint16)), "int8 path with no outliers is bit-exact"
    # and the fp16 layer next to it in the script (quantisation error only: 8-bit activations, exactly representable weights)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    err = np.abs(outs[0].astype(np.float64) - ref)
    assert err.max() < 0.02 * np.abs(ref).max()


def test_eetq_w8a16_gemm_case(oracle):
    """EETQ/examples/layers/test_w8a16_gemm.py:19-63 (M = 1, N = 13824, K = 5120, seed 1, torch.rand operands): `out1` (weights from
    quant_weights) and `out2` (weights from preprocess_weights of the unprocessed int8 matrix) print the same tensor, and
    `torch.sum(output - out_torch)` is small."""
    from EETQ import preprocess_weights, quant_weights, w8_a16_gemm
    from mixq_tensorrt_llm_amd import _lib
    torch.manual_seed(1)
    np.random.seed(1)
    M, N, K = 1, 13824, 5120
    inp = torch.rand(M, K, dtype=torch.float16)
    torch_weights_cpu = torch.rand(K, N, dtype=torch.float16)
    ref_torch_weights, processed_torch_weights, torch_weight_scales = quant_weights(torch_weights_cpu, torch.int8, True)
    out1 = w8_a16_gemm(inp.to(DEV), processed_torch_weights.to(DEV), torch_weight_scales.to(DEV))
    processed_w = preprocess_weights(ref_torch_weights)
    assert torch.equal(processed_w, processed_torch_weights)
    out2 = w8_a16_gemm(inp.to(DEV), processed_w.to(DEV), torch_weight_scales.to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(bits(out1), bits(out2))
    # the oracle: symmetric per-column quantisation (cutlass_preprocessors.cc:573-660), the interleave, the GEMV's fp32 sum
    q_o, sc_o = oracle.eetq_symmetric_quantize(torch_weights_cpu.numpy())
    assert np.array_equal(ref_torch_weights.numpy(), q_o) and np.array_equal(bits(torch_weight_scales), sc_o.view(np.uint16))
    q_un = np.empty((K, N), np.int8)
    _lib.load().mixq_unprocess_weights_int8(q_un.ctypes.data, processed_w.numpy().view(np.uint8).ctypes.data, K, N)
    assert np.array_equal(q_un, q_o)
    want = oracle.w8a16_gemv(inp.numpy(), q_o, sc_o)
    assert_elementwise(out1.cpu().numpy(), want, w8a16_slack(inp.numpy(), q_o, sc_o), "test_w8a16_gemm.py case vs oracle")
    # `print(torch.sum(output - out_torch))` (:60): against the fp16 product of the unquantised weights
    out_torch = (inp.float() @ torch_weights_cpu.float()).to(torch.float16)
    diff = (out1.cpu().float() - out_torch.float()).abs()
    assert float(diff.max()) < 5e-3 * float(out_torch.float().abs().max())   # 8-bit weights: <= 1/256 relative per column


def test_eetq_qlinear_case(oracle):
    """EETQ/examples/layers/test_qlinear.py:19-36: W8A16Linear.from_torch(nn.Linear(1024, 4096)) on a 128 x 1024 torch.rand input;
    the script prints torch.allclose(output, output_torch, atol=1e-2) -- asserted here with the script's own tolerance."""
    from EETQ import quant_weights, w8_a16_gemm
    torch.manual_seed(1)
    np.random.seed(1)
    M, N, K = 128, 4096, 1024
    torch_linear = torch.nn.Linear(K, N, bias=False, dtype=torch.float16)
    inp = torch.rand(M, K, dtype=torch.float16)
    # W8A16Linear.from_torch (EETQ/python/eetq/modules/qlinear.py): quant_weights of weight^T, int8, then w8_a16_gemm in forward
    weight_t = torch_linear.weight.detach().t().contiguous()
    unprocessed, qweight, scales = quant_weights(weight_t, torch.int8, True)
    output = w8_a16_gemm(inp.to(DEV), qweight.to(DEV), scales.to(DEV)).cpu()
    output_torch = (inp.float() @ weight_t.float()).to(torch.float16)
    assert torch.allclose(output, output_torch, atol=1e-2)
    want = oracle.w8a16_gemv(inp.numpy(), unprocessed.numpy(), scales.numpy())
    assert_elementwise(output.numpy(), want, w8a16_slack(inp.numpy(), unprocessed.numpy(), scales.numpy()) + ulp16(want) * 0,
                       "test_qlinear.py case vs oracle")

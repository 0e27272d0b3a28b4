"""GPU (-m gpu): the 4-bit (W4A4) flavour -- FindRowScale(bit=4), int4FusedDequantize, unpack_int4_to_fp16 and the
bit = 4 branch of MixLinear_GEMM (MixQ/src/mixquant/modules/linear.py, quantkernel/mix_cuda/cult.cu) through the C ABI,
against the oracle's numpy restatement.  Integer results bit-exact; fp16 output bit-exact given its addend."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


@pytest.mark.parametrize("M,K", [(1, 256), (33, 4096), (7, 11008), (5, 64)])
def test_find_row_scale_4bit_bit_exact(oracle, M, K):
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(M + K)
    A = (rng.standard_normal((M, K)) * rng.uniform(0.05, 20)).astype(np.float16)
    if M > 4:
        A[1] = 0                        # zero row: scale 0, 0/0 = NaN -> 0
        A[2, 3] = np.nan                # NaN element dropped from the max, quantised to 0
        A[3, :] = np.float16(6e-8)      # scale underflows to 0: x/0 = inf -> INT_MAX & 0xF = -1
        A[4, 7] = np.inf                # inf amax -> inf scale -> finite/inf = 0, inf/inf = NaN -> 0
    s = torch.empty(M, dtype=torch.float16, device="cuda:0")
    q = mixlib.FindRowScale(dev(A), s, M, K, 4)
    qo, so = oracle.quant4_rows(A)
    assert q.dtype == torch.uint8 and tuple(q.shape) == (M, K // 2)
    assert np.array_equal(bits(s.cpu().numpy()), bits(so))
    assert np.array_equal(q.cpu().numpy(), qo)


def test_unpack_int4_to_int8_and_columns(oracle):
    from mixq_tensorrt_llm_amd import _lib, mixlib
    rng = np.random.default_rng(9)
    R, C = 96, 4096
    packed = rng.integers(0, 256, (R, C // 2), dtype=np.uint8)
    src = dev(packed)
    dst = torch.empty((R, C), dtype=torch.int8, device="cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.load().mixq_unpack_int4_to_int8(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()),
                                                    packed.size, st), "unpack")
    want = oracle.unpack_i4(packed)
    assert np.array_equal(dst.cpu().numpy(), want)
    ind = np.array([0, 1, 4095, 4094, 77, 78, 2048], np.int32)
    got = mixlib.unpack_int4_to_fp16(src, dev(ind)).cpu().numpy()
    assert np.array_equal(got, want[:, ind].astype(np.float16))


@pytest.mark.parametrize("M,N,K,silu", [(40, 512, 1024, False), (300, 1024, 4096, False), (64, 256, 512, True),
                                        (2048, 4096, 4096, False)])
def test_int4_fused_dequantize_bit_exact(oracle, M, N, K, silu):
    """s4 x s4 -> s32 (exact) + the dequant epilogue with an fp16 addend: identical bits to the oracle."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-8, 8, (M, K), dtype=np.int8)
    b = rng.integers(-8, 8, (N, K), dtype=np.int8)
    sa = (rng.random(M) * 0.5 + 0.01).astype(np.float16)
    sb = (rng.random(N) * 1e-2 + 1e-4).astype(np.float16)
    y = (rng.standard_normal((M, N)) * 0.3).astype(np.float16)
    ap, bp = oracle.pack_i4(a), oracle.pack_i4(b)
    assert np.array_equal(oracle.unpack_i4(ap), a)
    fn = mixlib.int4FusedDequantizeSilu if silu else mixlib.int4FusedDequantize
    got = fn(dev(ap), dev(bp), dev(sa.reshape(M, 1)), dev(sb.reshape(1, N)), dev(y), M, N, K // 2).cpu().numpy()
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, C=y, silu=silu)
    if silu:
        g, w = got.astype(np.float64), want.astype(np.float64)
        assert np.abs(g - w).max() / np.abs(w).max() < 1e-3      # __expf vs expf: not bit-pinned (as for int8)
    else:
        assert np.array_equal(bits(got), bits(want))


def test_mixlinear_4bit_from_linear_and_forward(oracle):
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(17)
    N, K, M, FP = 768, 2048, 96, 256
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    scales = np.abs(rng.standard_normal(K)).astype(np.float32)
    cache = mixlinear.MixLibCache(inputdim=256, sigma=6, bit=4, device="cuda:0")
    layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), bit=4, cache=cache, dev="cuda:0",
                                                 layer_scales=torch.from_numpy(scales), fp_features_num=FP)
    qp, sc, ind, wc = oracle.mixlinear4_from_linear(W, scales, FP)
    assert np.array_equal(layer.ind.cpu().numpy(), ind)
    assert np.array_equal(bits(layer.scale_col.cpu().numpy().reshape(-1)), bits(sc))
    assert np.array_equal(layer.q_weight.cpu().numpy(), qp)
    assert np.array_equal(bits(layer.weight_cache.cpu().numpy()), bits(wc))
    x = (rng.standard_normal((M, K)) * 0.8).astype(np.float16)
    x[:, ind] *= 12.0                       # the calibrated outlier columns carry the large activations
    x_ref = x.copy()
    xt = dev(x)
    got = layer.forward(xt, cache, True).cpu().numpy()
    want = oracle.mixlinear4_forward(qp, sc, ind, wc, x_ref)
    assert layer.ind.numel() == FP, "no dynamic growth expected: the remaining activations stay below sigma"
    assert np.array_equal(bits(xt.cpu().numpy()), bits(x_ref))
    g, w = got.astype(np.float64), want.astype(np.float64)
    assert np.abs(g - w).max() / np.abs(w).max() < 1e-3
    # 4-bit weights + 4-bit activations are coarse, but the result must still track the fp product
    ref = x.astype(np.float64) @ W.astype(np.float64).T
    assert np.abs(g - ref).max() / np.abs(ref).max() < 0.35


def test_fused_norm_4bit_producer(oracle):
    """layernorm_forward_cuda_extract_outliers_int4 == RMSNorm -> extract + zero -> FindRowScale(bit = 4), one pass."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(31)
    M, K = 37, 4096
    x = (rng.standard_normal((M, K)) * 3).astype(np.float16)
    gamma = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    ind = np.sort(rng.choice(K, 64, replace=False)).astype(np.int32)
    x[:, ind] *= 10
    out = torch.empty((M, K), dtype=torch.float16, device="cuda:0")
    sc = torch.empty((M, 1), dtype=torch.float16, device="cuda:0")
    outl, q4 = mixlib.layernorm_forward_cuda_extract_outliers_int4(dev(x), dev(gamma), out, 1e-6, dev(ind), sc)
    o_ref, outl_ref, _, _ = oracle.rmsnorm_extract_quant(x, gamma, 1e-6, ind)
    out_np = out.cpu().numpy()
    # the normalised row is within one fp16 ulp of the oracle's (the sum of squares is reduced in another order) ...
    denom = np.maximum(np.abs(o_ref.astype(np.float64)), 1e-3)
    assert (np.abs(out_np.astype(np.float64) - o_ref.astype(np.float64)) / denom).max() < 1.1e-3
    assert np.all(out_np[:, ind] == 0)
    assert (np.abs(outl.cpu().numpy().astype(np.float64) - outl_ref.astype(np.float64))
            / np.maximum(np.abs(outl_ref.astype(np.float64)), 1e-3)).max() < 1.1e-3
    # ... and given the GPU's normalised row, scale and packed 4-bit rows are bit-exact
    q_ref, s_ref = oracle.quant4_rows(np.ascontiguousarray(out_np))
    assert np.array_equal(bits(sc.cpu().numpy().reshape(-1)), bits(s_ref))
    assert np.array_equal(q4.cpu().numpy(), q_ref)
    # ExtractOutliers (non-zeroing gather)
    xa = dev(x)
    got = mixlib.ExtractOutliers(dev(ind), xa).cpu().numpy()
    assert np.array_equal(bits(got), bits(x[:, ind])) and np.array_equal(bits(xa.cpu().numpy()), bits(x))


# ---- packed-int4 weight stream for decode batches (csrc/int4_gemm_kernels.hip; VERDICT r4 missing #2) ---------------------------
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("N,K", [(512, 1024), (4096, 4096), (272, 2080), (12288, 4096), (1024, 28672), (4096, 11008), (144, 256)])
@pytest.mark.parametrize("M", [1, 5, 16, 17, 33, 48, 64])
def test_int4_weight_stream_bit_exact(oracle, M, N, K, silu):
    """M <= 64: ONE launch reads both operands PACKED (N K / 2 weight bytes) and widens the nibbles in registers (x 16 per operand,
    accumulators shifted back by 8): int32 sums, and with them the fp16 outputs, bit-identical to the oracle's s4 x s4 GEMM +
    dequant epilogue AND to the unpack-to-int8 route this replaces; no workspace.  K = 2080: a partial last 128-element step."""
    from mixq_tensorrt_llm_amd import _lib, mixlib
    if silu and (N > 4096 or M not in (5, 33, 64)):
        pytest.skip("SiLU epilogue: a subset")
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-8, 8, (M, K), dtype=np.int8)
    b = rng.integers(-8, 8, (N, K), dtype=np.int8)
    if M > 1:
        a[1] = -8                       # extreme row x extreme feature: |sum| = 64 K, 256 x that inside the kernel
        b[min(3, N - 1)] = -8
    sa = (rng.random(M) * 0.5 + 0.01).astype(np.float16)
    sb = (rng.random(N) * 1e-2 + 1e-4).astype(np.float16)
    y = (rng.standard_normal((M, N)) * 0.3).astype(np.float16)
    ap, bp = dev(oracle.pack_i4(a)), dev(oracle.pack_i4(b))
    fn = mixlib.int4FusedDequantizeSilu if silu else mixlib.int4FusedDequantize
    got = fn(ap, bp, dev(sa.reshape(M, 1)), dev(sb.reshape(1, N)), dev(y), M, N, K // 2)
    torch.cuda.synchronize()
    assert b"gemm_skinny_s4_kernel" in lib.mixq_debug_last_gemm_kernel()
    got = got.cpu().numpy()
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, C=y, silu=silu)
    if silu:
        g, w = got.astype(np.float64), want.astype(np.float64)
        assert np.abs(g - w).max() / np.abs(w).max() < 1e-3      # __expf vs expf: not bit-pinned (as for int8)
    else:
        assert np.array_equal(bits(got), bits(want))
    # the route this replaces (both operands unpacked to int8 in a workspace, then the int8 kernels): same bits, SiLU included
    D = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_int4_fused_workspace_size(M, N, K // 2) == 0          # the stream route needs none ...
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.mixq_debug_set_gemm_variant(871)
    ws = torch.empty(lib.mixq_int4_fused_workspace_size(M, N, K // 2), dtype=torch.uint8, device="cuda:0")   # ... the unpack route both operands widened
    assert ws.numel() >= (M + N) * K
    try:
        f = lib.mixq_int4_fused_dequantize_silu if silu else lib.mixq_int4_fused_dequantize
        assert f(p(ap), p(bp), p(dev(sa)), p(dev(sb)), p(dev(y)), p(D), M, N, K // 2, p(ws), st) == 0
        torch.cuda.synchronize()
        assert b"s4" not in lib.mixq_debug_last_gemm_kernel()
    finally:
        lib.mixq_debug_set_gemm_variant(870)
    assert np.array_equal(bits(D.cpu().numpy()), bits(got))
    # round 5: the packed weight is read in 256-byte runs through a wave-private LDS tile (WROWS; K / 2 = 1040 and 5504 end in a
    # partial group, 128: half a group); knob 873 = the plain 64-byte fragment loads: same bits
    D1 = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    for knob in (873, 874):   # 874 = the 256-byte runs whatever the rule says (it takes them from 5 rows on 16 MiB and more)
        lib.mixq_debug_set_gemm_variant(knob)
        try:
            assert f(p(ap), p(bp), p(dev(sa)), p(dev(sb)), p(dev(y)), p(D1), M, N, K // 2, None, st) == 0
            torch.cuda.synchronize()
            assert b"gemm_skinny_s4_kernel" in lib.mixq_debug_last_gemm_kernel()
        finally:
            lib.mixq_debug_set_gemm_variant(872)
        assert np.array_equal(bits(D1.cpu().numpy()), bits(got)), knob
    # no addend, no workspace
    D2 = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    f = lib.mixq_int4_fused_dequantize_silu if silu else lib.mixq_int4_fused_dequantize
    assert f(p(ap), p(bp), p(dev(sa)), p(dev(sb)), None, p(D2), M, N, K // 2, None, st) == 0
    want0 = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, C=None, silu=silu)
    if not silu:
        assert np.array_equal(bits(D2.cpu().numpy()), bits(want0))


def test_int4_prefill_with_the_weight_widened_once(oracle):
    """M > 64: the int8 kernels; `mixq_int4_fused_dequantize_w8` takes the weight widened ONCE (mixlib.unpack_int4_to_int8) and
    widens only A per call: same bits as the route that widens both."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(5)
    M, N, K = 300, 1024, 4096
    a = rng.integers(-8, 8, (M, K), dtype=np.int8)
    b = rng.integers(-8, 8, (N, K), dtype=np.int8)
    sa = (rng.random(M) * 0.5 + 0.01).astype(np.float16)
    sb = (rng.random(N) * 1e-2 + 1e-4).astype(np.float16)
    y = (rng.standard_normal((M, N)) * 0.3).astype(np.float16)
    ap, bp = dev(oracle.pack_i4(a)), dev(oracle.pack_i4(b))
    b8 = mixlib.unpack_int4_to_int8(bp)
    assert np.array_equal(b8.cpu().numpy(), b)
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, C=y)
    for B_int8 in (None, b8):
        got = mixlib.int4FusedDequantize(ap, bp, dev(sa.reshape(M, 1)), dev(sb.reshape(1, N)), dev(y), M, N, K // 2, B_int8)
        assert np.array_equal(bits(got.cpu().numpy()), bits(want))


def test_mixlinear_4bit_decode_batch_and_prefill_agree_with_the_oracle(oracle):
    """MixLinear_GEMM(bit = 4).forward at a decode batch (16 rows: the packed weight stream) and at prefill size with
    prepare_prefill() (int8 copy of the weight): both against oracle.mixlinear4_forward."""
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(23)
    N, K, FP = 1024, 4096, 128
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    scales = np.abs(rng.standard_normal(K)).astype(np.float32)
    cache = mixlinear.MixLibCache(inputdim=512, sigma=6, bit=4, device="cuda:0")
    layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), bit=4, cache=cache, dev="cuda:0",
                                                 layer_scales=torch.from_numpy(scales), fp_features_num=FP)
    qp, sc, ind, wc = oracle.mixlinear4_from_linear(W, scales, FP)
    layer.prepare_prefill()
    assert layer.q_weight_i8 is not None and np.array_equal(layer.q_weight_i8.cpu().numpy(), oracle.unpack_i4(qp))
    for M in (16, 200):
        x = (rng.standard_normal((M, K)) * 0.8).astype(np.float16)
        x[:, ind] *= 12.0
        got = layer.forward(dev(x), cache, True).cpu().numpy()
        want = oracle.mixlinear4_forward(qp, sc, ind, wc, x.copy())
        g, w = got.astype(np.float64), want.astype(np.float64)
        assert np.abs(g - w).max() / np.abs(w).max() < 1e-3, M


@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("M,N,K", [(1, 512, 4096), (3, 256, 1024), (8, 1024, 4096), (16, 256, 2048), (5, 128, 2080), (2, 512, 11008), (1, 256, 28672),
                                   (9, 256, 11008), (40, 256, 1024)])
def test_int4_linear_forward_is_quant_then_gemm_in_one_call(oracle, M, N, K, silu):
    """mixq_int4_linear_forward: FindRowScale(bit = 4) + int4FusedDequantize[Silu] on the fp16 rows in ONE call -- for up to
    16 rows whose packed image fits LDS, ONE launch (the rows are quantised inside the weight-streaming kernel).  Bit for bit the two
    launches it replaces, x_scale included; rows with NaN / zeros / a huge element; against oracle.quant4_rows + the exact integer product."""
    from mixq_tensorrt_llm_amd import _lib, mixlib
    lib = _lib.load()
    rng = np.random.default_rng(M * 131 + K)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.05, 20)).astype(np.float16)
    x[0, 5] = np.float16(300.0)
    if M > 2:
        x[1] = 0
        x[2, 9] = np.nan
    b = rng.integers(-8, 8, (N, K), dtype=np.int8)
    bp = dev(oracle.pack_i4(b))
    sb = (rng.random(N) * 1e-2 + 1e-3).astype(np.float16)
    y = (rng.standard_normal((M, N)) * 0.5).astype(np.float16)
    xs = torch.zeros((max(M, 64), 1), dtype=torch.float16, device="cuda:0")
    fits = M <= 16 and M * (K // 2) <= 40 * 1024
    lib.mixq_debug_set_gemm_variant(877)     # the fused quantiser wherever the kernel serves the size (by default: one row only, the measured rule)
    try:
        got, q = mixlib.int4_linear_forward(dev(x), bp, dev(sb.reshape(1, N)), dev(y), xs, silu=silu)
        torch.cuda.synchronize()
        assert (b"<QF>" in lib.mixq_debug_last_gemm_kernel()) == fits, lib.mixq_debug_last_gemm_kernel()
    finally:
        lib.mixq_debug_set_gemm_variant(875)
    xs_d = torch.zeros_like(xs)
    got_d, q_d = mixlib.int4_linear_forward(dev(x), bp, dev(sb.reshape(1, N)), dev(y), xs_d, silu=silu)   # the default rule
    torch.cuda.synchronize()
    assert (b"<QF>" in lib.mixq_debug_last_gemm_kernel()) == (fits and M == 1)
    assert np.array_equal(bits(got_d.cpu().numpy()), bits(got.cpu().numpy())) and np.array_equal(bits(xs_d.cpu().numpy()), bits(xs.cpu().numpy()))
    # the two launches it replaces
    xs2 = torch.zeros_like(xs)
    q2 = mixlib.FindRowScale(dev(x), xs2, M, K, 4)
    fn = mixlib.int4FusedDequantizeSilu if silu else mixlib.int4FusedDequantize
    want2 = fn(q2, bp, xs2, dev(sb.reshape(1, N)), dev(y), M, N, K // 2)
    torch.cuda.synchronize()
    assert np.array_equal(bits(xs[:M].cpu().numpy()), bits(xs2[:M].cpu().numpy()))
    assert np.array_equal(bits(got.cpu().numpy()), bits(want2.cpu().numpy()))
    if not (fits and M == 1):
        assert torch.equal(q_d, q2)
    # the same with the fused quantiser switched off: two launches through q_packed, same bits
    lib.mixq_debug_set_gemm_variant(876)
    try:
        got3, _ = mixlib.int4_linear_forward(dev(x), bp, dev(sb.reshape(1, N)), dev(y), torch.zeros_like(xs), silu=silu)
        torch.cuda.synchronize()
        assert b"<QF>" not in lib.mixq_debug_last_gemm_kernel()
    finally:
        lib.mixq_debug_set_gemm_variant(875)
    assert np.array_equal(bits(got3.cpu().numpy()), bits(got.cpu().numpy()))
    # and the oracle
    qo, so = oracle.quant4_rows(x)
    assert np.array_equal(bits(xs[:M].cpu().numpy().reshape(-1)), bits(so))
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(oracle.unpack_i4(qo), b), so, sb, C=y, silu=silu)
    g, w = got.cpu().numpy().astype(np.float64), want.astype(np.float64)
    if silu:
        ok = np.isfinite(w)
        assert np.abs(g[ok] - w[ok]).max() / max(np.abs(w[ok]).max(), 1e-30) < 1e-3
    else:
        assert np.array_equal(bits(got.cpu().numpy()), bits(want))

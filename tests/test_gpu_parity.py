"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): int8 quantisation, per-token scales, outlier gather and int32 accumulators are
BIT-EXACT; the dequant epilogue is bit-exact once its addend is given; the complete operator (whose fp16 side GEMM has
an unspecified accumulation order in the reference too -- cuBLAS) is within 1e-3 relative.
"""
import ctypes

import numpy as np
import pytest

from conftest import assert_elementwise, assert_prefill_parity, make_layer, prefill_slack, ulp_histogram, w8a16_slack

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch.device("cuda:0")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.argwhere(g != w)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {g.size} fp16 values differ; first at {i}: got {got[i]!r} "
                             f"(0x{int(g[i]):04x}) want {want[i]!r} (0x{int(w[i]):04x}); max |diff| "
                             f"{np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64)))}")


REL_TOL = 1e-3  # north_star: fp16 output within 1e-3 relative (normalised by the output's max magnitude)


def rel_err(got, want):
    g, w = got.astype(np.float64), want.astype(np.float64)
    return np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)


# ------------------------------------------------------------------------------- quant + extract -------
@pytest.mark.parametrize("M,K", [(1, 256), (5, 512), (37, 1024), (64, 3584), (33, 4096), (17, 8192), (9, 11008),
                                 (6, 18944), (5, 28672), (3, 40960)])
def test_quant_rows_bit_exact(oracle, M, K):
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(K + M)
    A = (rng.standard_normal((M, K)) * rng.uniform(0.01, 30)).astype(np.float16)
    if M > 2:
        A[1] = 0                       # zero row -> scale 0, q 0
        A[2, K // 3] = np.float16(6.5e4)
    if M > 4:
        A[3, 5] = np.nan
        A[4, :] = np.float16(6e-8)     # subnormal amax -> scale 0 -> +-inf quotient
    x = to_dev(A)
    s = torch.empty(M, dtype=torch.float16, device=dev())
    q = mixlib.FindRowScale(x, s, M, K, 8)
    qo, so = oracle.quant_rows(A)
    assert np.array_equal(bits(s.cpu().numpy()), bits(so))
    assert np.array_equal(q.cpu().numpy(), qo)
    assert np.array_equal(bits(x.cpu().numpy()), bits(A)), "T-flavour quantiser must not modify A"
    # Int8quantize with the same scales reproduces the rows (cult.cu:1732-1771)
    q2 = mixlib.Int8quantize(x, s)
    assert np.array_equal(q2.cpu().numpy(), qo)


def test_quant_every_fp16_value_against_many_scales(oracle):
    """The quantiser multiplies by a per-row reciprocal and falls back to an exact division only near fp16 rounding
    breakpoints (mixq_device.h: quant_one_fast).  Exhaustive check of that shortcut: every finite fp16 magnitude as x,
    both signs, against 48 different row maxima (-> 48 scales), bit-exact against the division-based oracle."""
    from mixq_tensorrt_llm_amd import mixlib
    allpos = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16)          # every finite non-negative fp16
    rng = np.random.default_rng(42)
    caps = np.concatenate([np.float16([1e-7, 6.1e-5, 1e-3, 0.0999, 1.0, 127.0, 254.0, 333.3, 1000.0, 65504.0]),
                           np.exp(rng.uniform(np.log(1e-4), np.log(6e4), 38)).astype(np.float16)])
    K = 32768
    rows = []
    for c in caps:
        v = allpos[allpos <= c]
        for sign in (1, -1):
            r = np.zeros(K, np.float16)
            r[: v.size] = v * np.float16(sign)
            r[v.size:] = np.resize(v, K - v.size) if v.size else 0      # pad with repeats: keeps amax == c'
            rows.append(r)
    A = np.stack(rows)
    x = to_dev(A)
    s = torch.empty(A.shape[0], dtype=torch.float16, device=dev())
    q = mixlib.FindRowScale(x, s, A.shape[0], K, 8).cpu().numpy()
    qo, so = oracle.quant_rows(A)
    assert np.array_equal(bits(s.cpu().numpy()), bits(so))
    bad = np.argwhere(q != qo)
    assert bad.size == 0, f"{len(bad)} mismatches, first {bad[:3].tolist()}"
    # the wave-per-row kernels (K <= 8192) use the same shortcut: random rows with many scales
    B = (rng.standard_normal((512, 4096)) * np.exp(rng.uniform(-6, 6, (512, 1)))).astype(np.float16)
    xb = to_dev(B)
    sb = torch.empty(512, dtype=torch.float16, device=dev())
    qb = mixlib.FindRowScale(xb, sb, 512, 4096, 8).cpu().numpy()
    qbo, sbo = oracle.quant_rows(B)
    assert np.array_equal(qb, qbo) and np.array_equal(bits(sb.cpu().numpy()), bits(sbo))


@pytest.mark.parametrize("M,K,O", [(7, 512, 128), (40, 4096, 128), (12, 11008, 128), (5, 28672, 128), (9, 1024, 8)])
def test_fused_quant_extract_both_flavours(oracle, M, K, O):
    from mixq_tensorrt_llm_amd import _lib, mixlib
    rng = np.random.default_rng(M * K)
    A = rng.standard_normal((M, K)).astype(np.float16)
    ind = rng.permutation(K)[:O].astype(np.int32)
    A[:, ind[: O // 2]] *= np.float16(20)
    lib = _lib.load()
    # T-flavour (enqueue): gather + quant, A untouched, amax includes the outliers
    x = to_dev(A)
    q = torch.empty((M, K), dtype=torch.int8, device=dev())
    s = torch.empty(M, dtype=torch.float16, device=dev())
    f = torch.empty((M, O), dtype=torch.float16, device=dev())
    i_d = to_dev(ind)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mixq_quant_extract(M, K, x.data_ptr(), q.data_ptr(), s.data_ptr(), f.data_ptr(), i_d.data_ptr(), O, 0,
                                  st) == 0
    qo, so = oracle.quant_rows(A)
    assert np.array_equal(q.cpu().numpy(), qo) and np.array_equal(bits(s.cpu().numpy()), bits(so))
    assert np.array_equal(bits(f.cpu().numpy()), bits(A[:, ind]))
    assert np.array_equal(bits(x.cpu().numpy()), bits(A))
    # stand-alone gather API (kernel/i8gemm.cu:226-244)
    f2 = torch.empty_like(f)
    assert lib.mixq_extract_outliers(M, K, x.data_ptr(), f2.data_ptr(), i_d.data_ptr(), O, st) == 0
    assert torch.equal(f, f2)
    # P-flavour (mixlib): outliers zeroed before amax/quant, A mutated
    x2 = to_dev(A)
    s2 = torch.empty(M, dtype=torch.float16, device=dev())
    q2, out2 = mixlib.FindRowScaleFusedExtracOutliers(x2, s2, i_d, O, M, K)
    Az = A.copy()
    fo = oracle.extract_outliers(Az, ind, set_zero=True)
    qz, sz = oracle.quant_rows(Az)
    assert np.array_equal(bits(out2.cpu().numpy()), bits(fo))
    assert np.array_equal(q2.cpu().numpy(), qz) and np.array_equal(bits(s2.cpu().numpy()), bits(sz))
    assert np.array_equal(bits(x2.cpu().numpy()), bits(Az))
    # unfused mixlib pair gives the same thing
    x3 = to_dev(A)
    out3 = mixlib.ExtractOutliersAndSetToZeros(i_d, x3)
    s3 = torch.empty(M, dtype=torch.float16, device=dev())
    q3 = mixlib.FindRowScale(x3, s3, M, K, 8)
    assert torch.equal(out3, out2) and torch.equal(q3, q2) and torch.equal(s3, s2)


# ----------------------------------------------------------------------------------- int8 GEMM ---------
GEMM_SHAPES = [(5, 16, 16), (32, 128, 128), (33, 144, 272), (64, 256, 512), (100, 400, 1040), (129, 256, 384),
               (256, 512, 1024), (300, 768, 640), (513, 1280, 896), (32, 4096, 4096)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_int32_accumulators_bit_exact(oracle, M, N, K):
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    # asymmetric operands: a transposed or permuted output cannot pass
    got = mixlib.gemm(to_dev(a), to_dev(b), M, N, K).cpu().numpy()
    want = oracle.gemm_s8s8s32(a, b) if M * N * K < 3e8 else (a.astype(np.int32) @ b.astype(np.int32).T)
    assert np.array_equal(got, want)


@pytest.fixture
def variant():
    """Force a main-loop schedule (1 = 2-barrier kernel, 2 = 256x256 ping-pong kernel, 5 = 128x256 ping-pong kernel); auto
    again afterwards."""
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib.mixq_debug_set_gemm_variant
    lib.mixq_debug_set_gemm_variant(0)


@pytest.mark.parametrize("which", [1, 2, 5])
@pytest.mark.parametrize("M,N,K", [(5, 16, 16), (33, 144, 272), (129, 256, 384), (256, 512, 128), (300, 768, 640),
                                   (513, 1280, 896), (700, 528, 2064), (1024, 1024, 4096)])
def test_every_schedule_gives_identical_int32(oracle, variant, which, M, N, K):
    """Ragged M, N not a multiple of the tile, K not a multiple of the 128-byte slice, odd and even slice counts."""
    from mixq_tensorrt_llm_amd import mixlib
    variant(which)
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    got = mixlib.gemm(to_dev(a), to_dev(b), M, N, K).cpu().numpy()
    want = a.astype(np.int32) @ b.astype(np.int32).T
    assert np.array_equal(got, want)


@pytest.mark.parametrize("cfg", range(24))
def test_every_tile_configuration_of_the_two_barrier_kernel(oracle, variant, cfg):
    """variant 10 + cfg forces one tile / pipeline-depth configuration of gemm_w8a8o16_kernel (2..8 LDS stages): raw
    int32 on ragged shapes (K shorter than the prefetch depth, K tail, odd slice counts) + the fused operator."""
    from mixq_tensorrt_llm_amd import mixlib
    variant(10 + cfg)
    for (M, N, K) in [(70, 144, 48), (129, 272, 400), (200, 336, 1040), (515, 528, 2064)]:
        rng = np.random.default_rng(M + N + K + cfg)
        a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
        b = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
        got = mixlib.gemm(to_dev(a), to_dev(b), M, N, K).cpu().numpy()
        assert np.array_equal(got, a.astype(np.int32) @ b.astype(np.int32).T), (M, N, K)
    A, W, act = make_layer(150, 272, 704, seed=cfg)
    p = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, p)
    assert_prefill_parity(oracle, got, A, p, f"tile configuration {cfg}")


@pytest.mark.parametrize("M,N,K", [(300, 768, 1280), (700, 528, 2112), (520, 1024, 512), (257, 272, 704)])
@pytest.mark.parametrize("which", [1, 2, 5])
def test_every_schedule_full_operator(oracle, variant, which, M, N, K):
    """Both schedules on ragged M/N, a partial last K slice, odd slice counts."""
    variant(which)
    A, W, act = make_layer(M, N, K, seed=31)
    p = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, p)
    assert_prefill_parity(oracle, got, A, p, f"schedule {which} {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(300, 11008, 1024), (513, 5008, 448), (384, 12288, 512), (1500, 4096, 640)])
def test_mid_size_shapes_take_the_128x256_tiles_and_match_the_oracle(oracle, M, N, K):
    """88..256 tiles of 128 x 256 at short K: launch_gemm picks gemm_w8a8o16_pp128_kernel by itself (ragged M / N, a partial
    last K slice, one and several K slices); same bits as the two-barrier kernel, 1e-3 from the oracle."""
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    lib.mixq_debug_set_gemm_variant(0)
    A, W, act = make_layer(M, N, K, seed=M + K)
    p = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, p)
    assert b"pp128" in lib.mixq_debug_last_gemm_kernel(), lib.mixq_debug_last_gemm_kernel()
    assert_prefill_parity(oracle, got, A, p, f"128x256 tiles {M}x{N}x{K}")
    lib.mixq_debug_set_gemm_variant(1)
    try:
        two_barrier = run_enqueue(A, p)
    finally:
        lib.mixq_debug_set_gemm_variant(0)
    assert np.array_equal(bits(got), bits(two_barrier))


@pytest.mark.parametrize("epi", ["dequant", "dequant+y", "silu+y", "silu_mul"])
@pytest.mark.parametrize("M,N,K", [(300, 528, 2064), (129, 272, 384)])
def test_every_epilogue_on_the_128x256_tiles(variant, epi, M, N, K):
    """The P-flavour epilogues (addend y, SiLU, SiLU * up) through gemm_w8a8o16_pp128_kernel (variant 5) on ragged shapes
    with a partial last K slice: the same bits as the two-barrier kernel (variant 1)."""
    from mixq_tensorrt_llm_amd import mixlib
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, generator=g).to(dev())
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g).to(dev())
    sa = (torch.rand(M, generator=g) * 0.05 + 0.01).to(torch.float16).to(dev())
    sb = (torch.rand(N, generator=g) * 4e-4 + 1e-4).to(torch.float16).to(dev())
    y = torch.randn((M, N), generator=g).to(torch.float16).to(dev()) if "+y" in epi or epi == "silu_mul" else None
    up = torch.randn((M, N), generator=g).to(torch.float16).to(dev())

    def run():
        if epi.startswith("dequant"):
            return mixlib.int8FusedDequantize(a, b, sa, sb, y, M, N, K)
        if epi == "silu+y":
            return mixlib.int8FusedDequantizeSilu(a, b, sa, sb, y, M, N, K)
        return mixlib.int8FusedDequantizeSiluMul(a, b, sa, sb, y, up, M, N, K)
    variant(1)
    ref = run()
    variant(5)
    got = run()
    assert torch.equal(got, ref)


def test_schedules_agree_bitwise_on_the_full_operator(variant):
    from mixq_tensorrt_llm_amd import mixlib
    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 1100, 1536, 2176
    A = torch.randn((M, K), generator=g).to(torch.float16).to(dev())
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g).to(dev())
    sW = (torch.rand(N, generator=g) * 1e-3 + 1e-4).to(torch.float16).to(dev())
    fpw = (torch.randn((N, 128), generator=g) * 0.02).to(torch.float16).to(dev())
    ind = torch.randperm(K, generator=g)[:128].to(torch.int32).to(dev())
    outs = []
    for which in (1, 2, 5):
        variant(which)
        outs.append(mixlib.mixq_linear(A, W, sW, fpw, ind))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # more tiles than CUs, odd slice count
    M2, N2, K2 = 4200, 5136, 1152 + 128
    A2 = torch.randn((M2, K2), generator=g).to(torch.float16).to(dev())
    W2 = torch.randint(-127, 128, (N2, K2), dtype=torch.int8, generator=g).to(dev())
    sW2 = (torch.rand(N2, generator=g) * 1e-3 + 1e-4).to(torch.float16).to(dev())
    fpw2 = (torch.randn((N2, 128), generator=g) * 0.02).to(torch.float16).to(dev())
    ind2 = torch.randperm(K2, generator=g)[:128].to(torch.int32).to(dev())
    outs = []
    for which in (1, 2, 5):
        variant(which)
        outs.append(mixlib.mixq_linear(A2, W2, sW2, fpw2, ind2))
    assert torch.equal(outs[0], outs[1])


def test_int32_extremes_do_not_saturate(oracle):
    from mixq_tensorrt_llm_amd import mixlib
    M, N, K = 32, 128, 28672
    a = np.full((M, K), -128, np.int8)
    b = np.full((N, K), -128, np.int8)
    b[1::2] = 127
    got = mixlib.gemm(to_dev(a), to_dev(b), M, N, K).cpu().numpy()
    assert got[0, 0] == 128 * 128 * K and got[0, 1] == -128 * 127 * K


@pytest.mark.parametrize("M,N,K", [(7, 16, 32), (37, 256, 512), (130, 272, 1040), (300, 512, 256)])
@pytest.mark.parametrize("silu", [False, True])
def test_fused_dequant_epilogue(oracle, M, N, K, silu):
    """int8FusedDequantize(A,B,scale_row,scale_col,y): exact integer GEMM + fp32 FMA + one RNE -> bit-exact;
    the SiLU variant goes through expf, whose device implementation may differ by an ulp -> 1e-3."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(K)
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-127, 128, size=(N, K), dtype=np.int8)
    sa = (np.abs(rng.standard_normal(M)) * 0.05 + 1e-3).astype(np.float16)
    sb = (np.abs(rng.standard_normal(N)) * 1e-3 + 1e-5).astype(np.float16)
    y = rng.standard_normal((M, N)).astype(np.float16)
    fn = mixlib.int8FusedDequantizeSilu if silu else mixlib.int8FusedDequantize
    got = fn(to_dev(a), to_dev(b), to_dev(sa), to_dev(sb), to_dev(y), M, N, K).cpu().numpy()
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, y, silu=silu)
    if silu:
        # accurate expf + IEEE division on both sides (the product no longer uses the hardware v_exp_f32 shortcut); the
        # two libm's may still differ in the last ulp of expf, which survives the fp16 rounding only next to a breakpoint:
        # at most one fp16 ulp, in well under 0.1 % of the elements
        assert rel_err(got, want) < REL_TOL
        gb, wb = bits(got).astype(np.int32), bits(want).astype(np.int32)
        assert np.abs(gb - wb).max() <= 1, "SiLU epilogue: more than one fp16 ulp from the expf oracle"
        assert (gb != wb).mean() < 1e-3, f"SiLU epilogue: {(gb != wb).mean():.2e} of the outputs differ from the expf oracle"
    else:
        assert_bits_equal(got, want, "int8FusedDequantize")


def test_unfused_pair_matches_reference_rounding(oracle):
    """mixlib.gemm + dequantizeInt8 (P-flavour sm90 route, linear.py:231-238): two roundings, bit-exact."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(1)
    M, N, K = 45, 96, 256
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-127, 128, size=(N, K), dtype=np.int8)
    sa = (np.abs(rng.standard_normal(M)) * 0.05).astype(np.float16)
    sb = (np.abs(rng.standard_normal(N)) * 1e-3).astype(np.float16)
    y = rng.standard_normal((M, N)).astype(np.float16)
    acc = mixlib.gemm(to_dev(a), to_dev(b), M, N, K)
    got = mixlib.dequantizeInt8(acc, to_dev(sa), to_dev(sb), to_dev(y), 8, M, N).cpu().numpy()
    want = oracle.dequantization(oracle.gemm_s8s8s32(a, b), sa, sb, y)
    assert_bits_equal(got, want, "gemm + dequantizeInt8")


def test_unfused_silu_twin(oracle):
    """mixlib.gemm + dequantizeInt8Silu (cult.cu:2341-2348, P-flavour sm90 route linear.py:321-324): fp32 math, one
    rounding; equal to the oracle's expf form up to the last ulp of expf (<= 1 fp16 ulp, < 0.1 % of the elements)."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(11)
    M, N, K = 45, 96, 256
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-127, 128, size=(N, K), dtype=np.int8)
    sa = (np.abs(rng.standard_normal(M)) * 0.05).astype(np.float16)
    sb = (np.abs(rng.standard_normal(N)) * 1e-3).astype(np.float16)
    y = rng.standard_normal((M, N)).astype(np.float16)
    acc = mixlib.gemm(to_dev(a), to_dev(b), M, N, K)
    got = mixlib.dequantizeInt8Silu(acc, to_dev(sa), to_dev(sb), to_dev(y), 8, M, N).cpu().numpy()
    want = oracle.dequantization_silu(oracle.gemm_s8s8s32(a, b), sa, sb, y)
    gb, wb = bits(got).astype(np.int32), bits(want).astype(np.int32)
    assert np.abs(gb - wb).max() <= 1 and (gb != wb).mean() < 1e-3
    # and it is NOT the fused epilogue's rounding (that one multiplies the two scales first): keep them apart
    fused = oracle.dequant_epilogue(oracle.gemm_s8s8s32(a, b), sa, sb, y, silu=True)
    assert rel_err(got, fused) < REL_TOL


def test_fp16_side_gemm(oracle):
    from mixq_tensorrt_llm_amd import _lib
    rng = np.random.default_rng(2)
    M, N, O = 70, 208, 128
    fa = (rng.standard_normal((M, O)) * 10).astype(np.float16)
    fw = (rng.standard_normal((N, O)) * 0.02).astype(np.float16)
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    a_d, w_d = to_dev(fa), to_dev(fw)
    rc = _lib.load().mixq_gemm_fp16(a_d.data_ptr(), w_d.data_ptr(), out.data_ptr(), M, N, O,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert rel_err(out.cpu().numpy(), oracle.gemm_fp16(fa, fw)) < REL_TOL


# --------------------------------------------------------------------------- the whole operator --------
LINEAR_SHAPES = [(5, 128, 256), (32, 256, 512), (37, 256, 512), (100, 512, 1024), (129, 400, 640), (256, 1024, 2048),
                 (300, 768, 1280), (520, 512, 3584)]


def run_enqueue(A, p, lead=None):
    from mixq_tensorrt_llm_amd import plugin
    M, K = A.shape
    N = p["weight"].shape[0]
    layer = plugin.MixQLinear(K, N, device=dev()).load(p)
    x = to_dev(A)
    if lead is not None:
        x = x.reshape(*lead, K)
    out = layer(x)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(M, N)


@pytest.mark.parametrize("M,N,K", LINEAR_SHAPES)
def test_enqueue_prefill_matches_oracle(oracle, M, N, K):
    A, W, act = make_layer(M, N, K, seed=M + N)
    p = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, p)
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    assert rel_err(got, want) < REL_TOL
    # and the answer is the operator's, not just self-consistent: close to the unquantised product
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    assert rel_err(got, ref.astype(np.float32)) < 0.05
    mism = np.mean(bits(got) != bits(want))
    assert mism < 0.02, f"{mism:.4f} of outputs differ from the oracle by >= 1 fp16 ulp"
    # element by element: one fp16 rounding step + one ulp of the outlier product (the only order-dependent quantity)
    assert_elementwise(got, want, prefill_slack(parts, A, p), f"enqueue {M}x{N}x{K}")
    h = ulp_histogram(got, want)
    assert h["<=1"] > 0.999 and h[">2"] < 1e-4, h


def test_enqueue_gemm_stage_is_bit_exact_given_oracle_side_product(oracle):
    """Feeding the oracle's fp16 side product P as the addend removes the only order-dependent step: the remaining
    pipeline (quant -> int8 GEMM -> dequant FMA) must then agree with the oracle bit for bit."""
    from mixq_tensorrt_llm_amd import mixlib
    M, N, K = 150, 384, 1024
    A, W, act = make_layer(M, N, K, seed=77)
    p = oracle.pack_linear_weights(W, act)
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    x = to_dev(A)
    s = torch.empty(M, dtype=torch.float16, device=dev())
    q = mixlib.FindRowScale(x, s, M, K, 8)
    got = mixlib.int8FusedDequantize(q, to_dev(p["weight"]), s, to_dev(p["weights_scaling_factor"]),
                                     to_dev(parts["P"]), M, N, K).cpu().numpy()
    assert_bits_equal(got, want, "quant -> int8 GEMM -> dequant FMA")


def test_enqueue_leading_dims_and_registry_path(oracle):
    from mixq_tensorrt_llm_amd import plugin
    A, W, act = make_layer(48, 256, 512, seed=4)
    p = oracle.pack_linear_weights(W, act)
    got3d = run_enqueue(A, p, lead=(4, 12))
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    assert rel_err(got3d, want) < REL_TOL
    layer = plugin.MixQLinear(512, 256, device=dev()).load(p)
    out = plugin.mixgemm(48, 256, 512, [to_dev(A), layer.weight, layer.weights_scaling_factor, layer.fp_weight,
                                        layer.fp_ind, layer.qweight, layer.weights_scaling_factor])
    assert rel_err(out.cpu().numpy(), want) < REL_TOL


def test_enqueue_edge_rows(oracle):
    """Zero rows, a NaN, an all-outlier row, M = 5 (smallest prefill), ragged M."""
    A, W, act = make_layer(5, 128, 256, seed=8)
    p = oracle.pack_linear_weights(W, act)
    A[1] = 0
    A[2, :] = 0
    A[2, p["fp_ind"]] = np.float16(3.0)   # only outlier columns are non-zero
    got = run_enqueue(A, p)
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    with np.errstate(invalid="ignore"):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        assert np.abs(got[ok].astype(np.float64) - want[ok].astype(np.float64)).max() <= \
            REL_TOL * np.abs(want[ok].astype(np.float64)).max()


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(128, 256), (512, 1024), (384, 4096)])
def test_enqueue_decode_path(oracle, M, N, K):
    """M <= 4 routes to the W8A16 path on the EETQ-interleaved qweight (TsinghuaMixQPlugin.cpp:641-647).
    Two oracles: `w8a16_gemv` (fp16-rounded weights, fp32 sums -- what the HIP kernel computes, and what the
    reference's tensor-core route for M > 4 computes) and `w8a16_gemv_reforder` (the CUDA GEMV's own order: fp16 FMA
    chains per thread, kernel.h:425-470).  The product must be within the north-star 1e-3 of the first, and no farther
    from the reference-order result than that result is from exact arithmetic (+ 1e-3)."""
    A, W, act = make_layer(M, N, K, seed=N + M, outlier_gain=1.0)
    p = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, p)
    q_un = oracle.eetq_symmetric_quantize(W.T.copy())[0]
    want = oracle.w8a16_gemv(A, q_un, p["weights_scaling_factor"])   # plugin reuses max/127 scales
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, w8a16_slack(A, q_un, p["weights_scaling_factor"]), f"decode {M}x{N}x{K}")
    ref_order = oracle.w8a16_gemv_reforder(A, q_un, p["weights_scaling_factor"])
    exact = A.astype(np.float64) @ (q_un.astype(np.float64) * p["weights_scaling_factor"].astype(np.float64))
    ref_vs_exact = rel_err(ref_order, exact.astype(np.float32))
    assert rel_err(got, ref_order) < ref_vs_exact + REL_TOL
    assert rel_err(got, exact.astype(np.float32)) <= ref_vs_exact + 2e-4   # at least as close to exact as the reference order


@pytest.mark.parametrize("route", [856, 857, 858])
@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(384, 1024), (8200, 512)])
def test_enqueue_decode_batches_both_routes(oracle, variant, route, M, N, K):
    """Decode, 1..4 tokens: the GEMV (857), the MFMA skinny form (856: decode_kernels.hip's per-token cost made
    it slower than the 5-token kernel on wide outputs) and the automatic choice (858; N >= 8192 takes the skinny form),
    all against the same two oracles and bounds as test_enqueue_decode_path."""
    A, W, act = make_layer(M, N, K, seed=N + M + route, outlier_gain=1.0)
    p = oracle.pack_linear_weights(W, act)
    variant(route)
    try:
        got = run_enqueue(A, p)
    finally:
        variant(858)
    q_un = oracle.eetq_symmetric_quantize(W.T.copy())[0]
    want = oracle.w8a16_gemv(A, q_un, p["weights_scaling_factor"])
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, w8a16_slack(A, q_un, p["weights_scaling_factor"]), f"decode route {route} {M}x{N}x{K}")
    ref_order = oracle.w8a16_gemv_reforder(A, q_un, p["weights_scaling_factor"])
    exact = A.astype(np.float64) @ (q_un.astype(np.float64) * p["weights_scaling_factor"].astype(np.float64))
    ref_vs_exact = rel_err(ref_order, exact.astype(np.float32))
    assert rel_err(got, ref_order) < ref_vs_exact + REL_TOL
    assert rel_err(got, exact.astype(np.float32)) <= ref_vs_exact + 2e-4


def test_tp_shards_compose(oracle):
    """Row-sharded W on one GPU: concatenating the shards' outputs equals the unsharded operator bit for bit."""
    from mixq_tensorrt_llm_amd import pack, parallel
    A, W, act = make_layer(70, 512, 512, seed=21)
    full = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    whole = run_enqueue(A, full)
    parts = [run_enqueue(A, parallel.shard_packed(full, 4, r)) for r in range(4)]
    assert np.array_equal(bits(np.concatenate(parts, axis=1)), bits(whole))


# --------------------------------------------------- full-size properties (no oracle at these sizes) ---
def test_full_size_linearity_and_row_independence():
    """Llama-2-7B qkv shape, M = 2048: (i) int32 GEMM is linear in W: acc(W1) + acc(W2) == acc(W1 + W2) when no int8
    overflow; (ii) quantisation is per-row: permuting rows of A permutes rows of Out bit-exactly."""
    from mixq_tensorrt_llm_amd import mixlib
    g = torch.Generator(device="cpu").manual_seed(0)
    M, N, K = 2048, 12288, 4096
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(dev())
    w1 = torch.randint(-60, 60, (N, K), dtype=torch.int8, generator=g).to(dev())
    w2 = torch.randint(-60, 60, (N, K), dtype=torch.int8, generator=g).to(dev())
    acc = mixlib.gemm(a, w1, M, N, K) + mixlib.gemm(a, w2, M, N, K)
    assert torch.equal(acc, mixlib.gemm(a, (w1 + w2), M, N, K))
    # checksum against an independent formulation: column sums
    colsum = (a.to(torch.float64).sum(dim=0, keepdim=True) @ w1.to(torch.float64).t()).to(torch.int64)
    assert torch.equal(mixlib.gemm(a, w1, M, N, K).to(torch.int64).sum(dim=0, keepdim=True), colsum)
    A = torch.randn((M, K), generator=g).to(torch.float16).to(dev())
    sW = (torch.rand(N, generator=g) * 1e-3 + 1e-4).to(torch.float16).to(dev())
    fpw = (torch.randn((N, 128), generator=g) * 0.02).to(torch.float16).to(dev())
    ind = torch.randperm(K, generator=g)[:128].to(torch.int32).to(dev())
    out = mixlib.mixq_linear(A, w1, sW, fpw, ind)
    perm = torch.randperm(M, generator=g).to(dev())
    out_p = mixlib.mixq_linear(A[perm].contiguous(), w1, sW, fpw, ind)
    assert torch.equal(out[perm], out_p)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("M,N,K", [(3, 1024, 2048), (48, 1024, 2048), (700, 1024, 2048), (24, 4096, 2304), (57, 6144, 1280)])
def test_enqueue_is_capturable_in_a_hip_graph(oracle, M, N, K):
    """The operator is a fixed sequence of launches on the caller's stream (no allocation, no host sync): after one warm
    call it can be captured into a HIP graph and replayed on new data in the same buffers -- decode (M <= 4), the
    split-K tile kernel, the ping-pong kernel, and (round 5) the decode-batch GEMM on its 256-byte-run weight route with one and
    with two feature tiles per workgroup."""
    from mixq_tensorrt_llm_amd import plugin
    A, W, act = make_layer(M, N, K, seed=77 + M)
    p = oracle.pack_linear_weights(W, act)
    layer = plugin.MixQLinear(K, N, device=dev()).load(p)
    x = to_dev(np.zeros_like(A))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = layer(x)                           # warm-up: one-time function attributes, workspace allocation
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = layer(x)
    torch.cuda.current_stream().wait_stream(side)
    for trial in range(2):
        A2 = np.ascontiguousarray(np.roll(A, trial * 3, axis=0))
        x.copy_(to_dev(A2))
        graph.replay()
        torch.cuda.synchronize()
        eager = run_enqueue(A2, p)
        assert np.array_equal(bits(out.cpu().numpy().reshape(M, N)), bits(eager)), "graph replay == eager launch"


@pytest.mark.parametrize("M,N,K", [(5, 4096, 4096), (16, 4096, 4096), (17, 4096, 4096), (32, 4096, 4096), (31, 3584, 3584),
                                   (32, 512, 2064), (9, 12288, 4096), (24, 1024, 8192), (32, 4096, 1088), (32, 5120, 5120),
                                   (20, 5136, 2048), (32, 12288, 4096), (32, 18944, 3584), (33, 4096, 4096), (48, 8192, 4096),
                                   (64, 4096, 4096), (57, 3584, 3584), (64, 12288, 4096), (32, 4096, 11008), (48, 4096, 11008),
                                   (32, 3584, 18944)])
def test_enqueue_decode_batches_fragment_major_qa(oracle, variant, M, N, K):
    """Round 3: for decode batches that the weight-streaming skinny GEMM serves, mixq_enqueue's quantiser writes qA in that
    kernel's MFMA fragment order (one contiguous 1-KiB read per fragment load instead of 16 rows x 64 B).  The operator must
    give the SAME BITS as with the row-major image (knob 890) and match the oracle element by element -- whole and ragged 16-row
    tiles, K % 64 != 0 (2064: the last k-step is partial), 32 features per workgroup (N = 5120 / 5136: a little more than one
    round of 16-column workgroups; 5136 leaves the last workgroup half empty), the widest output the skinny kernel takes at 32 rows
    (12288) and one beyond it (18944: two-barrier tiles), shapes outside the layout's domain (K <= 1024, K = 8192 with scratch:
    row-major either way)."""
    A, W, act = make_layer(M, N, K, seed=3 * M + N + K)
    if K % 64 == 0:
        p = oracle.pack_linear_weights(W, act)
    else:   # (the decode-path `qweight` image needs whole 64-row tiles; the prefill path does not read it)
        sW = oracle.weight_scales(W)
        ind = oracle.select_outliers(act, 128)
        p = dict(weight=oracle.quantize_weight(W, sW, ind), weights_scaling_factor=sW,
                 fp_weight=np.ascontiguousarray(W[:, ind]), fp_ind=ind, qweight=np.zeros((K, N), np.uint8))
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                        return_parts=True)
    variant(891)
    got = run_enqueue(A, p)
    variant(890)
    try:
        ref = run_enqueue(A, p)
    finally:
        variant(891)
    assert_bits_equal(got, ref, f"fragment-major vs row-major qA {M}x{N}x{K}")
    assert rel_err(got, want) < REL_TOL
    assert_elementwise(got, want, prefill_slack(parts, A, p), f"decode batch {M}x{N}x{K}")

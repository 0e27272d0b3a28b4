"""GPU (-m gpu): the mid-M "deep" form of the fused GEMM (csrc/gemm_kernels.hip, ADMA: 128 x 128 tiles, asm-form LDS-DMA copies with
four / five LDS stages really in flight, K split over 1 / 2 / 4 / 8 workgroups per tile) -- 129..512-row calls whose other forms leave
CUs idle.  Must give the SAME BITS as the one-workgroup-per-tile kernels: every build x split, ragged shapes, a partial last K slice,
every epilogue, repeated launches on one scratch, the automatic table on the BASELINE shapes it was fitted on, through mixq_enqueue
against the oracle, and inside a HIP graph."""
import ctypes

import numpy as np
import pytest

from conftest import assert_prefill_parity, make_layer
from test_gpu_splitk import operands, p

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

OFF, AUTO = 1241, 1240


@pytest.fixture
def lib():
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib
    lib.mixq_debug_reset()


SHAPES = [(300, 528, 2064),     # ragged M and N, partial last K slice, 3 x 5 tiles
          (130, 1040, 4224),    # 2 rows into the second tile row
          (512, 1024, 8192),    # whole tiles
          (200, 784, 4352)]


@pytest.mark.parametrize("build", [0, 10, 20, 30, 31, 32])   # 4 waves x 64 x 64 | 8 waves x 64 x 32 | 8 waves, 5 stages | round 6: copy-only waves, one barrier per pair of slices (gemm_mid_kernels.hip; 30: every tile row starting at its own K slice -- a measurement option --, 31: all from slice 0 = the default, 32: 128-wide tiles always -- by rule the unsplit cases here take 96-wide ones; K % 128 != 0 falls back to build 10)
@pytest.mark.parametrize("xs", [1, 2, 4, 8])
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("O", [128, 0, 40, 256])   # 256: side GEMM first, then the addend form (C = D = Out)
def test_deep_form_gives_the_bits_of_the_one_workgroup_form(lib, build, xs, M, N, K, O):
    if build == 0 and O in (40, 256) or build == 20 and O != 128:
        pytest.skip("a subset for the builds the table does not select")
    if build >= 30:
        lib.mixq_debug_set_gemm_variant(1431 if build == 32 else 1412 - (build - 30))
        build = 30
    if xs == 8:
        K = 2 * K + 16          # 8 ways need >= 32 K slices
    if build == 30 and (M, N) != (300, 528):
        K = (K + 127) // 128 * 128 + (128 if xs == 2 else 0)   # whole slices (odd and even counts); the ragged shape keeps its partial slice = the fall-back
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M + N + K + O)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fa, fw = (p(fpA), p(fpW)) if O else (None, None)
    lib.mixq_debug_set_gemm_variant(OFF)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), fa, fw, p(ref), M, N, K, O, st) == 0
    assert b"DEEP" not in lib.mixq_debug_last_gemm_kernel()
    lib.mixq_debug_set_gemm_variant(OFF + build + xs)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    scr = torch.zeros(max(n, 16), dtype=torch.uint8, device="cuda:0")
    for round_ in range(3):   # the counter of every tile is left zero by the last workgroup
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), fa, fw, p(out), M, N, K, O, p(scr), n, st) == 0
        torch.cuda.synchronize()
        assert b"DEEP" in lib.mixq_debug_last_gemm_kernel(), lib.mixq_debug_last_gemm_kernel()
        assert torch.equal(out, ref), f"round {round_}"
    assert int(scr[:16384].to(torch.int32).sum()) == 0


@pytest.mark.parametrize("xs", [1, 4])
@pytest.mark.parametrize("epi", ["dequant+y", "silu", "silu_mul"])
@pytest.mark.parametrize("build,K", [(10, 8208), (30, 8192 + 128), (30, 8192 + 256), (31, 8192 + 128)])   # (round 6's schedules take whole 128-byte slices only)
def test_deep_form_other_epilogues(lib, epi, xs, build, K):
    M, N = 260, 1040
    qA, W, sA, sW, _, _ = operands(M, N, K, 0, seed=4)
    g = torch.Generator(device="cpu").manual_seed(5)
    y = (torch.randn((M, N), generator=g) * 0.5).to(torch.float16).to("cuda:0") if "+y" in epi else None
    mul = torch.randn((M, N), generator=g).to(torch.float16).to("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(scratch):
        out = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
        if epi == "silu_mul":
            rc = lib.mixq_int8_fused_dequantize_silu_mul(p(qA), p(W), p(sA), p(sW), None, p(mul), p(out), M, N, K,
                                                         p(scratch), st)
        else:
            fn = lib.mixq_int8_fused_dequantize_silu if epi.startswith("silu") else lib.mixq_int8_fused_dequantize
            rc = fn(p(qA), p(W), p(sA), p(sW), p(y), p(out), M, N, K, p(scratch), st)
        assert rc == 0
        torch.cuda.synchronize()
        return out

    lib.mixq_debug_set_gemm_variant(OFF)
    ref = run(None)
    if build >= 30:
        lib.mixq_debug_set_gemm_variant(1412 - (build - 30))
        build = 30
    lib.mixq_debug_set_gemm_variant(OFF + build + xs)
    scr = torch.zeros(max(lib.mixq_gemm_scratch_size(M, N, K), 16), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        got = run(scr)
        assert b"DEEP" in lib.mixq_debug_last_gemm_kernel()
        assert torch.equal(got, ref)


# the cells of the automatic table (csrc/gemm_kernels.hip deep_plan_auto), one per row of it, on BASELINE (N, K)
AUTO_CELLS = [(256, 12288, 4096, 1), (200, 11008, 4096, 1), (96, 18944, 3584, 1), (768, 4608, 3584, 1),
              (384, 4096, 11008, 2), (512, 3584, 8192, 2), (256, 4096, 11008, 4), (192, 3584, 18944, 4),
              (512, 1280, 8192, 4), (384, 1024, 28672, 8), (128, 12288, 4096, 2), (100, 11008, 4096, 2)]


@pytest.mark.parametrize("M,N,K,xs", AUTO_CELLS)
def test_enqueue_takes_the_deep_form_where_the_table_says_so_and_matches_the_oracle(oracle, lib, M, N, K, xs):
    from test_gpu_parity import bits, run_enqueue
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the table is fitted on 256 CUs")
    A, W, act = make_layer(M, N, K, seed=M + N)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(OFF)
    plain = run_enqueue(A, pk)
    assert b"DEEP" not in lib.mixq_debug_last_gemm_kernel()
    lib.mixq_debug_set_gemm_variant(AUTO)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    assert lib.mixq_enqueue_scratch_size(M, N, K) >= (16384 + tiles * xs * 65536 if xs > 1 else 0)
    for _ in range(2):
        got = run_enqueue(A, pk)
        assert b"DEEP" in lib.mixq_debug_last_gemm_kernel(), lib.mixq_debug_last_gemm_kernel()
        assert np.array_equal(bits(got), bits(plain))
    assert_prefill_parity(oracle, got, A, pk, f"deep form {M} x {N} x {K}")


def test_deep_form_with_k_split_is_capturable_in_a_hip_graph(oracle, lib):
    """quantiser (clears the hand-over words) + deep GEMM with 4 workgroups per tile: replayed on new data in the same buffers."""
    from mixq_tensorrt_llm_amd import plugin
    from test_gpu_parity import bits, run_enqueue, to_dev
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the table is fitted on 256 CUs")
    M, N, K = 256, 4096, 11008
    A, W, act = make_layer(M, N, K, seed=13)
    pk = oracle.pack_linear_weights(W, act)
    layer = plugin.MixQLinear(K, N, device="cuda:0").load(pk)
    x = to_dev(A)
    out = layer(x)          # lazy allocations outside the capture
    torch.cuda.synchronize()
    assert b"DEEP" in lib.mixq_debug_last_gemm_kernel()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = layer(x)
    for trial in range(3):
        A2 = np.ascontiguousarray(np.roll(A, trial * 7, axis=0))
        x.copy_(to_dev(A2))
        g.replay()
        torch.cuda.synchronize()
        lib.mixq_debug_set_gemm_variant(OFF)
        eager = run_enqueue(A2, pk)
        lib.mixq_debug_set_gemm_variant(AUTO)
        assert np.array_equal(bits(out.cpu().numpy()), bits(eager)), trial

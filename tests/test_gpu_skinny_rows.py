"""GPU (-m gpu): the decode-batch GEMM's row-major weight route in 256-byte runs (csrc/gemm_skinny_kernels.hip, WFRAG == 3, round 5):
every load instruction takes 4 rows x 256 contiguous bytes of `weight` and a wave-private LDS tile turns the four registers of a
256-byte group into the group's four MFMA fragments.  Must give the SAME BITS as the plain fragment loads (knob 885) -- through
mixq_enqueue (fragment-major qA) and through the row-major-qA entries, whole and ragged 16-row tiles, N that is not a multiple of 16
rows of workgroups, K ranges that do not divide by the four waves (K / 256 = 1, 5, 9, 43), every epilogue -- and match the oracle."""
import ctypes

import numpy as np
import pytest

from conftest import make_layer
from test_gpu_splitk import operands, p

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ON, OFF = 884, 885


@pytest.fixture
def lib():
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib
    lib.mixq_debug_reset()


@pytest.mark.parametrize("M", [5, 16, 17, 31, 32, 48, 57, 64])
@pytest.mark.parametrize("N,K", [(4096, 4096), (1040, 2304), (528, 256), (144, 1280), (4096, 11008), (3584, 3584),
                                 (5136, 1280), (8192, 2304), (6144, 256)])   # 5120..8192: two feature tiles per workgroup (5136: the last one half empty)
def test_enqueue_same_bits_with_and_without_the_256_byte_runs(lib, oracle, M, N, K):
    from test_gpu_parity import run_enqueue
    A, W, act = make_layer(M, N, K, seed=M + N + K)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(OFF)
    ref = run_enqueue(A, pk)
    kern_off = lib.mixq_debug_last_gemm_kernel()
    lib.mixq_debug_set_gemm_variant(ON)
    got = run_enqueue(A, pk)   # (33..64 rows on N = 5120..8192: the selection itself moves to the skinny kernel with the route on; same bits all the same)
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), f"{M}x{N}x{K} [{kern_off.decode()}]"


@pytest.mark.parametrize("epi", ["dequant", "dequant+y", "silu", "silu_mul"])
@pytest.mark.parametrize("M,N,K,O", [(8, 1040, 2304, 128), (32, 528, 4096, 40), (24, 4096, 1280, 0), (32, 272, 11008, 128), (13, 4096, 256, 256)])
def test_row_major_qa_entries_same_bits(lib, epi, M, N, K, O):
    """the reference-named entries pass a ROW-MAJOR qA (the skinny kernel up to 32 rows): the same route with plain qA fragment loads"""
    if epi != "dequant":
        O = 0
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M * 3 + N + K + O)
    g = torch.Generator(device="cpu").manual_seed(M + N)
    y = (torch.randn((M, N), generator=g) * 0.5).to(torch.float16).to("cuda:0") if "+y" in epi else None
    mul = torch.randn((M, N), generator=g).to(torch.float16).to("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
        if epi == "dequant":
            rc = lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA) if O else None, p(fpW) if O else None, p(out), M, N, K, O, st)
        elif epi == "silu_mul":
            rc = lib.mixq_int8_fused_dequantize_silu_mul(p(qA), p(W), p(sA), p(sW), None, p(mul), p(out), M, N, K, None, st)
        else:
            fn = lib.mixq_int8_fused_dequantize_silu if epi.startswith("silu") else lib.mixq_int8_fused_dequantize
            rc = fn(p(qA), p(W), p(sA), p(sW), p(y), p(out), M, N, K, None, st)
        assert rc == 0
        torch.cuda.synchronize()
        return out

    lib.mixq_debug_set_gemm_variant(OFF)
    ref = run()
    assert b"skinny" in lib.mixq_debug_last_gemm_kernel(), lib.mixq_debug_last_gemm_kernel()
    lib.mixq_debug_set_gemm_variant(ON)
    got = run()
    assert not torch.isnan(got).any()
    assert torch.equal(got, ref)

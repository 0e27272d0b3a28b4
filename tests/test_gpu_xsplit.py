"""GPU (-m gpu): the small-tile kernels with K split over 2 / 4 WORKGROUPS per tile as well (csrc/gemm_kernels.hip, XS) --
decode batches and short chunks on narrow outputs with long K, where a handful of 32x64 / 64x64 tiles would otherwise
stream all of K on a fraction of the CUs.  Every workgroup parks its partial tile and counts itself in; the last one to
arrive adds the others' sums and runs the epilogue.  Same bits as the one-workgroup form; nobody waits for anybody."""
import ctypes

import numpy as np
import pytest

from conftest import make_layer
from test_gpu_splitk import operands, p

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture
def lib():
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib
    lib.mixq_debug_set_gemm_variant(69)   # automatic choice again
    lib.mixq_debug_set_gemm_variant(0)


SHAPES = [(24, 1040, 4240),     # 32x64 tiles (17), partial last K slice, ragged N
          (48, 528, 4224),      # 2 x 9 tiles of 32x64
          (100, 2064, 4352),    # 64x64 tiles: 2 x 33 = 66 -> 2 ways only
          (128, 1024, 8192),    # 64x64 tiles: 2 x 16 = 32
          (7, 784, 4608)]       # below the skinny limit: only when forced


@pytest.mark.parametrize("factor", [2, 4, 8, 6])     # 6 = knob value of 16 ways
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("O", [128, 0, 40, 256])   # 256: side GEMM first, then the addend form (C = D = Out)
def test_small_tile_split_gives_the_bits_of_the_one_workgroup_form(lib, factor, M, N, K, O):
    if factor in (8, 6):
        K = 4 * K        # 8 / 16 ways need 64 / 128 K slices
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M + N + K + O)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fa, fw = (p(fpA), p(fpW)) if O else (None, None)
    lib.mixq_debug_set_gemm_variant(60)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), fa, fw, p(ref), M, N, K, O, st) == 0
    lib.mixq_debug_set_gemm_variant(60 + factor)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    if n == 0:
        pytest.skip("this factor does not apply to the shape (too many tiles)")
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    for round_ in range(4):   # the counter of every tile is left zero by the last workgroup
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), fa, fw, p(out), M, N, K, O, p(scr), n, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"round {round_}"
    assert int(scr[:16384].to(torch.int32).sum()) == 0


@pytest.mark.parametrize("epi", ["dequant+y", "silu", "silu_mul"])
def test_small_tile_split_other_epilogues(lib, epi):
    M, N, K = 40, 1040, 8208
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

    ref = run(None)
    n = lib.mixq_gemm_scratch_size(M, N, K)     # automatic choice: 34 tiles of 32x64, 65 slices -> 4 ways (16 slices each)
    assert n == 16384 + 34 * 4 * 32 * 64 * 4
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        assert torch.equal(run(scr), ref)


@pytest.mark.parametrize("M", [24, 32, 64, 128])
def test_enqueue_uses_it_for_decode_batches_and_matches_the_oracle(oracle, lib, M):
    """Automatic choice through the plugin: the quantiser clears the hand-over words on its way, the GEMM splits K."""
    from test_gpu_parity import REL_TOL, bits, rel_err, run_enqueue
    N, K = 1024, 8192
    A, W, act = make_layer(M, N, K, seed=40 + M)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(60)
    plain = run_enqueue(A, pk)
    lib.mixq_debug_set_gemm_variant(69)
    assert lib.mixq_gemm_scratch_size(M, N, K) > 0
    got = run_enqueue(A, pk)
    assert np.array_equal(bits(got), bits(plain))
    want = oracle.linear_prefill(A, pk["weight"], pk["weights_scaling_factor"], pk["fp_weight"], pk["fp_ind"])
    assert rel_err(got, want) < REL_TOL


def test_shared_scratch_across_shapes_and_both_split_forms(lib):
    """One scratch per stream serves every layer: small-tile splits, 256x256 splits and unsplit launches in any order."""
    lib.mixq_debug_set_gemm_variant(79)
    lib.mixq_debug_set_gemm_variant(69)
    shapes = [(32, 4096, 11008), (1024, 4096, 11008), (64, 1024, 28672), (512, 12288, 11008), (128, 4096, 16384),
              (24, 2048, 8192), (256, 4096, 16384), (32, 4096, 11008)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sizes = [lib.mixq_gemm_scratch_size(*s) for s in shapes]
    assert all(n > 0 for n in sizes)
    scr = torch.zeros(max(sizes), dtype=torch.uint8, device="cuda:0")
    sets, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        ops = operands(M, N, K, 128, seed=500 + i)
        r = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed(*[p(t) for t in ops], p(r), M, N, K, 128, st) == 0     # no scratch: never split
        sets.append(ops), refs.append(r)
    for round_ in range(3):
        outs = []
        for (M, N, K), ops, n in zip(shapes, sets, sizes):
            o = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
            assert lib.mixq_gemm_mixed_scratch(*[p(t) for t in ops], p(o), M, N, K, 128, p(scr), n, st) == 0
            outs.append(o)
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert torch.equal(o, r), (round_, shapes[i])
    assert int(scr[:16384].to(torch.int32).sum()) == 0


def test_enqueue_with_small_tile_split_is_capturable_in_a_hip_graph(oracle, lib):
    """quantiser (clears the hand-over words) + split GEMM: two kernel nodes, replayed on new data in the same buffers."""
    from mixq_tensorrt_llm_amd import plugin
    from test_gpu_parity import bits, run_enqueue, to_dev
    M, N, K = 48, 1024, 8192
    A, W, act = make_layer(M, N, K, seed=13)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(69)
    assert lib.mixq_gemm_scratch_size(M, N, K) > 0
    layer = plugin.MixQLinear(K, N, device=torch.device("cuda:0")).load(pk)
    x = to_dev(np.zeros_like(A))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = layer(x)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = layer(x)
    torch.cuda.current_stream().wait_stream(side)
    for trial in range(3):
        A2 = np.ascontiguousarray(np.roll(A, trial * 7, axis=0))
        x.copy_(to_dev(A2))
        graph.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(M, N)
        lib.mixq_debug_set_gemm_variant(60)
        eager = run_enqueue(A2, pk)
        lib.mixq_debug_set_gemm_variant(69)
        assert np.array_equal(bits(got), bits(eager)), trial

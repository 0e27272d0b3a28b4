"""GPU (-m gpu): the K split over 2 / 4 workgroups per tile (csrc/gemm_pp_kernels.hip, SPLITK) -- mid-size problems whose
256x256 tiles cover at most half / a quarter of the CUs.  The split form must give the SAME BITS as the
one-workgroup-per-tile kernels (integer partial sums commute; the epilogue is shared), on ragged shapes, partial last K
slices, every epilogue, repeated launches on one scratch (the arrival words re-arm themselves), inside a HIP graph, and
through the plugin's enqueue, which carves the scratch from its workspace."""
import ctypes

import numpy as np
import pytest

from conftest import assert_prefill_parity, make_layer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture
def lib():
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib
    lib.mixq_debug_set_gemm_variant(79)   # automatic choice again
    lib.mixq_debug_set_gemm_variant(1240)


def operands(M, N, K, O, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = "cuda:0"
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, generator=g).to(d)
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g).to(d)
    sA = (torch.rand(M, generator=g) * 0.05 + 0.01).to(torch.float16).to(d)
    sW = (torch.rand(N, generator=g) * 4e-4 + 1e-4).to(torch.float16).to(d)
    fpA = torch.randn((M, max(O, 8)), generator=g).to(torch.float16).to(d)[:, :O].contiguous()
    fpW = (torch.randn((N, max(O, 8)), generator=g) * 0.02).to(torch.float16).to(d)[:, :O].contiguous()
    return qA, W, sA, sW, fpA, fpW


SHAPES = [(300, 528, 2064),     # ragged M and N, partial last K slice, 2 x 3 tiles
          (512, 1024, 2048),    # whole tiles, 16 slices (4 per workgroup in the 4-way form)
          (257, 272, 4224),     # one row / 16 columns into the second tile, odd slice count
          (1100, 1536, 2176),   # 5 x 6 tiles: 8 groups of XCDs unevenly filled
          (200, 784, 2304)]     # fewer rows than one 256-row tile (129..255 rows take the 4- / 8-way plan since round 4)


@pytest.mark.parametrize("factor", [2, 4, 8])
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("O", [128, 0, 40, 256])   # 256: side GEMM first, then the addend form (C = D = Out)
def test_split_form_gives_the_bits_of_the_one_workgroup_form(lib, factor, M, N, K, O):
    if factor == 8:
        K = 2 * K + 16          # 8 ways need >= 32 K slices
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M + N + K + O)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    fa, fw = (p(fpA), p(fpW)) if O else (None, None)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), fa, fw, p(ref), M, N, K, O, st) == 0
    lib.mixq_debug_set_gemm_variant(70 + factor)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    assert n == 16384 + tiles * factor * 262144   # hand-over words + S slots of 256 KiB per split tile
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    for round_ in range(3):   # the same scratch again and again: the last reader of every tile re-arms its words
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), fa, fw, p(out), M, N, K, O, p(scr), n, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref), f"round {round_}"
    assert int(scr[:16384].to(torch.int32).sum()) == 0, "arrival words left non-zero"
    # a scratch that is too small (or absent) selects the one-workgroup form, silently and correctly
    out = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), fa, fw, p(out), M, N, K, O, p(scr), n - 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("M,N,K", [(2048, 8448, 2176),    # 8 x 33 = 264 tiles: one whole wave of 256 + 8 split tiles
                                   (2000, 8464, 2064)])   # ragged, 8 x 34 = 272 tiles, partial last K slice
def test_whole_waves_solo_plus_split_tail_in_one_launch(lib, factor, M, N, K):
    """More tiles than CUs: the whole waves of tiles run one workgroup per tile, the tiles of the last, partial wave are
    split -- all inside one launch of the split kernel."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if not cus < tiles <= cus + cus // factor:
        pytest.skip(f"shape is sized for 256 CUs (device has {cus})")
    O = 128
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M + N)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    lib.mixq_debug_set_gemm_variant(70)
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st) == 0
    lib.mixq_debug_set_gemm_variant(70 + factor)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    tail = tiles % cus
    assert n == 16384 + tail * factor * 262144
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr), n,
                                           st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("factor", [2, 4, 8])
@pytest.mark.parametrize("epi", ["dequant", "dequant+y", "silu", "silu+y", "silu_mul"])
def test_split_form_every_epilogue_of_the_p_flavour(lib, factor, epi):
    """int8FusedDequantize / ...Silu / ...SiluMul (mixlib): the `workspace` argument of the reference's signature carries
    the scratch."""
    M, N, K = 520, 784, 2320 if factor < 8 else 4624
    qA, W, sA, sW, _, _ = operands(M, N, K, 0, seed=11)
    g = torch.Generator(device="cpu").manual_seed(3)
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
    lib.mixq_debug_set_gemm_variant(70 + factor)
    scr = torch.zeros(lib.mixq_gemm_scratch_size(M, N, K), dtype=torch.uint8, device="cuda:0")
    assert scr.numel() > 0
    for _ in range(2):
        assert torch.equal(run(scr), ref)


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_split_form_through_enqueue_matches_the_oracle(oracle, lib, factor):
    """The plugin carves the scratch from its workspace and zeroes the arrival words itself (the workspace is shared)."""
    from test_gpu_parity import REL_TOL, bits, rel_err, run_enqueue
    M, N, K = 600, 784, 2304 if factor < 8 else 4352
    A, W, act = make_layer(M, N, K, seed=5)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(70)
    plain = run_enqueue(A, pk)
    lib.mixq_debug_set_gemm_variant(70 + factor)
    assert lib.mixq_gemm_scratch_size(M, N, K) > 0
    got = run_enqueue(A, pk)
    assert np.array_equal(bits(got), bits(plain))
    assert_prefill_parity(oracle, got, A, pk, f"K split {factor} ways through enqueue")


@pytest.mark.parametrize("M,N,K,ways", [(192, 3584, 18944, 8),    # Qwen2-7B down projection, decode batch of 192: 14 tiles, 148 slices
                                        (144, 8192, 8192, 4),     # 32 tiles = 1/8 of the CUs, K = 8192: 4 ways
                                        (224, 5120, 8192, 8),     # 20 tiles, 4 x 80 tiles of 64 x 64 = more than one wave: 8 ways
                                        (192, 5120, 8192, 0),     # ... 3 x 80 fit one wave: the small tiles keep it
                                        (192, 12288, 4096, 0),    # short K: never
                                        (128, 3584, 18944, 0)])   # up to 128 rows: the small-tile forms
def test_rows_129_to_255_take_the_plan_of_one_tile_row(oracle, lib, M, N, K, ways):
    """Round 4 (cold-weight fit, profiles/r04_splitk_cold_fit.txt): a decode batch of 129..255 rows on a long-K linear runs the 256 x 256
    tiles with K split 4 / 8 ways, bit-identical to the one-workgroup-per-tile forms, and through mixq_enqueue equal to the oracle."""
    from test_gpu_parity import bits, run_enqueue
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus != 256:
        pytest.skip("plan thresholds are written for 256 CUs")
    lib.mixq_debug_set_gemm_variant(79)
    lib.mixq_debug_set_gemm_variant(1241)   # (round 5: the mid-M deep form takes some of these cells first; this test is about the plan under it)
    tiles = (N + 255) // 256
    n = lib.mixq_gemm_scratch_size(M, N, K)
    if ways:
        assert n == 16384 + tiles * ways * 262144
        assert lib.mixq_enqueue_scratch_size(M, N, K) == n
    else:
        assert n < 16384 + tiles * 2 * 262144     # nothing, or the small-tile form's 16-KiB slots
    A, W, act = make_layer(M, N, K, seed=M + N)
    pk = oracle.pack_linear_weights(W, act)
    got = run_enqueue(A, pk)
    kern = lib.mixq_debug_last_gemm_kernel().decode()
    assert ("SPLITK" in kern) == bool(ways), kern
    lib.mixq_debug_set_gemm_variant(70)
    plain = run_enqueue(A, pk)
    assert np.array_equal(bits(got), bits(plain))
    assert_prefill_parity(oracle, got, A, pk, f"{M} rows on {N} x {K}")


def test_automatic_choice_and_graph_replay(lib):
    """Default selection on a shape it is made for (64 tiles, 86 K slices -> 4 workgroups per tile), captured into a HIP
    graph and replayed: every replay re-uses the scratch the previous one left re-armed."""
    M, N, K, O = 1024, 4096, 11008, 128
    lib.mixq_debug_set_gemm_variant(79)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    assert n == 16384 + 64 * 4 * 262144
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=1)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st) == 0
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    out = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sst = ctypes.c_void_p(side.cuda_stream)
        call = lambda: lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                                   p(scr), n, sst)
        assert call() == 0
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            assert call() == 0
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_mixlib_wrappers_pick_up_a_per_stream_scratch(lib):
    from mixq_tensorrt_llm_amd import mixlib
    M, N, K = 512, 12288, 11008       # 96 tiles, long K -> 2 workgroups per tile (at K = 4096 the 128x256 tiles take it)
    assert lib.mixq_gemm_scratch_size(M, N, K) > 0
    qA, W, sA, sW, _, _ = operands(M, N, K, 0, seed=2)
    lib.mixq_debug_set_gemm_variant(70)
    ref = mixlib.int8FusedDequantize(qA, W, sA.reshape(M, 1), sW.reshape(1, N), None, M, N, K)
    lib.mixq_debug_set_gemm_variant(79)
    got = mixlib.int8FusedDequantize(qA, W, sA.reshape(M, 1), sW.reshape(1, N), None, M, N, K)
    assert mixlib.gemm_scratch(qA, M, N, K) is not None
    assert torch.equal(got, ref)


@pytest.mark.parametrize("factor", [2, 4, 8])
@pytest.mark.parametrize("M,N,K", SHAPES[:3] + [(2048, 8448, 2176)])
def test_nobody_waits_every_workgroup_but_the_last_arriver_defers(lib, factor, M, N, K):
    """Patience 0 (variant 90): a workgroup that does not find all its partners' tickets at its first look parks its own
    share too, marks it DEFERRED and leaves; the last arriver of the tile finishes every deferred share.  Same bits, words
    re-armed -- the path a launch takes when its workgroups are NOT co-resident (HIP promises no dispatch order)."""
    if factor == 8:
        K = 2 * K + 16
    O = 128
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M + N + K + 1)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st) == 0
    lib.mixq_debug_set_gemm_variant(70 + factor)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    if n == 0:
        pytest.skip("this factor does not apply to the shape")
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    try:
        for patience in (90, 91, 90):   # deferring, default, deferring again on the same scratch
            lib.mixq_debug_set_gemm_variant(patience)
            for round_ in range(4):
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda:0")
                assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr), n,
                                                   st) == 0
                torch.cuda.synchronize()
                assert torch.equal(out, ref), (patience, round_)
            assert int(scr[:16384].to(torch.int32).sum()) == 0, "hand-over words left non-zero"
    finally:
        lib.mixq_debug_set_gemm_variant(91)


def test_split_launches_on_two_streams_under_a_cu_hog_1000_iterations(lib):
    """VERDICT r1 item 7 / ADVICE r1: two split launches on two streams (a scratch each) while a long-running stream of
    large GEMMs occupies the CUs: 1000 iterations, no trap (there is none any more), bit-identical results."""
    M, N, K, O = 1024, 4096, 11008, 128
    lib.mixq_debug_set_gemm_variant(79)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    assert n > 0
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=31)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st0) == 0
    torch.cuda.synchronize()
    hog, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.randn((8192, 8192), device="cuda:0", dtype=torch.float16)
    scr = [torch.zeros(n, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    out = [torch.empty((M, N), dtype=torch.float16, device="cuda:0") for _ in range(2)]
    bad = 0
    for it in range(1000):
        if it % 4 == 0:
            with torch.cuda.stream(hog):
                y = (x @ x).clamp_(-1, 1)   # ~1 ms of every CU, on its own queue
        for which, stream in enumerate((s1, s2)):
            assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out[which]), M, N, K, O,
                                               p(scr[which]), n, ctypes.c_void_p(stream.cuda_stream)) == 0
        if it % 100 == 99:
            torch.cuda.synchronize()
            bad += int(not torch.equal(out[0], ref)) + int(not torch.equal(out[1], ref))
    torch.cuda.synchronize()
    del y
    assert bad == 0 and torch.equal(out[0], ref) and torch.equal(out[1], ref)
    assert int(scr[0][:16384].to(torch.int32).sum()) == 0 and int(scr[1][:16384].to(torch.int32).sum()) == 0


def test_split_form_next_to_other_work_on_the_gpu(lib):
    """The protocol must make progress when the kernel does not get the whole chip: (a) behind / next to large
    one-workgroup-per-tile GEMMs on another stream, (b) two split GEMMs on two streams with a scratch each.  Every result
    must still be the plain kernel's bits."""
    M, N, K, O = 1024, 4096, 11008, 128
    lib.mixq_debug_set_gemm_variant(79)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    assert n > 0
    qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=21)
    ref = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
    st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st0) == 0
    # the neighbour: 8192 x 12288 x 4096 (1536 tiles, ~0.35 ms per launch, every CU busy)
    Mb, Nb, Kb = 8192, 12288, 4096
    qB, WB, sAB, sWB, fpAB, fpWB = operands(Mb, Nb, Kb, O, seed=22)
    outB = torch.empty((Mb, Nb), dtype=torch.float16, device="cuda:0")
    torch.cuda.synchronize()
    sa_, sb_, sc_ = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    scr = [torch.zeros(n, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    outs = [[torch.empty((M, N), dtype=torch.float16, device="cuda:0") for _ in range(12)] for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(12):
        assert lib.mixq_gemm_mixed(p(qB), p(WB), p(sAB), p(sWB), p(fpAB), p(fpWB), p(outB), Mb, Nb, Kb, O,
                                   ctypes.c_void_p(sb_.cuda_stream)) == 0
        for which, stream in enumerate((sa_, sc_)):
            assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(outs[which][it]), M, N, K,
                                               O, p(scr[which]), n, ctypes.c_void_p(stream.cuda_stream)) == 0
    torch.cuda.synchronize()
    for which in range(2):
        for it in range(12):
            assert torch.equal(outs[which][it], ref), (which, it)


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_scratch_reuse_with_changing_data_never_sees_stale_partial_sums(lib, factor):
    """Back-to-back launches on one stream and ONE scratch with alternating operands: a hand-over that could be served
    from a stale cache line (the previous launch's partial sums at the same scratch address, in another XCD's L2) would
    produce the other operand set's result."""
    M, N, K, O = (1024, 4096, 4352, 128) if factor < 8 else (512, 4096, 4352, 128)   # 64 / 32 tiles, 34 slices
    sets = [operands(M, N, K, O, seed=100 + i) for i in range(2)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    refs = []
    lib.mixq_debug_set_gemm_variant(70)
    for qA, W, sA, sW, fpA, fpW in sets:
        r = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(r), M, N, K, O, st) == 0
        refs.append(r)
    assert not torch.equal(refs[0], refs[1])
    lib.mixq_debug_set_gemm_variant(70 + factor)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    assert n > 0
    scr = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    outs = [torch.empty((M, N), dtype=torch.float16, device="cuda:0") for _ in range(10)]
    torch.cuda.synchronize()
    for it, out in enumerate(outs):          # no host synchronisation in between
        qA, W, sA, sW, fpA, fpW = sets[it & 1]
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr), n,
                                           st) == 0
    torch.cuda.synchronize()
    for it, out in enumerate(outs):
        assert torch.equal(out, refs[it & 1]), it


def test_enqueue_with_split_form_is_capturable_in_a_hip_graph(oracle, lib):
    """mixq_enqueue on a split shape = memset of the arrival words + quantiser + split GEMM: all three are captured and
    replayed on new data in the same buffers."""
    from mixq_tensorrt_llm_amd import plugin
    from test_gpu_parity import bits, run_enqueue, to_dev
    M, N, K = 600, 784, 2304
    A, W, act = make_layer(M, N, K, seed=9)
    pk = oracle.pack_linear_weights(W, act)
    lib.mixq_debug_set_gemm_variant(72)
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
        A2 = np.ascontiguousarray(np.roll(A, trial * 5, axis=0))
        x.copy_(to_dev(A2))
        graph.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(M, N)
        lib.mixq_debug_set_gemm_variant(70)
        eager = run_enqueue(A2, pk)
        lib.mixq_debug_set_gemm_variant(72)
        assert np.array_equal(bits(got), bits(eager)), trial


@pytest.mark.parametrize("M,N,K", [(1024, 4608, 3584),     # Qwen2-7B qkv, short prefill (72 tiles)
                                   (2048, 3584, 18944),    # Qwen2-7B down projection (112 tiles)
                                   (8192, 1024, 28672),    # Llama-2-70B down projection, TP = 8 shard (128 tiles)
                                   (4096, 1280, 8192),     # Llama-2-70B qkv, TP = 8 shard (80 tiles)
                                   (1536, 11008, 4096)])   # Llama-2-7B gate: 258 tiles = one wave + 2 split tiles
def test_model_shapes_of_the_baseline_configs_automatic_choice(lib, oracle, M, N, K):
    """SURVEY 8d configs 4 / 5 (and a tail case of config 2) at the chunk sizes where the split form is selected by
    default: same bits as the one-workgroup kernels."""
    O = 128
    g = torch.Generator(device="cuda:0").manual_seed(M + N + K)
    d = "cuda:0"
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=d, generator=g)
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=d, generator=g)
    sA = (torch.rand(M, device=d, generator=g) * 0.05 + 0.01).to(torch.float16)
    sW = (torch.rand(N, device=d, generator=g) * 4e-4 + 1e-4).to(torch.float16)
    fpA = torch.randn((M, O), device=d, generator=g).to(torch.float16)
    fpW = (torch.randn((N, O), device=d, generator=g) * 0.02).to(torch.float16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.mixq_debug_set_gemm_variant(70)
    ref = torch.empty((M, N), dtype=torch.float16, device=d)
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st) == 0
    lib.mixq_debug_set_gemm_variant(79)
    n = lib.mixq_gemm_scratch_size(M, N, K)
    takes_half_height_tiles = False
    scr = torch.zeros(max(n, 16), dtype=torch.uint8, device=d)
    out = torch.empty((M, N), dtype=torch.float16, device=d)
    for _ in range(2):
        out.zero_()
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr), n,
                                           st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        takes_half_height_tiles = b"pp128" in lib.mixq_debug_last_gemm_kernel()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        # on a 256-CU part every one of these shapes leaves the one-workgroup-per-256x256-tile form: either K is split
        # over workgroups (scratch) or -- short K, about one wave of them -- 128 x 256 tiles are used (no scratch)
        assert (n > 0) != takes_half_height_tiles, (n, lib.mixq_debug_last_gemm_kernel())
    # against the ORACLE on 64 sampled rows (this used to be a 64-point host recomputation at 2e-3): int8 GEMM,
    # fp16 side product and dequant epilogue restated from the reference; element-wise bound + north-star 1e-3
    from conftest import SIDE_GAMMA, assert_elementwise, ulp16
    rows = np.unique(np.concatenate([[0, 255, 256, M - 1], np.random.default_rng(M + N).integers(0, M, 60)]))[:64]
    ridx = torch.from_numpy(rows).to(d)
    qa_s, sa_s, fa_s = qA[ridx].cpu().numpy(), sA[ridx].cpu().numpy(), fpA[ridx].cpu().numpy()
    acc = oracle.gemm_s8s8s32(qa_s, W.cpu().numpy())
    P16 = oracle.gemm_fp16(fa_s, fpW.cpu().numpy())
    want = oracle.dequant_epilogue(acc, sa_s, sW.cpu().numpy(), P16)
    got = out[ridx].cpu().numpy()
    g64, w64 = got.astype(np.float64), want.astype(np.float64)
    assert np.abs(g64 - w64).max() / np.abs(w64).max() < 1e-3
    slack = ulp16(P16) + SIDE_GAMMA * (np.abs(fa_s.astype(np.float32)) @ np.abs(fpW.cpu().numpy().astype(np.float32)).T).astype(np.float64)
    assert_elementwise(got, want, slack, f"{M} x {N} x {K} ({lib.mixq_debug_last_gemm_kernel().decode()})")


def test_one_scratch_serves_launches_of_different_shapes(lib):
    """The mixlib wrappers keep ONE scratch per stream for every layer: the hand-over words sit at a fixed place at the
    start of the scratch, so the data a launch parks can never be mistaken for another shape's arrival words."""
    lib.mixq_debug_set_gemm_variant(79)
    shapes = [(1024, 4096, 11008), (512, 12288, 11008), (1536, 11008, 4096), (2048, 4096, 11008), (768, 4096, 8192),
              (1024, 4096, 11008)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scr = torch.zeros(max(lib.mixq_gemm_scratch_size(*s) for s in shapes), dtype=torch.uint8, device="cuda:0")
    sets, refs = [], []
    lib.mixq_debug_set_gemm_variant(70)
    for i, (M, N, K) in enumerate(shapes):
        ops = operands(M, N, K, 128, seed=300 + i)
        r = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
        assert lib.mixq_gemm_mixed(*[p(t) for t in ops], p(r), M, N, K, 128, st) == 0
        sets.append(ops), refs.append(r)
    lib.mixq_debug_set_gemm_variant(79)
    for round_ in range(3):
        outs = []
        for (M, N, K), ops in zip(shapes, sets):      # back to back, no synchronisation
            o = torch.empty((M, N), dtype=torch.float16, device="cuda:0")
            n = lib.mixq_gemm_scratch_size(M, N, K)
            assert n > 0
            assert lib.mixq_gemm_mixed_scratch(*[p(t) for t in ops], p(o), M, N, K, 128, p(scr), n, st) == 0
            outs.append(o)
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert torch.equal(o, r), (round_, shapes[i])
    assert int(scr[:16384].to(torch.int32).sum()) == 0

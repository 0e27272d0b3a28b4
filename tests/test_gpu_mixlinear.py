"""GPU (-m gpu): P-flavour `MixLinear_GEMM` with dynamic outlier detection (MixQ/src/mixquant/modules/linear.py:155-286)
through the C ABI, against the oracle's numpy restatement of the same state machine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REL_TOL = 1e-3


def rel_err(got, want):
    g, w = got.astype(np.float64), want.astype(np.float64)
    return np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.parametrize("M,K", [(1, 64), (7, 4096), (333, 4096), (1024, 11008), (64, 32768 + 64)])
def test_find_outliers_matches_unique_where(oracle, M, K):
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(M * 7 + K)
    A = rng.standard_normal((M, K)).astype(np.float16)
    cols = rng.choice(K, size=min(K, 37), replace=False)
    A[rng.integers(0, M, cols.size), cols] = np.float16(9.5) * rng.choice([-1, 1], cols.size)
    A[0, K - 1] = np.float16(-6.01)          # just above sigma, last column
    A[M - 1, 0] = np.float16(6.0)            # == sigma: NOT an outlier (strict >)
    if M > 2:
        A[1, 5] = np.nan                     # NaN never compares greater
        A[2, 9] = np.inf
    got = mixlinear.find_outliers(dev(A), 6.0).cpu().numpy()
    want = oracle.find_outliers(A, 6.0)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    # capacity smaller than the set: the first `capacity` indices, no overflow
    cap = max(1, want.size // 2)
    assert np.array_equal(mixlinear.find_outliers(dev(A), 6.0, capacity=cap).cpu().numpy(), want[:cap])
    # nothing above sigma -> empty
    assert mixlinear.find_outliers(dev(np.zeros((M, K), np.float16)), 6.0).numel() == 0


def test_dequant_weight_columns_bit_exact(oracle):
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(5)
    N, K = 1000, 4096
    W = rng.integers(-128, 128, (N, K), dtype=np.int8)
    s = (rng.random(N) * 1e-2 + 1e-4).astype(np.float16)
    ind = np.sort(rng.choice(K, 77, replace=False)).astype(np.int32)
    got = mixlinear.dequant_weight_columns(dev(W), dev(s.reshape(1, N)), dev(ind)).cpu().numpy()
    want = oracle.dequant_weight_columns(W, s, ind)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("odd_o", [37, 136, 129])
def test_outlier_product_any_width(oracle, odd_o):
    """The side product of a dynamic outlier set has an arbitrary width O (not a multiple of 8)."""
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(odd_o)
    a = (rng.standard_normal((70, odd_o)) * 8).astype(np.float16)
    w = (rng.standard_normal((96, odd_o)) * 0.02).astype(np.float16)
    got = mixlinear.outlier_product(dev(a), dev(w)).cpu().numpy()
    assert rel_err(got, oracle.gemm_fp16(a, w)) < REL_TOL


def make_layer(rng, N, K, bias):
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    b = (rng.standard_normal(N) * 0.1).astype(np.float16) if bias else None
    return W, b


@pytest.mark.parametrize("bias", [False, True])
def test_mixlinear_dynamic_outliers_state_machine(oracle, bias):
    """Three forwards: call 1 finds outlier set S1, call 2 extends it with new columns (hstack order), call 3 runs with
    the frozen set (cache.stop = 2).  ind / weight_cache / output follow the oracle at every step; x is mutated (its
    outlier columns zeroed) like in the reference."""
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(11)
    N, K, M = 512, 1024, 48
    W, b = make_layer(rng, N, K, bias)
    cache = mixlinear.MixLibCache(inputdim=1024, sigma=6, device="cuda:0")
    layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), None if b is None else torch.from_numpy(b),
                                                 cache=cache, dev="cuda:0")
    # from_linear parity: scale / int8 weight (linear.py:113-120)
    sc = (np.abs(W).max(axis=1) / np.float16(127)).astype(np.float16)
    assert np.array_equal(layer.scale_col.cpu().numpy().reshape(-1).view(np.uint16), sc.view(np.uint16))
    qw = np.rint(W.astype(np.float16) / sc[:, None]).astype(np.int8)
    assert np.array_equal(layer.q_weight.cpu().numpy(), qw)
    st = oracle.MixLinearState(qw, sc, sigma=6.0, stop=2, bias=b)

    sets = [[3, 500, 77], [900, 12, 77], [5]]   # column 77 repeats: already zeroed by the first extraction -> not re-found
    for step, hot in enumerate(sets):
        x = rng.standard_normal((M, K)).astype(np.float16)
        x[rng.integers(0, M, len(hot)), hot] = np.float16(20.0)
        x_ref = x.copy()
        xt = dev(x)
        got = layer.forward(xt, cache, True).cpu().numpy()
        want = oracle.mixlinear_forward(st, x_ref)
        assert np.array_equal(layer.ind.cpu().numpy(), st.ind), f"step {step}: outlier index set"
        assert layer.add_outliers == st.add_outliers and layer.cnt == st.cnt
        if st.ind.size:
            assert np.array_equal(layer.weight_cache.cpu().numpy().view(np.uint16), st.weight_cache.view(np.uint16))
        assert np.array_equal(xt.cpu().numpy().view(np.uint16), x_ref.view(np.uint16)), "input mutation (zeroed outliers)"
        assert rel_err(got, want) < REL_TOL, f"step {step}"
    assert st.ind.tolist() == [3, 77, 500, 12, 900]  # sorted within a call, appended across calls; frozen afterwards
    assert not layer.add_outliers


def test_mixlinear_no_outliers_and_3d_input(oracle):
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(3)
    N, K = 256, 512
    W, _ = make_layer(rng, N, K, False)
    cache = mixlinear.MixLibCache(inputdim=256, sigma=6, device="cuda:0")
    layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), cache=cache, dev="cuda:0")
    st = oracle.MixLinearState(layer.q_weight.cpu().numpy(), layer.scale_col.cpu().numpy(), sigma=6.0)
    x = (rng.standard_normal((2, 9, K)) * 0.5).astype(np.float16)       # nothing above sigma
    got = layer.forward(dev(x), cache, True).cpu().numpy()
    assert got.shape == (2, 9, N)
    want = oracle.mixlinear_forward(st, x.reshape(-1, K).copy())
    assert layer.ind.numel() == 0 and rel_err(got.reshape(-1, N), want) < REL_TOL


@pytest.mark.parametrize("with_outliers", [False, True])
def test_mixlinear_arch9_takes_the_reference_sm90_route(oracle, with_outliers):
    """linear.py:231-238 / :317-324: with `arch == 9` the module runs mixlib.gemm -> (outlier product) -> mixlib.dequantizeInt8[Silu]
    instead of the fused GEMM.  Op for op against the oracle's restatement of those two kernels (bit-exact given the side product: the
    unfused epilogue rounds the scaled sum to fp16 BEFORE adding y, unlike the fused one), and close to the fused route's answer."""
    from mixq_tensorrt_llm_amd import mixlib, mixlinear
    rng = np.random.default_rng(41)
    N, K, M = 384, 1024, 24
    W, _ = make_layer(rng, N, K, False)
    outs = {}
    for arch in (0, 9):
        cache = mixlinear.MixLibCache(inputdim=64, sigma=6, device="cuda:0")
        layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), cache=cache, dev="cuda:0")
        layer.arch = arch
        x = (rng.standard_normal((M, K)) * 0.5).astype(np.float16) if arch == 0 else x0.copy()
        if arch == 0:
            if with_outliers:
                x[3, 70] = np.float16(25.0)
                x[11, 901] = np.float16(-31.0)
            x0 = x.copy()
        xt = dev(x)
        got = layer.forward(xt, cache, True)
        torch.cuda.synchronize()
        assert layer.ind.numel() == (2 if with_outliers else 0)
        outs[arch] = got.cpu().numpy()
        if arch == 9:   # the same ops on the same cache contents, restated
            acc = oracle.gemm_s8s8s32(cache.q_xcache.cpu().numpy(), layer.q_weight.cpu().numpy())
            assert np.array_equal(mixlib.gemm(cache.q_xcache, layer.q_weight, M, N, K).cpu().numpy(), acc)
            y = (mixlinear.outlier_product(cache.activation_outliers, layer.weight_cache).cpu().numpy() if with_outliers
                 else np.zeros((M, N), np.float16))
            want = oracle.dequantization(acc, cache.x_scale[:M].cpu().numpy(), layer.scale_col.cpu().numpy(), y)
            assert np.array_equal(outs[9].view(np.uint16), want.view(np.uint16))
            # the gate projection's twin (:317-324) on the same cache
            gate = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), cache=cache, dev="cuda:0")
            gate.arch = 9
            g = gate.forward_without_preconditionFusedSilu(xt, cache).cpu().numpy()
            yg = (mixlinear.outlier_product(cache.activation_outliers, gate.weight_cache).cpu().numpy() if with_outliers
                  else np.zeros((M, N), np.float16))
            want_g = oracle.dequantization_silu(acc, cache.x_scale[:M].cpu().numpy(), gate.scale_col.cpu().numpy(), yg)
            d = np.abs(g.astype(np.float64) - want_g.astype(np.float64))
            assert (d <= 2.0 ** -10 * np.maximum(np.abs(want_g.astype(np.float64)), 2.0 ** -14) + 1e-7).all()   # SiLU: <= 1 ulp (conftest bounds)
    assert rel_err(outs[9], outs[0]) < REL_TOL


def test_mixlinear_weight_only_mode(oracle):
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(4)
    N, K = 512, 1024
    W, _ = make_layer(rng, N, K, False)
    layer = mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), weight_only=True,
                                                 cache=mixlinear.MixLibCache(64, device="cuda:0"), dev="cuda:0")
    x = rng.standard_normal((3, K)).astype(np.float16)
    got = layer.forward(dev(x)).cpu().numpy()
    q_un, scales = oracle.eetq_symmetric_quantize(W.T.copy())
    assert rel_err(got, oracle.w8a16_gemv(x, q_un, scales)) < 1e-3


def test_fused_norm_and_llama_mlp_block(oracle):
    """norm.py:6-40 + mlp.py:37-68: FasterTransformerRMSNorm (next_layer = up_proj) fills the cache in one pass,
    up_proj consumes it (and discovers outlier columns dynamically), gate_proj re-uses the SAME quantised activation with
    SiLU fused into its epilogue and adopts the new columns, gate *= up, down_proj quantises its own input.
    Followed step by step against the oracle; then the packaged MixLlamaMLP must give the same bits as the steps."""
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(23)
    H, F, M = 512, 1024, 40
    Wu, _ = make_layer(rng, F, H, False)
    Wg, _ = make_layer(rng, F, H, False)
    Wd, _ = make_layer(rng, H, F, False)
    gamma = (1.0 + 0.1 * rng.standard_normal(H)).astype(np.float16)
    eps = 1e-6
    hidden = rng.standard_normal((M, H)).astype(np.float16)
    hidden[3, 17] = np.float16(60.0)        # survives the norm as |x| > sigma -> dynamic outlier column 17
    hidden[9, 300] = np.float16(-45.0)

    def build():
        cache = mixlinear.MixLibCache(inputdim=64, sigma=6, device="cuda:0")
        mk = lambda W: mixlinear.MixLinear_GEMM.from_linear(torch.from_numpy(W), cache=cache, dev="cuda:0")  # noqa: E731
        up, gate, down = mk(Wu), mk(Wg), mk(Wd)
        norm = mixlinear.FasterTransformerRMSNorm(dev(gamma), eps, cache)
        norm.next_layer = up
        return cache, norm, up, gate, down

    # ---- oracle, step by step -------------------------------------------------------------------------
    cache, norm, up, gate, down = build()
    st = {n: oracle.MixLinearState(l.q_weight.cpu().numpy(), l.scale_col.cpu().numpy(), sigma=6.0)
          for n, l in (("up", up), ("gate", gate), ("down", down))}
    x_ref, outl, qx, xs = oracle.rmsnorm_extract_quant(hidden, gamma, eps, np.zeros(0, np.int32))
    x_ref = np.ascontiguousarray(x_ref)
    assert np.abs(x_ref.astype(np.float32)).max() > 6.0, "the test needs a dynamic outlier"
    new_ind = oracle.find_outliers(x_ref, 6.0)
    act_out = oracle.extract_outliers(x_ref, new_ind, set_zero=True)
    st["up"].weight_cache = oracle.dequant_weight_columns(st["up"].q_weight, st["up"].scale_col, new_ind)
    st["up"].ind = new_ind
    qx, xs = oracle.quant_rows(x_ref)
    up_want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(qx, st["up"].q_weight), xs, st["up"].scale_col,
                                      C=oracle.gemm_fp16(act_out, st["up"].weight_cache))
    wc_gate = oracle.dequant_weight_columns(st["gate"].q_weight, st["gate"].scale_col, new_ind)
    gate_want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(qx, st["gate"].q_weight), xs, st["gate"].scale_col,
                                        C=oracle.gemm_fp16(act_out, wc_gate), silu=True)

    # ---- GPU, step by step ----------------------------------------------------------------------------
    x = norm(dev(hidden))
    assert np.array_equal(cache.q_xcache.cpu().numpy(), oracle.rmsnorm_extract_quant(
        hidden, gamma, eps, np.zeros(0, np.int32))[2]), "fused norm -> int8 rows"
    up_got = up(x, cache)
    assert np.array_equal(up.ind.cpu().numpy(), new_ind) and 17 in new_ind and 300 in new_ind
    assert np.array_equal(cache.q_xcache.cpu().numpy(), qx), "rows re-quantised after the outlier columns were zeroed"
    assert rel_err(up_got.cpu().numpy(), up_want) < REL_TOL
    gate_got = gate.forward_without_preconditionFusedSilu(x, cache)
    assert np.array_equal(gate.ind.cpu().numpy(), new_ind)
    assert np.array_equal(gate.weight_cache.cpu().numpy().view(np.uint16), wc_gate.view(np.uint16))
    assert rel_err(gate_got.cpu().numpy(), gate_want) < REL_TOL
    gate_got *= up_got
    h = gate_got.cpu().numpy().copy()
    y_got = down(gate_got, None, True).cpu().numpy()
    y_want = oracle.mixlinear_forward(st["down"], h)          # same input bits: isolates the down projection
    assert rel_err(y_got, y_want) < REL_TOL

    # ---- the packaged block gives the same bits ------------------------------------------------------------
    cache2, norm2, up2, gate2, down2 = build()
    mlp = mixlinear.MixLlamaMLP(gate2, down2, up2, cache2)
    y2 = mlp(norm2(dev(hidden))).cpu().numpy()
    assert np.array_equal(y2.view(np.uint16), y_got.view(np.uint16))


@pytest.mark.parametrize("M,N,K", [(16, 512, 1024), (200, 768, 640), (300, 4096, 512), (2048, 3072, 512)])
def test_fused_silu_mul_epilogue_is_the_two_step_sequence(M, N, K):
    """int8FusedDequantizeSiluMul == int8FusedDequantizeSilu followed by `*= up` (fused/mlp.py:61-63), bit for bit, on
    every GEMM kernel (skinny, two-barrier incl. split-K, ping-pong)."""
    from mixq_tensorrt_llm_amd import mixlib
    g = torch.Generator(device="cpu").manual_seed(M + N)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, generator=g).to("cuda:0")
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, generator=g).to("cuda:0")
    sa = (torch.rand(M, 1, generator=g) * 1e-2 + 1e-3).to(torch.float16).to("cuda:0")
    sb = (torch.rand(1, N, generator=g) * 1e-3 + 1e-4).to(torch.float16).to("cuda:0")
    y = (torch.randn(M, N, generator=g) * 0.5).to(torch.float16).to("cuda:0")
    up = torch.randn(M, N, generator=g).to(torch.float16).to("cuda:0")
    two_step = mixlib.int8FusedDequantizeSilu(a, b, sa, sb, y, M, N, K)
    two_step *= up
    fused = mixlib.int8FusedDequantizeSiluMul(a, b, sa, sb, y, up, M, N, K)
    assert torch.equal(fused, two_step)
    fused0 = mixlib.int8FusedDequantizeSiluMul(a, b, sa, sb, None, up, M, N, K)     # no addend
    ref0 = mixlib.int8FusedDequantizeSilu(a, b, sa, sb, None, M, N, K)
    ref0 *= up
    assert torch.equal(fused0, ref0)


def test_product_reproduces_the_reference_generated_fixture():
    """pflavour_small.npz = outputs of the reference's own linear.py (see tests/golden/gen_golden.py): the product's
    from_linear (bit 8 and 4, on the GPU) and the find_outliers kernel reproduce them bit for bit."""
    import os
    from conftest import GOLDEN
    from mixq_tensorrt_llm_amd import mixlinear
    g = np.load(os.path.join(GOLDEN, "pflavour_small.npz"))
    W = torch.from_numpy(g["W"])
    cache = mixlinear.MixLibCache(inputdim=64, sigma=6, device="cuda:0")
    l8 = mixlinear.MixLinear_GEMM.from_linear(W, bit=8, cache=cache, dev="cuda:0")
    assert np.array_equal(l8.q_weight.cpu().numpy(), g["w8_q_weight"])
    assert np.array_equal(l8.scale_col.cpu().numpy().reshape(-1).view(np.uint16), g["w8_scale_col"].view(np.uint16))
    l4 = mixlinear.MixLinear_GEMM.from_linear(W, bit=4, cache=cache, dev="cuda:0",
                                              layer_scales=torch.from_numpy(g["layer_scales"]), fp_features_num=256)
    assert np.array_equal(l4.ind.cpu().numpy(), g["w4_ind"])
    assert np.array_equal(l4.q_weight.cpu().numpy(), g["w4_q_weight"])
    assert np.array_equal(l4.scale_col.cpu().numpy().reshape(-1).view(np.uint16), g["w4_scale_col"].view(np.uint16))
    assert np.array_equal(l4.weight_cache.cpu().numpy().view(np.uint16), g["w4_weight_cache"].view(np.uint16))
    assert np.array_equal(mixlinear.find_outliers(dev(g["fo_A"]), 6.0).cpu().numpy(), g["fo_ind"])


@pytest.mark.parametrize("M,N,K", [(32, 4096, 4096), (16, 1024, 2048), (24, 512, 3584)])
def test_p_flavour_decode_batches_use_the_fragment_major_image_and_keep_their_bits(M, N, K):
    """Round 3: producers inside the library (fused norm, one-call linear) hand the int8 activation to the skinny GEMM in its
    fragment-major order for decode batches (mixlib.qa_layout).  Norm -> up projection -> gate projection (SiLU * up in the
    epilogue) must give the same bits with the layout on and off (knob 890), and the one-call unfused forward too."""
    from mixq_tensorrt_llm_amd import _lib, mixlib, mixlinear
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    W_up = (torch.randn((N, K), device=dev, generator=g) * 0.02).to(torch.float16)
    W_gate = (torch.randn((N, K), device=dev, generator=g) * 0.02).to(torch.float16)
    scales = torch.rand(K, device=dev, generator=g)
    gamma = (torch.rand(K, device=dev, generator=g) + 0.5).to(torch.float16)
    X = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
    outs = {}
    for knob in (891, 890):
        lib.mixq_debug_set_gemm_variant(knob)
        try:
            cache = mixlinear.MixLibCache(inputdim=64, device=dev)
            up = mixlinear.MixLinear_GEMM.from_linear(W_up, None, bit=8, cache=cache, layer_scales=scales, fp_features_num=128)
            gate = mixlinear.MixLinear_GEMM.from_linear(W_gate, None, bit=8, cache=cache, layer_scales=scales,
                                                        fp_features_num=128)
            up.add_outliers = gate.add_outliers = False
            norm = mixlinear.FasterTransformerRMSNorm(gamma, 1e-6, cache)
            norm.next_layer = up
            xn = norm(X.clone())
            assert cache.q_layout == (mixlib.qa_layout(M, N, K) if knob == 891 else 0)
            if knob == 891:
                assert cache.q_layout == mixlib.QA_FRAGMENT_MAJOR, "these shapes are in the layout's domain"
            y_up = up(xn, cache)
            y_gate = gate.forward_without_preconditionFusedSilu(xn, cache, mul=y_up)
            y_unfused = up(X.clone(), cache, unfused=True)       # one call, two launches; its own producer
            torch.cuda.synchronize()
            outs[knob] = [t.cpu() for t in (y_up, y_gate, y_unfused)]
        finally:
            lib.mixq_debug_set_gemm_variant(891)
    for a, b, name in zip(outs[891], outs[890], ("up", "gate (SiLU * up)", "unfused one-call")):
        assert torch.equal(a, b), f"{name}: fragment-major and row-major images give different bits"
    assert torch.isfinite(outs[891][0].float()).all()
    # the one-call entry with a static outlier set of 128 columns, both layouts, same bits (and the same x_scale / outliers)
    ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
    up2 = mixlinear.MixLinear_GEMM.from_linear(W_up, None, bit=8, cache=mixlinear.MixLibCache(inputdim=64, device=dev),
                                               layer_scales=scales)
    wc = mixlinear.dequant_weight_columns(up2.q_weight, up2.scale_col, ind)
    res = []
    for lay in (0, mixlib.qa_layout(M, N, K)):
        xs = torch.zeros((64, 1), dtype=torch.float16, device=dev)
        x = X.clone()
        out, q_x, outl = mixlib.mixlinear_forward(x, ind, up2.q_weight, up2.scale_col, wc, xs, lay)
        torch.cuda.synchronize()
        res.append((out.cpu(), xs.cpu(), outl.cpu(), x.cpu()))
    assert mixlib.qa_layout(M, N, K) == 1
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)

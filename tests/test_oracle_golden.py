"""CPU: the oracle against the committed golden vectors (generated from the importable reference pieces by
tests/golden/gen_golden.py) and against independent restatements (numpy float16 for the fp16 primitives)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_fp16_conversions_exhaustive(oracle):
    lib = oracle.lib()
    import ctypes
    lib.mixq_oracle_h2f.restype = ctypes.c_float
    lib.mixq_oracle_h2f.argtypes = [ctypes.c_uint16]
    lib.mixq_oracle_f2h.restype = ctypes.c_uint16
    lib.mixq_oracle_f2h.argtypes = [ctypes.c_float]
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    got = np.array([lib.mixq_oracle_h2f(int(b)) for b in bits], dtype=np.float32)
    assert np.array_equal(ref.view(np.uint32)[~np.isnan(ref)], got.view(np.uint32)[~np.isnan(ref)])
    assert np.all(np.isnan(got[np.isnan(ref)]))
    # float -> half RNE: every half value, every midpoint between neighbours, and random floats
    rng = np.random.default_rng(0)
    f = np.concatenate([
        ref[~np.isnan(ref)],
        ((ref[:-1].astype(np.float64) + ref[1:].astype(np.float64)) / 2).astype(np.float32)[~np.isnan(ref[:-1] + ref[1:])],
        rng.standard_normal(20000).astype(np.float32) * 100,
        (rng.standard_normal(20000) * 1e-6).astype(np.float32),
        np.array([65504, 65519.99, 65520, 70000, 1e-8, 2.98e-8, 2.9802322e-8, 5.96e-8, -0.0, np.inf, -np.inf], np.float32),
    ])
    want = f.astype(np.float16).view(np.uint16)
    got = np.array([lib.mixq_oracle_f2h(float(x)) for x in f], dtype=np.uint16)
    assert np.array_equal(want, got)


def test_weight_packing_matches_reference_golden(oracle):
    g = np.load(os.path.join(GOLDEN, "pack_small.npz"))
    W = g["W"]
    sW = oracle.weight_scales(W)
    assert np.array_equal(sW.view(np.uint16), g["weights_scaling_factor"].view(np.uint16))
    # to_quantized_weight (reference function) on the reference's own fp_ind
    Wq = oracle.quantize_weight(W, sW, g["fp_ind"])
    assert np.array_equal(Wq, g["weight_int8"])
    assert np.array_equal(W[:, g["fp_ind"]].view(np.uint16), g["fp_weight"].view(np.uint16))
    assert np.all(Wq[:, g["fp_ind"]] == 0)


def test_outlier_selection_matches_reference_up_to_ties(oracle):
    """torch.sort (unstable) leaves the order inside groups of equal scales unspecified; everything else must agree:
    the selected scale values position by position, and the index set wherever no tie straddles the cut."""
    a = np.load(os.path.join(GOLDEN, "act_scales_llama.npz"))
    for i in range(3):
        s, ref = a[f"scales_{i}"], a[f"fp_ind_{i}"]
        mine = oracle.select_outliers(s)
        assert np.array_equal(s[mine], s[ref])
        cut = s[ref[0]]
        strict_ref = set(ref[s[ref] > cut].tolist())
        strict_mine = set(mine[s[mine] > cut].tolist())
        assert strict_ref == strict_mine
        if np.sum(s == cut) == np.sum(s[ref] == cut):  # no tie across the cut -> identical sets
            assert set(ref.tolist()) == set(mine.tolist())
    g = np.load(os.path.join(GOLDEN, "pack_small.npz"))
    mine = oracle.select_outliers(g["layer_scales"])
    assert np.array_equal(g["layer_scales"][mine], g["layer_scales"][g["fp_ind"]])


def test_quant_rows_against_numpy_float16(oracle):
    """numpy's float16 division is correctly rounded (computed in fp32, rounded once) == the oracle's __hdiv."""
    rng = np.random.default_rng(3)
    A = (rng.standard_normal((9, 512)) * 3).astype(np.float16)
    A[2] = 0                      # zero row: 0/0 -> NaN -> 0, scale 0
    A[3, 5] = np.float16(60000)   # huge amax
    A[4, 7] = np.nan              # NaN element is dropped by __hmax and quantises to 0
    A[5, :] = np.float16(6e-8)    # subnormal amax: scale underflows to 0 -> x/0 = inf -> INT_MAX -> int8 -1
    qA, sA = oracle.quant_rows(A)
    with np.errstate(all="ignore"):
        amax = np.nanmax(np.abs(A.astype(np.float32)), axis=1).astype(np.float16)
        s = (amax / np.float16(127)).astype(np.float16)
        assert np.array_equal(s.view(np.uint16), sA.view(np.uint16))
        q = (A / s[:, None]).astype(np.float16).astype(np.float32)
        want = np.where(np.isnan(q), 0, np.where(np.isinf(q), np.where(q > 0, 2**31 - 1, -2**31), np.rint(q)))
        want = (want.astype(np.int64) & 0xff).astype(np.uint8).view(np.int8)
    assert np.array_equal(qA, want)
    assert np.all(qA[2] == 0) and sA[2] == 0
    assert qA[4, 7] == 0
    assert np.all(qA[5] == -1)
    assert np.abs(qA[[0, 1, 3, 6, 7, 8]].astype(np.int32)).max() == 127


def test_linear_prefill_against_float64(oracle):
    from conftest import make_layer
    A, W, act = make_layer(37, 256, 512, seed=5)
    p = oracle.pack_linear_weights(W, act)
    out, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"],
                                       return_parts=True)
    # int32 accumulators vs exact integer matmul
    acc = parts["qA"].astype(np.int64) @ p["weight"].astype(np.int64).T
    assert np.array_equal(acc, parts["acc"].astype(np.int64))
    # the whole operator approximates A @ W^T (quantisation error only)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    err = np.abs(out.astype(np.float64) - ref)
    assert err.max() < 0.05 * np.abs(ref).max()
    # epilogue identity on the parts
    o2 = oracle.dequant_epilogue(parts["acc"], parts["sA"], p["weights_scaling_factor"], parts["P"])
    assert np.array_equal(o2.view(np.uint16), out.view(np.uint16))


def test_eetq_layout_roundtrip_and_structure(oracle):
    rng = np.random.default_rng(7)
    K, N = 128, 64
    q = rng.integers(-128, 128, size=(K, N), dtype=np.int8)
    img = oracle.eetq_preprocess(q).reshape(-1)
    # closed form of the layout (csrc/decode_kernels.hip header), derived independently of the loop restatement
    P = [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
    for n in range(N):
        for tb in range(K // 64):
            for x in range(64):
                xs = (x & ~3) | ((x & 1) << 1) | ((x & 2) >> 1)
                k = 64 * tb + 16 * (xs // 16) + P[xs % 16]
                assert img[(n // 2) * 2 * K + tb * 128 + (n % 2) * 64 + x] == (int(q[k, n]) + 128)


def test_w8a16_gemv_matches_float64(oracle):
    rng = np.random.default_rng(11)
    K, N, M = 256, 64, 3
    Wt = (rng.standard_normal((K, N)) * 0.02).astype(np.float16)
    q, sc = oracle.eetq_symmetric_quantize(Wt)
    A = rng.standard_normal((M, K)).astype(np.float16)
    out = oracle.w8a16_gemv(A, q, sc)
    ref = A.astype(np.float64) @ (q.astype(np.float64) * sc.astype(np.float64)[None, :])
    assert np.allclose(out.astype(np.float64), ref, rtol=2e-2, atol=2e-3)
    assert np.abs(q).max() <= 128 and np.abs(q.astype(np.int32)).max() >= 126


def test_dynamic_outlier_helpers_against_the_reference_torch_expressions(oracle):
    """linear.py:155-161 and :207-209 are plain torch expressions; evaluate them verbatim with CPU torch and compare the
    oracle's numpy restatement (pins `find_outliers` / `dequant_weight_columns`)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(21)
    A = rng.standard_normal((50, 512)).astype(np.float16).clip(-5, 5)
    A[rng.integers(0, 50, 40), rng.integers(0, 512, 40)] = np.float16(8.25)
    A[3, 10] = np.nan
    A[4, 11] = np.inf
    A[5, 12] = np.float16(6.0)
    sigma = torch.zeros((1, 1), dtype=torch.float16)
    sigma[0] = 6
    At = torch.from_numpy(A)
    want = torch.unique(torch.where(At.abs() > sigma)[1]).to(torch.int32).numpy()
    got = oracle.find_outliers(A, 6.0)
    assert np.array_equal(got, want)
    assert 11 in got and 12 not in got and 10 not in got   # inf is an outlier; == sigma and NaN are not
    q = torch.from_numpy(rng.integers(-128, 128, (64, 512), dtype=np.int8))
    scale_col = torch.from_numpy((rng.random((1, 64)) * 1e-2 + 1e-4).astype(np.float16))
    ind = torch.from_numpy(want[:17].astype(np.int64))
    want_wc = (q[:, ind].to(torch.float16) * scale_col.T).numpy()
    got_wc = oracle.dequant_weight_columns(q.numpy(), scale_col.numpy(), want[:17])
    assert np.array_equal(got_wc.view(np.uint16), want_wc.view(np.uint16))


def test_int4_packing_and_weight_quant_against_torch_expressions(oracle):
    """linear.py:11-17 (pack_to_i4) and :121-143 (bit = 4 `from_linear`) are torch expressions: evaluate them with CPU
    torch and compare the oracle's numpy restatement."""
    torch = pytest.importorskip("torch")
    from mixq_tensorrt_llm_amd import mixlinear
    rng = np.random.default_rng(8)
    q = rng.integers(-8, 8, (16, 64), dtype=np.int8)
    assert np.array_equal(mixlinear.pack_to_i4(torch.from_numpy(q)).numpy(), oracle.pack_i4(q))
    assert np.array_equal(oracle.unpack_i4(oracle.pack_i4(q)), q)
    W = (rng.standard_normal((32, 128)) * 0.02).astype(np.float16)
    scales = np.abs(rng.standard_normal(128)).astype(np.float32)
    fp = 16
    # the reference's expressions, verbatim
    ind = torch.sort(torch.from_numpy(scales))[1][-fp:]
    tmp = torch.from_numpy(W.copy())
    weight_cache = tmp[:, ind].clone()
    tmp[:, ind] = 0
    scale = (torch.max(torch.abs(tmp), dim=1)[0].unsqueeze(1) / (10)).to(torch.float16).reshape((1, 32))
    tmp /= scale.T
    tmp = torch.clamp(tmp.round(), -8, 7)
    packed = mixlinear.pack_to_i4(tmp.to(torch.int8))
    qp, sc, oind, wc = oracle.mixlinear4_from_linear(W, scales, fp)
    assert np.array_equal(oind, ind.numpy().astype(np.int32))
    assert np.array_equal(sc.view(np.uint16), scale.numpy().reshape(-1).view(np.uint16))
    assert np.array_equal(qp, packed.numpy())
    assert np.array_equal(wc.view(np.uint16), weight_cache.numpy().view(np.uint16))


def test_pflavour_fixture_from_the_reference_linear_py(oracle):
    """tests/golden/pflavour_small.npz holds outputs of the reference's own MixQ/src/mixquant/modules/linear.py
    (pack_to_i4, MixLinear_GEMM.from_linear for bit 8 and 4, FindOutliers), captured by gen_golden.py.  The oracle's
    restatements and the product's host-side `from_linear` / `pack_to_i4` must reproduce them exactly."""
    torch = pytest.importorskip("torch")
    from mixq_tensorrt_llm_amd import mixlinear
    g = np.load(os.path.join(GOLDEN, "pflavour_small.npz"))
    # pack_to_i4
    assert np.array_equal(oracle.pack_i4(g["i4_in"]), g["i4_packed"])
    assert np.array_equal(mixlinear.pack_to_i4(torch.from_numpy(g["i4_in"])).numpy(), g["i4_packed"])
    assert np.array_equal(oracle.unpack_i4(g["i4_packed"]), g["i4_in"])
    # FindOutliers
    assert np.array_equal(oracle.find_outliers(g["fo_A"], 6.0), g["fo_ind"])
    # from_linear, bit = 4 (oracle restatement)
    qp, sc, ind, wc = oracle.mixlinear4_from_linear(g["W"], g["layer_scales"], 256)
    assert np.array_equal(np.sort(ind), np.sort(g["w4_ind"]))      # (tie order of torch.sort is unspecified)
    assert np.array_equal(sc.view(np.uint16), g["w4_scale_col"].view(np.uint16))
    assert np.array_equal(qp, g["w4_q_weight"])
    order = {c: i for i, c in enumerate(g["w4_ind"])}
    assert np.array_equal(wc[:, [list(ind).index(c) for c in g["w4_ind"]]].view(np.uint16),
                          g["w4_weight_cache"].view(np.uint16)) and len(order) == 256
    # from_linear, bit = 8: scale = fp16(max|w| / 127), q = round(w / scale), no clamp
    W = g["W"]
    sc8 = (np.abs(W).max(axis=1) / np.float16(127)).astype(np.float16)
    assert np.array_equal(sc8.view(np.uint16), g["w8_scale_col"].view(np.uint16))
    assert np.array_equal(np.rint((W / sc8[:, None]).astype(np.float16).astype(np.float64)).astype(np.int8), g["w8_q_weight"])


def test_decode_oracle_in_the_cuda_kernels_own_order(oracle):
    """w8a16_gemv_reforder restates weightOnlyBatchedGemv/kernel.h:300-470 for Int8b per-channel (NPerBlock 2, block 256):
    fp16 FMA chains per thread, fp32 butterflies.  Checked here: (i) its fp16 FMA primitive against an exact rational
    evaluation, (ii) on integer-valued data, where every partial sum is exact in fp16, it equals the plain dot product,
    (iii) on ordinary data it sits within ~1e-3 of exact -- the size of the reference's own fp16-accumulation error."""
    import ctypes
    from fractions import Fraction
    rng = np.random.default_rng(3)
    # (i) hfma_exact is static: exercise it through a K = 64 GEMV with one non-zero product per thread chain
    K, N = 64, 4
    for trial in range(200):
        A = np.zeros((1, K), np.float16)
        Wq = np.zeros((K, N), np.int8)
        k0 = int(rng.integers(0, K - 1))
        k1 = k0 + 1 if (k0 % 16) != 15 else k0 - 1          # same 16-run: same thread, consecutive FMAs
        a0, a1 = rng.standard_normal(2).astype(np.float16) * np.float16(2.0 ** int(rng.integers(-8, 8)))
        A[0, k0], A[0, k1] = a0, a1
        Wq[k0, 0], Wq[k1, 0] = rng.integers(-128, 128, 2)
        sc = np.array([rng.random() * 1e-2 + 1e-4] * N, np.float16)
        got = oracle.w8a16_gemv_reforder(A, Wq, sc)[0, 0]
        w = [np.float16(np.float32(Wq[k, 0]) * np.float32(sc[0])) for k in (min(k0, k1), max(k0, k1))]
        a = [A[0, min(k0, k1)], A[0, max(k0, k1)]]
        first = np.float16(np.float32(w[0]) * np.float32(a[0]))              # fma(w, a, 0): product exact in fp32
        exact = Fraction(float(w[1])) * Fraction(float(a[1])) + Fraction(float(first))
        # RNE of the exact rational to fp16
        cands = [np.float16(float(exact)), np.nextafter(np.float16(float(exact)), np.float16(np.inf)),
                 np.nextafter(np.float16(float(exact)), np.float16(-np.inf))]
        best = min(cands, key=lambda c: (abs(Fraction(float(c)) - exact), int(c.view(np.uint16)) & 1))
        assert got.view(np.uint16) == np.float16(best).view(np.uint16) or (got == 0 and best == 0)
    # (ii) exact regime
    K, N, M = 256, 8, 3
    A = rng.integers(-2, 3, size=(M, K)).astype(np.float16)
    Wq = rng.integers(-3, 4, size=(K, N)).astype(np.int8)
    sc = np.ones(N, np.float16)
    want = (A.astype(np.int64) @ Wq.astype(np.int64)).astype(np.float16)
    assert np.array_equal(oracle.w8a16_gemv_reforder(A, Wq, sc), want)
    assert np.array_equal(oracle.w8a16_gemv(A, Wq, sc), want)
    # (iii) ordinary data, K with a ragged last trip (11008 = 5.375 trips of 2048)
    K, N, M = 11008, 16, 4
    A = rng.standard_normal((M, K)).astype(np.float16)
    Wq = rng.integers(-128, 128, size=(K, N), dtype=np.int8)
    sc = (rng.random(N) * 1e-3 + 1e-4).astype(np.float16)
    exact = A.astype(np.float64) @ (Wq.astype(np.float64) * sc.astype(np.float64))
    err = lambda x: np.abs(x.astype(np.float64) - exact).max() / np.abs(exact).max()  # noqa: E731
    assert err(oracle.w8a16_gemv(A, Wq, sc)) < 1e-3
    assert err(oracle.w8a16_gemv_reforder(A, Wq, sc)) < 3e-3


def test_dequantization_silu_oracle_against_numpy(oracle):
    rng = np.random.default_rng(5)
    M, N = 9, 24
    x = rng.integers(-50000, 50000, size=(M, N)).astype(np.int32)
    sa = (rng.random(M) * 0.05).astype(np.float16)
    sw = (rng.random(N) * 1e-3).astype(np.float16)
    y = rng.standard_normal((M, N)).astype(np.float16)
    got = oracle.dequantization_silu(x, sa, sw, y)
    v = (x.astype(np.float64) * sa.astype(np.float64)[:, None]) * sw.astype(np.float64)[None, :] + y.astype(np.float64)
    want = v / (1 + np.exp(-v))
    assert np.abs(got.astype(np.float64) - want).max() <= 1e-3 * np.abs(want).max()


def test_top_level_mixlib_module_name():
    """`import mixlib` (MixQ/src/mixquant/modules/linear.py:5) resolves to the MI355X op surface with every op name the
    reference's int8_mix callers use (pybind_mix.cpp:256-335)."""
    import importlib
    m = importlib.import_module("mixlib")
    for name in ("FindRowScale", "ExtractOutliersAndSetToZeros", "int8FusedDequantize", "int8FusedDequantizeSilu", "gemm",
                 "dequantizeInt8", "dequantizeInt8Silu", "Int8quantize", "FindRowScaleFusedExtracOutliers", "int_to_half",
                 "int8_matrix_to_half", "int_matrix_to_half", "layernorm_forward_cuda_extract_outliers"):
        assert callable(getattr(m, name)), name


def test_top_level_eetq_module_name(oracle):
    """`from EETQ import quant_weights, preprocess_weights, w8_a16_gemm` (linear.py:9, model_config_utils.py:434): same
    return convention as the extension (eetpy.cpp:14-17) and the oracle's restatement of symmetric_quantize +
    preprocess_weights (cutlass_preprocessors.cc:497-660) byte for byte."""
    import torch
    from EETQ import preprocess_weights, quant_weights, w8_a16_gemm
    assert callable(w8_a16_gemm)
    rng = np.random.default_rng(8)
    Wt = (rng.standard_normal((128, 64)) * 0.02).astype(np.float16)          # [K, N] = W^T as the callers pass it
    processed, scales = quant_weights(torch.from_numpy(Wt), torch.int8, False)
    assert processed.dtype == torch.int8 and tuple(processed.shape) == (128, 64) and scales.dtype == torch.float16
    q_un, sc = oracle.eetq_symmetric_quantize(Wt)
    assert np.array_equal(scales.numpy().view(np.uint16), sc.view(np.uint16))
    assert np.array_equal(processed.numpy().view(np.uint8), oracle.eetq_preprocess(q_un))
    un, processed2, _ = quant_weights(torch.from_numpy(Wt), torch.int8, True)
    assert np.array_equal(un.numpy(), q_un) and torch.equal(processed2, processed)
    assert np.array_equal(preprocess_weights(torch.from_numpy(q_un)).numpy().view(np.uint8), oracle.eetq_preprocess(q_un))

def test_row_scale_equals_the_reference_torch_expression(oracle):
    """MixQ/src/benchmark/scale_benchmark.py:16 (and fuse_scale_benchmerk.py:29, linear.py's callers) hold the row scale as a TORCH
    expression next to the kernel that computes it: ``torch.max(x.abs(), dim=1)[0] / 127.0``.  Evaluated verbatim with CPU torch in
    fp16 (the script's shape and seed, plus rows that stress the rounding), it must equal `oracle.quant_rows`' sA bit for bit: a
    reference-held expression pinning one quantity of the otherwise unpinnable device half."""
    torch = pytest.importorskip("torch")
    torch.manual_seed(0)
    M, N = 32, 12288
    inputs = torch.randn((M, N), dtype=torch.float16)
    rng = np.random.default_rng(9)
    extra = (rng.standard_normal((64, N)) * np.exp(rng.uniform(-8, 8, (64, 1)))).astype(np.float16)
    extra[0] = 0
    extra[1, 7] = np.float16(65504)
    extra[2, :] = np.float16(6e-8)
    x = torch.cat([inputs, torch.from_numpy(extra)])
    x_scale = torch.max(x.abs(), dim=1)[0] / 127.0
    assert x_scale.dtype == torch.float16
    _, sA = oracle.quant_rows(x.numpy())
    assert np.array_equal(x_scale.numpy().view(np.uint16), sA.view(np.uint16))
    # the script's own check (:38): the sum of the differences is exactly zero
    assert float(torch.sum(x_scale - torch.from_numpy(sA))) == 0.0


def _hdiv_bound_measurements(oracle):
    from conftest import make_layer
    out = {}
    A, _, _ = make_layer(32, 4096, 4096, seed=0)          # BASELINE configs[0]: one 4096 x 4096 linear, 32 rows
    q0, s0 = oracle.quant_rows(A)
    allpos = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16)
    rng = np.random.default_rng(42)
    caps = np.concatenate([np.float16([1e-7, 6.1e-5, 1e-3, 0.0999, 1.0, 127.0, 254.0, 333.3, 1000.0, 65504.0]),
                           np.exp(rng.uniform(np.log(1e-4), np.log(6e4), 38)).astype(np.float16)])
    grids = [np.stack([allpos[allpos <= c], -allpos[allpos <= c]]) for c in caps if np.any(allpos <= c)]
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(1_000_000) * 3).astype(np.float16)
    b = (np.abs(rng.standard_normal(1_000_000)) * 0.05 + 0.01).astype(np.float16)
    ieee = (a.astype(np.float32) / b.astype(np.float32)).astype(np.float16)
    for u in (-1, 0, 1):
        q, s = oracle.quant_rows_cuda_hdiv(A, u)
        cell = {"config0_elements": int(q.size), "config0_qA_differ": int((q != q0).sum()),
                "config0_sA_differ": int((s.view(np.uint16) != s0.view(np.uint16)).sum())}
        n = d = 0
        for g in grids:
            qg0, _ = oracle.quant_rows(g)
            qg, _ = oracle.quant_rows_cuda_hdiv(g, u)
            n += g.size
            d += int((qg != qg0).sum())
        cell["every_fp16_x_48_scales_elements"] = n
        cell["every_fp16_x_48_scales_qA_differ"] = d
        h = oracle.hdiv_cuda(a, b, u)
        cell["fp16_quotients_sampled"] = int(a.size)
        cell["fp16_quotients_differ"] = int((h.view(np.uint16) != ieee.view(np.uint16)).sum())
        out[f"rcp_{u:+d}_ulp"] = cell
    return out


def test_cuda_hdiv_deviation_is_the_committed_measured_bound(oracle):
    """CUDA's device `__hdiv` multiplies by `rcp.approx.ftz.f32` (1 ulp) where the oracle divides (IEEE).  `oracle.quant_rows_cuda_hdiv`
    evaluates the reference's quantiser with that division and the reciprocal at -1 / 0 / +1 fp32 ulp -- every conforming rcp lies in
    between.  The counts of differing values are a committed fixture (tests/golden/hdiv_rcp_bound.json; regenerate with
    ``python tests/test_oracle_golden.py``): about 1e-5 of fp16 QUOTIENTS move by one fp16 ulp when the reciprocal is off by one
    ulp, and none of them crosses an integer rounding boundary -- no int8 value of the configs[0] input, nor of every finite fp16 x
    against 48 scales, changes.  DESIGN.md §7 quotes these numbers instead of a recalled "~1e-5"."""
    import json
    want = json.load(open(os.path.join(GOLDEN, "hdiv_rcp_bound.json")))
    got = _hdiv_bound_measurements(oracle)
    assert got == want["counts"], got
    for cell in got.values():   # the bound the parity suite relies on
        assert cell["config0_qA_differ"] == 0 and cell["config0_sA_differ"] == 0 and cell["every_fp16_x_48_scales_qA_differ"] == 0
    assert got["rcp_+0_ulp"]["fp16_quotients_differ"] == 0   # a correctly rounded reciprocal reproduces the IEEE quotient on the sample


if __name__ == "__main__":   # regenerate the fixture
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle as _o
    _o.build()
    json.dump({"what": "values that differ between the reference's quantiser evaluated with CUDA's __hdiv (fp16(fa * rcp(fb)), rcp at "
                       "-1 / 0 / +1 fp32 ulp from the correctly rounded reciprocal) and with the IEEE quotient (oracle.quant_rows)",
               "counts": _hdiv_bound_measurements(_o)}, open(os.path.join(GOLDEN, "hdiv_rcp_bound.json"), "w"), indent=1)

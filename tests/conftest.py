import os
import sys

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # the tests force kernel forms through mixq_debug_*: opt in before the library loads

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_tie_order_applies():
    """fp_ind in the goldens carries the tie order of ONE torch build's unstable CPU sort (the reference's very call).  Under another
    build only the column set is pinned (tests/golden/META.json)."""
    import json
    try:
        import torch
        return json.load(open(os.path.join(GOLDEN, "META.json")))["torch"] == torch.__version__
    except Exception:  # noqa: BLE001
        return False


def assert_same_outlier_columns(mine, golden, what=""):
    """Exact order under the goldens' torch build, the same column set otherwise."""
    mine, golden = np.asarray(mine), np.asarray(golden)
    if golden_tie_order_applies():
        np.testing.assert_array_equal(mine, golden, err_msg=what)
    else:
        assert sorted(mine.tolist()) == sorted(golden.tolist()), what


def make_layer(M, N, K, O=128, seed=0, outlier_gain=20.0):
    """Synthetic MixQ linear in the shape SURVEY.md §8d prescribes: A ~ N(0,1) with the outlier columns scaled up,
    W ~ N(0, 0.02^2), activation scales |N(0,1)| (the real Llama vectors are used where K matches, see test_golden)."""
    rng = np.random.default_rng(seed)
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    A = rng.standard_normal((M, K)).astype(np.float32)
    ind = np.argsort(act, kind="stable")[-O:].astype(np.int32)
    A[:, ind] *= outlier_gain
    return A.astype(np.float16), W, act


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


# ---------------------------------------------------------------------------------------------------------------
# Element-wise parity bounds ("1e-3 relative" normalised by the output's maximum lets small-magnitude outputs
# be far off).  Everything on the hot path is exact except two things: fp32 accumulation order (unspecified in the
# reference too: cuBLAS / CUTLASS) and the fp16 rounding of a sum that straddles a rounding boundary.  So, per ELEMENT:
#   |got - want| <= ulp16(want)  [one fp16 rounding step, <= 2^-10 |want| < 1e-3 |want|]  +  order slack
# where the order slack of the prefill operator is one ulp of the fp16-rounded outlier product P16 that enters the dequant
# FMA (the only order-dependent quantity: the int8 part is exact), and that of the fp16 x int8 GEMV / GEMM is GAMMA x
# sum_k |a_k w_k| (fp32 accumulation of K products in two different orders).
def ulp16(x):
    """Spacing of fp16 numbers at magnitude |x| (2^-24 below the normal range)."""
    a = np.abs(np.asarray(x, np.float64))
    _, e = np.frexp(np.maximum(a, 2.0 ** -14))     # a = m 2^e, m in [0.5, 1)
    return np.ldexp(1.0, e - 11)


def elementwise_violations(got, want, slack):
    g, w = got.astype(np.float64), want.astype(np.float64)
    bound = ulp16(np.maximum(np.abs(g), np.abs(w))) + slack
    with np.errstate(invalid="ignore"):
        d = np.abs(g - w)
        bad = ~((d <= bound) | (np.isnan(g) & np.isnan(w)) | (g == w))
    return bad, d, bound


def assert_elementwise(got, want, slack, what=""):
    bad, d, bound = elementwise_violations(got, want, slack)
    if bad.any():
        i = tuple(np.argwhere(bad)[0])
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} outputs outside the element-wise bound; first at {i}: "
                             f"got {got[i]!r} want {want[i]!r} |diff| {d[i]:.3e} bound {bound[i]:.3e}")


def ulp_histogram(got, want):
    """Fractions of outputs that differ from the oracle by 0, <= 1, <= 2 and > 2 fp16 ulps (of the larger magnitude)."""
    g, w = got.astype(np.float64), want.astype(np.float64)
    with np.errstate(invalid="ignore"):
        u = np.abs(g - w) / ulp16(np.maximum(np.abs(g), np.abs(w)))
    u = u[np.isfinite(u)]
    return {"0": float(np.mean(u == 0)), "<=1": float(np.mean(u <= 1)), "<=2": float(np.mean(u <= 2)),
            ">2": float(np.mean(u > 2)), "max": float(u.max()) if u.size else 0.0}


W8A16_GAMMA = 4e-6   # fp32 accumulation of <= 28672 fp16 x fp16 products in two different orders, relative to sum |a w|


SIDE_GAMMA = 2.0 ** -22   # fp32 accumulation of the 128 outlier products in two different orders, relative to sum |a w|


def prefill_slack(parts, A=None, p=None):
    """Order slack of the W8A8O16 operator (oracle.linear_prefill parts): one ulp of the fp16-rounded outlier product P16 -- the
    only order-dependent quantity, the int8 part being exact -- plus, when the operands are given, the fp32 accumulation error
    of that product itself (SIDE_GAMMA x sum_j |fpA fpW|): where the 128 terms cancel to a tiny P, two summation orders can
    differ by several ulps OF THAT TINY P (seen at 2048 x 12288 x 512: 2 of 25 M outputs, 3 ulps of a 1.6e-4 result)."""
    slack = ulp16(parts["P"])
    if A is not None and p is not None:
        fa = np.abs(A[:, p["fp_ind"]].astype(np.float32))
        fw = np.abs(p["fp_weight"].astype(np.float32))
        slack = slack + SIDE_GAMMA * (fa @ fw.T).astype(np.float64)
    return slack


def assert_prefill_parity(oracle, got, A, p, what="", bias=None):
    """The W8A8O16 operator against the oracle, BOTH ways: north_star's 1e-3 of the output's maximum, and element by element
    within one fp16 rounding step + the order slack of the outlier product (prefill_slack).  ``bias``: the fp16 addition the
    caller applies after the operator (plugin.py:158-160) -- one more rounding, one more ulp of the un-biased value."""
    want, parts = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"], return_parts=True)
    slack = prefill_slack(parts, A, p)
    if bias is not None:
        slack = slack + ulp16(want)
        want = (want.astype(np.float16) + np.asarray(bias, np.float16)[None, :]).astype(np.float16)
    g, w = got.astype(np.float64), want.astype(np.float64)
    rel = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
    assert rel < 1e-3, f"{what}: max-normalised error {rel:.3e}"
    assert_elementwise(got, want, slack, what)
    return want


def w8a16_slack(A, q_rm, scale):
    """Order slack of the fp16 x int8 paths: GAMMA x sum_k |a_k| |fp16(q_k s)|."""
    wd = np.abs((q_rm.astype(np.float32) * scale.astype(np.float32)[None, :]).astype(np.float16).astype(np.float32))
    return W8A16_GAMMA * (np.abs(A.astype(np.float32)) @ wd).astype(np.float64)

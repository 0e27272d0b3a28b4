"""numpy front-end of oracle/mixq_oracle.c (ctypes).  TEST INFRASTRUCTURE ONLY -- see __init__.py.

All fp16 tensors are numpy ``float16`` arrays (bit patterns are handed to C as uint16).  Function
names follow the reference functions they restate; the C source cites file:line for each.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmixq_oracle.so")

__all__ = [
    "build", "lib", "num_threads", "usable_cpus", "quant_rows", "extract_outliers", "gemm_s8s8s32", "gemm_fp16",
    "dequant_epilogue", "dequantization", "dequantization_silu", "w8a16_gemv_reforder", "linear_prefill", "weight_scales", "quantize_weight",
    "select_outliers", "pack_linear_weights", "eetq_symmetric_quantize", "eetq_preprocess", "w8a16_gemv",
    "int_to_half", "int8_matrix_to_half", "rmsnorm_extract_quant", "find_outliers", "dequant_weight_columns",
    "MixLinearState", "mixlinear_forward", "quant4_rows", "unpack_i4", "pack_i4", "mixlinear4_from_linear",
    "mixlinear4_forward", "hdiv_cuda", "quant_rows_cuda_hdiv",
]


def build(force=False):
    """Compile libmixq_oracle.so with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "mixq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        # OpenMP sizes its team from the CPUs it can SEE; a container's CPU quota is not one of them (the GPU boxes of this pool
        # show 256 logical CPUs under a cgroup quota of 16: 128 threads ran the same GEMM 20 % SLOWER than 16,
        # tools/cpu_probe.py).  Cap the team at what the process may actually use, before libgomp reads the environment.
        usable = usable_cpus()
        try:
            asked = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
        except ValueError:
            asked = 0
        if asked <= 0 or asked > usable:
            os.environ["OMP_NUM_THREADS"] = str(usable)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.mixq_oracle_num_threads.restype = ctypes.c_int
        if _lib.mixq_oracle_num_threads() > usable:   # (an OpenMP runtime loaded before us -- torch's -- already read the environment)
            _lib.mixq_oracle_set_num_threads(usable)
    return _lib


def usable_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup v2 / v1 CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def num_threads():
    return int(lib().mixq_oracle_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _h(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.float16, a.dtype
    return a


def _i64(v):
    return ctypes.c_int64(int(v))


def quant_rows(A, zero_cols=None):
    """kernel/i8gemm.cu:66-107 -> (qA int8 [M,K], sA fp16 [M])."""
    A = _h(A)
    M, K = A.shape
    qA = np.empty((M, K), np.int8)
    sA = np.empty((M,), np.float16)
    zc = None if zero_cols is None else np.ascontiguousarray(zero_cols, np.int32)
    lib().mixq_oracle_quant_rows(_i64(M), _i64(K), _p(A), _p(qA), _p(sA), _p(zc),
                                 ctypes.c_int(0 if zc is None else zc.size))
    return qA, sA


def hdiv_cuda(a, b, rcp_ulps=0):
    """CUDA's DEVICE `__hdiv` as cuda_fp16.hpp writes it (the form kernel/i8gemm.cu:99-104 really executes; `quant_rows` uses the IEEE
    quotient): v = fp16(fa * rcp.approx.ftz.f32(fb)), with the fma correction step for results below the half 0x008F.
    `rcp.approx` is specified to 1 ulp, not bit for bit: ``rcp_ulps`` in {-1, 0, +1} evaluates the reciprocal at that many fp32 ulps from
    the correctly rounded one, which brackets every conforming implementation.  numpy, vectorised; a, b float16 arrays (broadcast)."""
    a = np.asarray(a, np.float16)
    b = np.asarray(b, np.float16)
    fa, fb = np.broadcast_arrays(a.astype(np.float32), b.astype(np.float32))
    with np.errstate(all="ignore"):
        rcp = (np.float32(1.0) / fb).astype(np.float32)
        if rcp_ulps:
            toward = np.where(rcp_ulps > 0, np.float32(np.inf), np.float32(-np.inf)) * np.where(rcp < 0, np.float32(-1), np.float32(1))
            shifted = rcp
            for _ in range(abs(int(rcp_ulps))):
                shifted = np.nextafter(shifted, toward.astype(np.float32)).astype(np.float32)
            rcp = np.where(np.isfinite(rcp) & (rcp != 0), shifted, rcp)
        tiny = np.float32(2.0 ** -126)
        rcp = np.where(np.abs(rcp) < tiny, np.copysign(np.float32(0), rcp), rcp).astype(np.float32)   # .ftz on the result
        fv = (rcp * fa).astype(np.float32)
        v = fv.astype(np.float16)
        den = np.array([0x008F], np.uint16).view(np.float16)[0]
        fix = (np.abs(v) < den) & (np.abs(v) > 0)
        if np.any(fix):   # (results this small quantise to 0 either way; kept for fidelity)
            err = (-fb.astype(np.float64) * fv.astype(np.float64) + fa.astype(np.float64)).astype(np.float32)            # __fmaf_rn(-fb, fv, fa)
            fv2 = (rcp.astype(np.float64) * err.astype(np.float64) + fv.astype(np.float64)).astype(np.float32)           # __fmaf_rn(rcp, err, fv)
            v = np.where(fix, fv2.astype(np.float16), v)
    return v


def quant_rows_cuda_hdiv(A, rcp_ulps=0):
    """kernel/i8gemm.cu:66-107 with BOTH divisions (:99 the scale, :104 the elements) through `hdiv_cuda` -> (qA int8, sA fp16).
    With the IEEE quotient this is `quant_rows`; the difference between the two is the known deviation DESIGN.md §7 bounds."""
    A = _h(A)
    with np.errstate(all="ignore"):
        amax = np.fmax.reduce(np.abs(A.astype(np.float32)), axis=1, initial=0).astype(np.float16)   # __hmax drops NaN operands
        sA = hdiv_cuda(amax, np.float16(127.0), rcp_ulps)
        q = hdiv_cuda(A, sA[:, None], rcp_ulps).astype(np.float32)
        r = np.where(np.isnan(q), 0, np.where(np.isinf(q), np.where(q > 0, 2.0 ** 31 - 1, -2.0 ** 31), np.rint(q)))   # __half2int_rn
    qA = (r.astype(np.int64) & 0xff).astype(np.uint8).view(np.int8)
    return qA, sA.astype(np.float16)


def extract_outliers(A, ind, set_zero=False):
    """kernel/i8gemm.cu:198-244 (set_zero=False) / cult.cu:1406-1432 (set_zero=True, mutates A)."""
    assert A.dtype == np.float16 and A.flags.c_contiguous
    ind = np.ascontiguousarray(ind, np.int32)
    M, K = A.shape
    fpA = np.empty((M, ind.size), np.float16)
    lib().mixq_oracle_extract_outliers(_i64(M), _i64(K), _p(A), _p(fpA), _p(ind), ctypes.c_int(ind.size),
                                       ctypes.c_int(1 if set_zero else 0))
    return fpA


def gemm_s8s8s32(qA, W):
    qA = np.ascontiguousarray(qA, np.int8)
    W = np.ascontiguousarray(W, np.int8)
    M, K = qA.shape
    N = W.shape[0]
    assert W.shape[1] == K
    acc = np.empty((M, N), np.int32)
    lib().mixq_oracle_gemm_s8s8s32(_i64(M), _i64(N), _i64(K), _p(qA), _p(W), _p(acc))
    return acc


def gemm_fp16(fpA, fpW):
    fpA, fpW = _h(fpA), _h(fpW)
    M, O = fpA.shape
    N = fpW.shape[0]
    P = np.empty((M, N), np.float16)
    lib().mixq_oracle_gemm_fp16(_i64(M), _i64(N), ctypes.c_int(O), _p(fpA), _p(fpW), _p(P))
    return P


def dequant_epilogue(acc, sA, sW, C=None, silu=False):
    acc = np.ascontiguousarray(acc, np.int32)
    M, N = acc.shape
    sA, sW = _h(sA).reshape(-1), _h(sW).reshape(-1)
    assert sA.size == M and sW.size == N
    C = None if C is None else _h(C)
    D = np.empty((M, N), np.float16)
    lib().mixq_oracle_dequant_epilogue(_i64(M), _i64(N), _p(acc), _p(sA), _p(sW), _p(C), _p(D),
                                       ctypes.c_int(1 if silu else 0))
    return D


def dequantization(x, sA, sW, out):
    """kernel/i8gemm.cu:258-279; returns a new array (out is the addend)."""
    x = np.ascontiguousarray(x, np.int32)
    M, N = x.shape
    res = _h(out).copy()
    lib().mixq_oracle_dequantization(_i64(M), _i64(N), _p(x), _p(_h(sA).reshape(-1)), _p(_h(sW).reshape(-1)), _p(res))
    return res


def linear_prefill(A, W, sW, fpW, ind, return_parts=False):
    """TsinghuaMixQPlugin.cpp:518-532.  A fp16 [M,K], W int8 [N,K], sW fp16 [N], fpW fp16 [N,O], ind int32 [O]."""
    A, sW, fpW = _h(A), _h(sW).reshape(-1), _h(fpW)
    W = np.ascontiguousarray(W, np.int8)
    ind = np.ascontiguousarray(ind, np.int32)
    M, K = A.shape
    N, O = fpW.shape
    assert W.shape == (N, K) and sW.size == N and ind.size == O
    Out = np.empty((M, N), np.float16)
    qA = np.empty((M, K), np.int8)
    sA = np.empty((M,), np.float16)
    acc = np.empty((M, N), np.int32)
    P = np.empty((M, N), np.float16)
    lib().mixq_oracle_linear_prefill(_i64(M), _i64(N), _i64(K), ctypes.c_int(O), _p(A), _p(W), _p(sW), _p(fpW),
                                     _p(ind), _p(Out), _p(qA), _p(sA), _p(acc), _p(P))
    if return_parts:
        return Out, dict(qA=qA, sA=sA, acc=acc, P=P)
    return Out


def weight_scales(W):
    W = _h(W)
    N, K = W.shape
    sW = np.empty((N,), np.float16)
    lib().mixq_oracle_weight_scales(_i64(N), _i64(K), _p(W), _p(sW))
    return sW


def quantize_weight(W, sW, zero_cols=()):
    W, sW = _h(W), _h(sW).reshape(-1)
    N, K = W.shape
    zc = np.ascontiguousarray(zero_cols, np.int32)
    Wq = np.empty((N, K), np.int8)
    lib().mixq_oracle_quantize_weight(_i64(N), _i64(K), _p(W), _p(sW), _p(zc), ctypes.c_int(zc.size), _p(Wq))
    return Wq


def select_outliers(act_scale, num=128):
    s = np.ascontiguousarray(act_scale, np.float32)
    ind = np.empty((num,), np.int32)
    lib().mixq_oracle_select_outliers(_i64(s.size), _p(s), ctypes.c_int(num), _p(ind))
    return ind


def int_to_half(ind_i32):
    """mixlib.int_to_half (quantkernel/mix_cuda/cult.cu:3046-3085): bit reinterpretation int32[n] -> fp16[2n]."""
    return np.ascontiguousarray(ind_i32, np.int32).view(np.float16)


def int8_matrix_to_half(w_i8):
    """mixlib.int8_matrix_to_half: int8 [R,C] -> fp16 [R,C/2], same bytes."""
    w = np.ascontiguousarray(w_i8)
    assert w.dtype in (np.int8, np.uint8)
    return w.view(np.float16)


def eetq_symmetric_quantize(Wt):
    """cutlass_preprocessors.cc:573-660 on Wt fp16 [K,N] -> (unprocessed int8 [K,N], scales fp16 [N])."""
    Wt = _h(Wt)
    K, N = Wt.shape
    q = np.empty((K, N), np.int8)
    sc = np.empty((N,), np.float16)
    lib().mixq_oracle_eetq_symmetric_quantize(_i64(K), _i64(N), _p(Wt), _p(q), _p(sc))
    return q, sc


def eetq_preprocess(q_rm):
    """cutlass_preprocessors.cc:497-534 on row-major int8 [K,N] -> uint8 [K,N] (interleaved layout)."""
    q = np.ascontiguousarray(q_rm, np.int8)
    K, N = q.shape
    assert K % 64 == 0 and N % 2 == 0
    out = np.empty((K, N), np.uint8)
    lib().mixq_oracle_eetq_preprocess(_i64(K), _i64(N), _p(q), _p(out))
    return out


def w8a16_gemv(A, Wq_rm, scale):
    A, scale = _h(A), _h(scale).reshape(-1)
    Wq = np.ascontiguousarray(Wq_rm, np.int8)
    M, K = A.shape
    N = Wq.shape[1]
    Out = np.empty((M, N), np.float16)
    lib().mixq_oracle_w8a16_gemv(_i64(M), _i64(N), _i64(K), _p(A), _p(Wq), _p(scale), _p(Out))
    return Out


def w8a16_gemv_reforder(A, Wq_rm, scale):
    """The decode GEMV in the CUDA kernel's own summation order (fp16 per-thread FMA chains, kernel.h:425-470); M <= 4."""
    A, scale = _h(A), _h(scale).reshape(-1)
    Wq = np.ascontiguousarray(Wq_rm, np.int8)
    M, K = A.shape
    N = Wq.shape[1]
    assert M <= 4 and N % 4 == 0 and K % 64 == 0
    Out = np.empty((M, N), np.float16)
    lib().mixq_oracle_w8a16_gemv_reforder(_i64(M), _i64(N), _i64(K), _p(A), _p(Wq), _p(scale), _p(Out))
    return Out


def dequantization_silu(x, sA, sW, y):
    """cult.cu:2305-2324 (mixlib dequantizeInt8Silu): fp16(silu((x*sA)*sW + y)), new array."""
    x = np.ascontiguousarray(x, np.int32)
    M, N = x.shape
    y = _h(y)
    res = np.empty((M, N), np.float16)
    lib().mixq_oracle_dequantization_silu(_i64(M), _i64(N), _p(x), _p(_h(sA).reshape(-1)), _p(_h(sW).reshape(-1)),
                                          _p(y), _p(res))
    return res


def pack_linear_weights(W, act_scale, num_outliers=128):
    """modelopt/torch/export/model_config_utils.py:429-466 for ONE linear layer.

    W fp16 [N,K] (original weight), act_scale fp32 [K'] (K' may be < K: SURVEY A.3 #4).
    Returns the 7-tensor contract of SURVEY A.1 as a dict (true dtypes + the fp16 carrier views).
    """
    W = _h(W)
    sW = weight_scales(W)                                    # :429-430 (before zeroing)
    q_un, eetq_scales = eetq_symmetric_quantize(W.T.copy())  # :437-438 (un-zeroed W^T)
    qweight = eetq_preprocess(q_un)
    ind = select_outliers(act_scale, num_outliers)           # :446-448
    fp_weight = np.ascontiguousarray(W[:, ind])              # :452
    Wq = quantize_weight(W, sW, ind)                         # :453, :460-466
    return dict(
        weight=Wq, weights_scaling_factor=sW, fp_weight=fp_weight, fp_ind=ind, qweight=qweight,
        scales=eetq_scales,
        weight_as_half=int8_matrix_to_half(Wq), fp_ind_as_half=int_to_half(ind),
        qweight_as_half=int8_matrix_to_half(qweight),
    )


def rmsnorm_extract_quant(x, gamma, eps, ind=None):
    """layernorm.cu:122-198 (ind given) / :100-117 (ind None).  Returns out fp16 [M,K] (outlier columns zeroed),
    and with ind: outliers fp16 [M,len], q int8 [M,K], scale fp16 [M]."""
    x, gamma = _h(x), _h(gamma).reshape(-1)
    M, K = x.shape
    out = np.empty((M, K), np.float16)
    if ind is None:
        lib().mixq_oracle_rmsnorm_extract_quant(_i64(M), _i64(K), _p(x), _p(gamma), ctypes.c_float(eps), None,
                                                ctypes.c_int(0), _p(out), None, None, None)
        return out
    ind = np.ascontiguousarray(ind, np.int32)
    outl = np.empty((M, ind.size), np.float16)
    q = np.empty((M, K), np.int8)
    sc = np.empty((M,), np.float16)
    lib().mixq_oracle_rmsnorm_extract_quant(_i64(M), _i64(K), _p(x), _p(gamma), ctypes.c_float(eps), _p(ind),
                                            ctypes.c_int(ind.size), _p(out), _p(outl), _p(q), _p(sc))
    return out, outl, q, sc


# ---- P-flavour forward with dynamic outliers (MixQ/src/mixquant/modules/linear.py:155-286), numpy restatement ----
def find_outliers(A, sigma):
    """linear.py:155-161: torch.unique(torch.where(A.abs() > sigma)[1]).to(int32); fp16 compare, NaN -> False."""
    A = _h(A)
    with np.errstate(invalid="ignore"):
        hit = np.abs(A) > np.float16(sigma)
    return np.unique(np.where(hit)[1]).astype(np.int32)


def dequant_weight_columns(q_weight, scale_col, ind):
    """linear.py:207-209: q_weight[:, ind].to(float16) * scale_col.T  (fp16 x fp16 -> fp16, one rounding)."""
    q_weight = np.ascontiguousarray(q_weight, np.int8)
    s = _h(scale_col).reshape(-1, 1)
    return (q_weight[:, np.asarray(ind, np.int64)].astype(np.float16) * s).astype(np.float16)


class MixLinearState:
    """The mutable state of one MixLinear_GEMM layer + its cache, for `mixlinear_forward` (bit = 8)."""

    def __init__(self, q_weight, scale_col, sigma=6.0, stop=2, bias=None):
        self.q_weight = np.ascontiguousarray(q_weight, np.int8)
        self.scale_col = _h(scale_col).reshape(-1)
        self.sigma = np.float16(sigma)
        self.stop = stop
        self.bias = None if bias is None else _h(bias)
        self.ind = np.zeros((0,), np.int32)
        self.weight_cache = None
        self.cnt = 0
        self.add_outliers = True


def mixlinear_forward(st, x):
    """linear.py:163-286 with unfused=True (quantisation inside the layer).  Mutates `st` and zeroes the outlier columns
    of `x` in place, exactly like the reference mutates its input.  Returns fp16 [M, N]."""
    assert x.dtype == np.float16 and x.flags.c_contiguous and x.ndim == 2
    act_out = None
    if st.ind.size:
        act_out = extract_outliers(x, st.ind, set_zero=True)
    qx, xs = quant_rows(x)
    if st.add_outliers:
        # :201  x_scale.max() > sigma / 127 (fp16 tensor ops; NaN scales make the comparison False or propagate as in torch.max)
        thr = np.float16(st.sigma / np.float16(127))
        with np.errstate(invalid="ignore"):
            mx = np.max(xs) if not np.isnan(xs).any() else np.float16(np.nan)
            trig = bool(mx > thr)
        if trig:
            ind = find_outliers(x, st.sigma)
            new_out = extract_outliers(x, ind, set_zero=True)
            wc = dequant_weight_columns(st.q_weight, st.scale_col, ind)
            if st.ind.size == 0:
                act_out, st.weight_cache = new_out, wc
            else:
                act_out = np.hstack((act_out, new_out))
                st.weight_cache = np.hstack((st.weight_cache, wc))
            st.ind = np.hstack((st.ind, ind)).astype(np.int32)
            qx, xs = quant_rows(x)
        st.cnt += 1
        if st.cnt >= st.stop or st.ind.size > 256:
            st.add_outliers = False
    acc = gemm_s8s8s32(qx, st.q_weight)
    y = None
    if st.ind.size:
        y = gemm_fp16(np.ascontiguousarray(act_out), np.ascontiguousarray(st.weight_cache))
    out = dequant_epilogue(acc, xs, st.scale_col, C=y)
    if st.bias is not None:
        out = (out + st.bias[None, :]).astype(np.float16)
    return out


# ---- 4-bit (W4A4) flavour, numpy restatement (cult.cu:2515-2567, linear.py:11-17, 121-143) ----
def quant4_rows(A):
    """FindRowScaleKernel4bit: s = fp16(amax / 7) (NaN dropped from the max like __hmax), q = low 4 bits of
    half2int_rn(hdiv(x, s)); returns (packed uint8 [M, K/2] with element 2i in the low nibble, s fp16 [M])."""
    A = _h(A)
    M, K = A.shape
    with np.errstate(all="ignore"):
        absA = np.abs(A)
        amax = np.where(np.isnan(absA).all(axis=1), np.float16(np.nan),
                        np.nanmax(np.where(np.isnan(absA), np.float16(-1), absA), axis=1)).astype(np.float16)
        s = (amax / np.float16(7)).astype(np.float16)
        qh = (A / s[:, None]).astype(np.float16)                    # __hdiv: correctly rounded fp16 quotient
        r = np.rint(qh.astype(np.float64))                           # __half2int_rn: RNE
        r = np.where(np.isnan(r), 0, np.clip(r, -2147483648, 2147483647))   # NaN -> 0, +-inf saturate
    q = (r.astype(np.int64) & 0xF).astype(np.uint8)
    return (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8), s


def unpack_i4(packed):
    """packed uint8 [R, C/2] -> int8 [R, C] (sign-extended; low nibble = even column)."""
    p = np.asarray(packed, np.uint8)
    out = np.empty((p.shape[0], p.shape[1] * 2), np.int8)
    out[:, 0::2] = ((p & 0xF).astype(np.int16) ^ 8) - 8
    out[:, 1::2] = ((p >> 4).astype(np.int16) ^ 8) - 8
    return out


def pack_i4(x):
    """linear.py:11-17 pack_to_i4."""
    x = np.asarray(x, np.int8).astype(np.int16)
    u = np.where(x < 0, 16 + x, x).astype(np.uint8)
    return (u[:, 0::2] | (u[:, 1::2] << 4)).astype(np.uint8)


def mixlinear4_from_linear(W, layer_scales, fp_features_num=256):
    """linear.py:121-143 -> (q_weight packed, scale_col fp16 [N], ind int32, weight_cache fp16 [N, fp])."""
    W = _h(W).copy()
    ind = np.argsort(np.asarray(layer_scales, np.float32), kind="stable")[-fp_features_num:].astype(np.int32)
    wc = W[:, ind].copy()
    W[:, ind] = 0
    sc = (np.abs(W).max(axis=1) / np.float16(10)).astype(np.float16)
    q = np.clip(np.rint((W / sc[:, None]).astype(np.float16).astype(np.float64)), -8, 7).astype(np.int8)
    return pack_i4(q), sc, ind, wc


def mixlinear4_forward(q_weight_packed, scale_col, ind, weight_cache, x):
    """bit = 4 forward with a frozen outlier set (linear.py:183-193, 259-267): zero + extract the outlier columns,
    4-bit row quantisation, s4 x s4 -> s32 GEMM, fp16 outlier product as the addend of the dequant epilogue."""
    assert x.dtype == np.float16 and x.flags.c_contiguous
    act_out = extract_outliers(x, ind, set_zero=True)
    qx, xs = quant4_rows(x)
    acc = gemm_s8s8s32(unpack_i4(qx), unpack_i4(q_weight_packed))
    y = gemm_fp16(np.ascontiguousarray(act_out), np.ascontiguousarray(weight_cache))
    return dequant_epilogue(acc, xs, scale_col, C=y)

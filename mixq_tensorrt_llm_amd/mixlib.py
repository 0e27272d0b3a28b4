"""``mixlib`` operator surface (reference: quantkernel/mix_cuda/pybind_mix.cpp:256-335) on MI355X.

Same op names, argument order and return conventions as the reference's PyTorch extension, for the ops on the
``int8_mix`` path.  Each function is a thin launch of libmixq_mi355x.so on the current HIP stream;
tensors are only carriers of device pointers.  No op has a CPU implementation.
"""
import ctypes
import functools

import torch

from . import _lib

__all__ = ["FindRowScale", "ExtractOutliersAndSetToZeros", "int8FusedDequantize", "int8FusedDequantizeSilu", "gemm",
           "dequantizeInt8", "Int8quantize", "FindRowScaleFusedExtracOutliers", "layernorm_forward_cuda",
           "layernorm_forward_cuda_extract_outliers", "int_to_half", "int_matrix_to_half",
           "int8_matrix_to_half", "w8_a16_gemm", "preprocess_weights", "mixq_linear", "int4FusedDequantize",
           "int4FusedDequantizeSilu", "unpack_int4_to_fp16", "unpack_int4_to_int8", "int8FusedDequantizeSiluMul",
           "layernorm_forward_cuda_extract_outliers_int4", "ExtractOutliers", "dequantizeInt8Silu", "mixlinear_forward",
           "qa_layout", "qa_to_row_major", "QA_ROW_MAJOR", "QA_FRAGMENT_MAJOR", "WeightImage"]

QA_ROW_MAJOR, QA_FRAGMENT_MAJOR = 0, 1   # include/mixq.h MIXQ_QA_*


def qa_layout(M, N, K):
    """The int8 activation image a producer should write for a [M,K] x [N,K]^T consumer (include/mixq.h ``mixq_qa_layout``):
    QA_FRAGMENT_MAJOR for decode batches the weight-streaming skinny GEMM serves (its qA loads become contiguous 1-KiB reads),
    else QA_ROW_MAJOR.  The reference-named ops below default to row-major; our own producers / consumers
    (``mixlinear.FasterTransformerRMSNorm``, ``MixLinear_GEMM``) pass the layout along in the cache."""
    return int(_lib.load().mixq_qa_layout(int(M), int(N), int(K)))


def qa_to_row_major(q, M, K):
    """The fragment-major int8 image (QA_FRAGMENT_MAJOR: 1-KiB blocks [16-row tile][64-byte k-step], lane l = row % 16 + 16 * (k / 16 % 4)
    holding 16 bytes at l * 16 -- csrc/quant_kernels.hip FRAG) back to row-major [M, K].  A rare path: a consumer that shares a
    producer's cache but whose own shape the weight-streaming skinny GEMM does not serve (mixlinear.MixLinear_GEMM)."""
    tiles, steps = (M + 15) // 16, (K + 63) // 64
    img = q.reshape(-1)[:tiles * steps * 1024].view(tiles, steps, 4, 16, 16)     # [tile, k-step, k quarter, row, byte]
    return img.permute(0, 3, 1, 2, 4).reshape(tiles * 16, steps * 64)[:M, :K].contiguous()


class WeightImage:
    """A registered fragment-major copy of an int8 weight [N, K] (include/mixq.h ``mixq_weight_image_*``, MI355X extension): while it
    is alive, every decode-batch call (5 .. 64 rows) of the library on that weight POINTER streams the copy -- one contiguous 1-KiB
    read per load instead of 64 bytes of 16 rows -- with bit-identical results (-10..-15 % operator time at 32 rows).  Costs N * K
    bytes.  ``close()`` (or garbage collection) unregisters it; the weight tensor must not be freed or rewritten before that.  The
    library records a content tag of the weight at registration and compares the bytes behind the pointer with it around the image's first
    use (asynchronously: until that check has completed, calls read the weight itself; ``verify()`` checks on demand and synchronises): a tensor that was replaced behind the same address is served from its own bytes, not from
    the stale image.  A captured HIP graph holds the image pointer: keep this object alive as long as the graph."""

    def __init__(self, weight_int8):
        assert weight_int8.is_cuda and weight_int8.is_contiguous() and weight_int8.element_size() == 1 and weight_int8.dim() == 2
        n, k = weight_int8.shape
        lib = _lib.load()
        nbytes = int(lib.mixq_weight_image_bytes(n, k))
        if nbytes == 0:
            raise ValueError(f"no weight image for a [{n}, {k}] weight (N % 16 == 0 and K % 64 == 0 are needed)")
        self._weight = weight_int8                       # (keeps the registered address alive)
        self.image = torch.empty(nbytes, dtype=torch.int8, device=weight_int8.device)
        with torch.cuda.device(weight_int8.device):
            _lib.check(lib.mixq_weight_image_register(_p(weight_int8), n, k, _p(self.image), _st(weight_int8)), "weight_image_register")

    def verify(self) -> bool:
        """True if the weight still has the registered content (synchronises the weight's stream); False drops the registration."""
        if self._weight is None:
            return False
        with torch.cuda.device(self._weight.device):
            return _lib.load().mixq_weight_image_verify(_p(self._weight), _st(self._weight)) == 0

    def close(self):
        if self._weight is not None:
            _lib.load().mixq_weight_image_unregister(_p(self._weight))
            self._weight, self.image = None, None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass


def _alloc_q(m, k, layout, device):
    if layout == QA_FRAGMENT_MAJOR:   # opaque image: whole 16-row tiles x whole 64-byte k-steps
        return torch.empty(int(_lib.load().mixq_qa_bytes(m, k, layout)), dtype=torch.int8, device=device)
    return torch.empty((m, k), dtype=torch.int8, device=device)



_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # ~0.2 us; torch.cuda.current_stream() costs ~1.8 us
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _st(t):
    """The current HIP stream of ``t``'s device as a raw pointer (profiles/r03_mixlib_overhead.txt: the Python-level
    torch.cuda.current_stream(...).cuda_stream was 1.8 of the ~6.6 us a wrapper added to a direct C-ABI call)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else _cur_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t):
    return t.data_ptr() if t is not None else None


def _on_tensor_device(fn):
    """Run the op with the CUDA/HIP current device set to the device of its first tensor argument: the library caches
    per-device state under hipGetDevice() and launches on that tensor's current stream.  (The common case -- the tensor
    lives on the current device -- costs one attribute read and one C call.)"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, torch.Tensor):
                if a.is_cuda and a.device.index != _cur_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


def _rows_fit(scaleRow, rows, what):
    # the reference preallocates x_scale [inputdim, 1] (MixQ/src/mixquant/Cache.py:8) and never checks it
    assert scaleRow.numel() >= rows, f"{what}: scaleRow holds {scaleRow.numel()} rows, {rows} needed"


def _dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.MixQLibraryError("mixlib ops run on the GPU only (no CPU path exists)")
        assert t.is_contiguous(), "mixlib ops need contiguous tensors"


@_on_tensor_device
def FindRowScale(x, scaleRow, rows, cols, bit=8):
    """cult.cu:2569-2608.  Writes the per-row fp16 scale into ``scaleRow`` and returns the int8 rows."""
    _dev(x, scaleRow)
    _rows_fit(scaleRow, rows, "FindRowScale")
    if bit == 4:  # cult.cu:2588-2606: packed int4 pairs, scale = amax / 7
        assert cols % 2 == 0
        out = torch.empty((rows, cols // 2), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.load().mixq_int4quant(rows, cols, _p(x), _p(out), _p(scaleRow), _st(x)), "FindRowScale(4)")
        return out
    assert bit == 8
    out = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
    _lib.check(_lib.load().mixq_int8quant(rows, cols, _p(x), _p(out), _p(scaleRow), _st(x)), "FindRowScale")
    return out


@_on_tensor_device
def ExtractOutliersAndSetToZeros(ind, input):
    """cult.cu:1433-1465.  Returns input[:, ind] (fp16 [M, len]) and ZEROES those columns of ``input`` in place."""
    _dev(ind, input)
    assert ind.dtype == torch.int32 and input.dtype == torch.float16
    m, k = input.shape
    n = ind.shape[0]
    out = torch.empty((m, n), dtype=torch.float16, device=input.device)  # (every element is written: no memset launch)
    _lib.check(_lib.load().mixq_extract_outliers_set_zero(m, k, _p(input), _p(out), _p(ind), n, _st(input)),
               "ExtractOutliersAndSetToZeros")
    return out


_W8A16_PERSISTENT_SCRATCH_LIMIT = 64 << 20   # bytes of fpA_intB scratch kept per stream; larger = per call (two-pass form)
_GEMM_SCRATCH = {}
_GEMM_SCRATCH_RETIRED = []  # superseded buffers stay alive: a HIP graph captured earlier still holds their pointers


def _scratch_bytes(t, n):
    """The zero-initialised exchange scratch of (device, current stream), at least ``n`` bytes, or None while the stream
    is being captured and no buffer of that size exists yet.  Launches on one stream are ordered, which is what the
    kernels' hand-over words need; every kernel leaves those words zero, so all of them share the one buffer."""
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)
    buf = _GEMM_SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        if torch.cuda.is_current_stream_capturing():
            return None
        if buf is not None:
            _GEMM_SCRATCH_RETIRED.append(buf)
        # sized once for every int8 shape (the bound is ~56 MiB), so growth is the exception
        n = max(n, int(_lib.load().mixq_gemm_scratch_bound()))
        buf = torch.zeros(n, dtype=torch.uint8, device=t.device)
        _GEMM_SCRATCH[key] = buf
    return buf


def gemm_scratch(t, M, N, K):
    """Device scratch for the K split over workgroups (include/mixq.h, mixq_gemm_mixed_scratch): None when the shape does
    not use it.  Never allocates while the stream is being captured."""
    n = int(_lib.load().mixq_gemm_scratch_size(M, N, K))
    if n == 0:
        return None
    return _scratch_bytes(t, n)


def _fused(name, A, B, scale_row, scale_col, y, M, N, K, qa_layout=0, epilogue=0, mul=None):
    _dev(*(t for t in (A, B, scale_row, scale_col, y, mul) if t is not None))  # y = None: no addend (zeros in the reference)
    D = torch.empty((M, N), dtype=torch.float16, device=A.device)
    lib = _lib.load()
    if qa_layout:   # the fragment-major image (decode batches): one entry for the three epilogues
        _lib.check(lib.mixq_int8_fused_dequantize_layout(_p(A), _p(B), _p(scale_row), _p(scale_col), _p(y), _p(mul), _p(D), M, N,
                                                         K, epilogue, qa_layout, None, _st(A)), name)
        return D
    fn = getattr(lib, name)
    _lib.check(fn(_p(A), _p(B), _p(scale_row), _p(scale_col), _p(y), _p(D), M, N, K, _p(gemm_scratch(A, M, N, K)),
                  _st(A)), name)
    return D


@_on_tensor_device
def int8FusedDequantize(A, B, scale_row, scale_col, y, M, N, K, qa_layout=0):
    """cult.cu:1937-2000: D = fp16(float(A.B^T) * (scale_col[n]*scale_row[m]) + y), new tensor D.
    ``qa_layout`` (MI355X extension): the layout of ``A`` as its producer wrote it (``qa_layout()``); default row-major."""
    return _fused("mixq_int8_fused_dequantize", A, B, scale_row, scale_col, y, M, N, K, qa_layout, 0)


@_on_tensor_device
def int8FusedDequantizeSilu(A, B, scale_row, scale_col, y, M, N, K, qa_layout=0):
    """cult.cu:2067-2117: same with SiLU applied before the fp16 rounding."""
    return _fused("mixq_int8_fused_dequantize_silu", A, B, scale_row, scale_col, y, M, N, K, qa_layout, 1)


def _fused4(name, A, B, scale_row, scale_col, y, M, N, K, B_int8=None):
    _dev(*(t for t in (A, B, scale_row, scale_col, y) if t is not None))
    lib = _lib.load()
    D = torch.empty((M, N), dtype=torch.float16, device=A.device)
    need = int(lib.mixq_int4_fused_workspace_size(M, N, K))   # 0: the call streams the packed weight in ONE launch (decode batches the
    if need == 0:                                             # stream kernel serves, csrc/int4_gemm_kernels.hip); asked, not assumed
        ws = None
    elif B_int8 is not None:   # prefill size with the weight widened once at load: only A is widened per call
        ws = torch.empty(max(16, lib.mixq_int4_fused_workspace_size(M, 0, K)), dtype=torch.uint8, device=A.device)
        _lib.check(lib.mixq_int4_fused_dequantize_w8(_p(A), _p(B_int8), _p(scale_row), _p(scale_col), _p(y), _p(D), M, N, K,
                                                     1 if name.endswith("_silu") else 0, _p(ws), _st(A)), name + "_w8")
        return D
    else:
        ws = torch.empty(max(16, need), dtype=torch.uint8, device=A.device)
    _lib.check(getattr(lib, name)(_p(A), _p(B), _p(scale_row), _p(scale_col), _p(y), _p(D), M, N, K, _p(ws), _st(A)),
               name)
    return D


@_on_tensor_device
def int4FusedDequantize(A, B, scale_row, scale_col, y, M, N, K, B_int8=None):
    """cult.cu:2005-2060: packed-int4 A [M,K] / B [N,K] (K = packed bytes per row = in_features // 2, as the reference
    passes it).  No int4 MFMA on gfx950, same int32 all the same: up to 64 rows the packed weight is streamed once and widened in
    registers (one launch); above, the operands are sign-extended to int8 for the int8 kernels -- ``B_int8`` (MI355X extension):
    the weight already widened at load time (``unpack_int4_to_int8``), then only A is widened per call."""
    return _fused4("mixq_int4_fused_dequantize", A, B, scale_row, scale_col, y, M, N, K, B_int8)


@_on_tensor_device
def int4FusedDequantizeSilu(A, B, scale_row, scale_col, y, M, N, K, B_int8=None):
    """cult.cu:2119-2181."""
    return _fused4("mixq_int4_fused_dequantize_silu", A, B, scale_row, scale_col, y, M, N, K, B_int8)


@_on_tensor_device
def int4_linear_forward(x, B, scale_col, y, x_scale, silu=False):
    """MI355X extension (``mixq_int4_linear_forward``): FindRowScale(x, x_scale, M, K, 4) + int4FusedDequantize[Silu] in ONE call; a single
    row runs as ONE launch (quantised inside the weight-streaming kernel), more rows as the two launches.  x fp16 [M, K], B packed uint8
    [N, K // 2]; ``x_scale`` is written like the reference's cache.x_scale.  Returns (out fp16 [M, N], q packed uint8 [M, K // 2])."""
    _dev(*(t for t in (x, B, scale_col, y, x_scale) if t is not None))
    M, K = x.shape
    N, kp = B.shape
    assert K == 2 * kp and x.dtype == torch.float16 and B.dtype == torch.uint8
    _rows_fit(x_scale, M, "int4_linear_forward")
    lib = _lib.load()
    D = torch.empty((M, N), dtype=torch.float16, device=x.device)
    q = torch.empty((M, kp), dtype=torch.uint8, device=x.device)   # (written unless the call ran as one launch)
    need = int(lib.mixq_int4_fused_workspace_size(M, N, kp))
    ws = torch.empty(max(16, need), dtype=torch.uint8, device=x.device) if need else None
    spare = q
    _lib.check(lib.mixq_int4_linear_forward(_p(x), _p(B), _p(x_scale), _p(spare), _p(scale_col), _p(y), _p(D), M, N, kp, 1 if silu else 0,
                                            _p(ws), _st(x)), "int4_linear_forward")
    return D, q


@_on_tensor_device
def unpack_int4_to_int8(packed):
    """Sign-extending unpack of a packed int4 tensor [rows, cols / 2] -> int8 [rows, cols] (``mixq_unpack_int4_to_int8``)."""
    _dev(packed)
    assert packed.dtype == torch.uint8 and packed.is_contiguous() and packed.numel() % 16 == 0
    out = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=torch.int8, device=packed.device)
    _lib.check(_lib.load().mixq_unpack_int4_to_int8(_p(packed), _p(out), packed.numel(), _st(packed)), "unpack_int4_to_int8")
    return out


@_on_tensor_device
def unpack_int4_to_fp16(weight, ind):
    """cult.cu:3088-3118: fp16 [rows, len(ind)] = the int4 values of columns `ind` of packed `weight` [rows, cols/2]."""
    _dev(weight, ind)
    assert weight.dtype == torch.uint8 and ind.dtype == torch.int32
    rows, colsp = weight.shape
    n = ind.shape[0]
    out = torch.zeros((rows, n), dtype=torch.float16, device=weight.device)
    _lib.check(_lib.load().mixq_unpack_int4_to_fp16(_p(weight), _p(ind), rows, colsp, n, _p(out), _st(weight)),
               "unpack_int4_to_fp16")
    return out


@_on_tensor_device
def int8FusedDequantizeSiluMul(A, B, scale_row, scale_col, y, mul, M, N, K, qa_layout=0):
    """MI355X extension (no reference op): int8FusedDequantizeSilu followed by ``*= mul`` (fused/mlp.py:61-63) in ONE
    kernel: D = fp16(fp16(silu(...)) * mul), the same bits as the two-step sequence."""
    if qa_layout:
        return _fused("int8FusedDequantizeSiluMul", A, B, scale_row, scale_col, y, M, N, K, qa_layout, 3, mul)
    _dev(*(t for t in (A, B, scale_row, scale_col, y, mul) if t is not None))
    D = torch.empty((M, N), dtype=torch.float16, device=A.device)
    _lib.check(_lib.load().mixq_int8_fused_dequantize_silu_mul(_p(A), _p(B), _p(scale_row), _p(scale_col), _p(y), _p(mul),
                                                              _p(D), M, N, K, _p(gemm_scratch(A, M, N, K)), _st(A)),
               "int8FusedDequantizeSiluMul")
    return D


@_on_tensor_device
def gemm(mat1, mat2, m, n, k):
    """cult.cu:180-220 (cuBLAS s8 x s8 -> s32): int32 [m,n] = mat1[m,k] . mat2[n,k]^T."""
    _dev(mat1, mat2)
    out = torch.empty((m, n), dtype=torch.int32, device=mat1.device)
    _lib.check(_lib.load().mixq_gemm_s8s8s32(_p(mat1), _p(mat2), _p(out), m, n, k, _st(mat1)), "gemm")
    return out


@_on_tensor_device
def dequantizeInt8(x, scaleRow, scaleCol, y, bits, M, N):
    """cult.cu:2258-2288: out = hadd(fp16((float(x)*scaleRow[m])*scaleCol[n]), y), new tensor."""
    _dev(x, scaleRow, scaleCol, y)
    out = y.clone()
    _lib.check(_lib.load().mixq_dequantization(_p(out), _p(x), _p(scaleRow), _p(scaleCol), M, N, _st(x)),
               "dequantizeInt8")
    return out


@_on_tensor_device
def dequantizeInt8Silu(x, scaleRow, scaleCol, y, bits, M, N):
    """cult.cu:2341-2348 (-> dequantizationKernelSilu :2305-2324): fp16(silu((float(x)*scaleRow[m])*scaleCol[n] + y)),
    new tensor.  The P-flavour's sm90 route calls it after mixlib.gemm (linear.py:321-324)."""
    _dev(x, scaleRow, scaleCol, y)
    out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().mixq_dequantization_silu(_p(out), _p(x), _p(scaleRow), _p(scaleCol), _p(y), M, N, _st(x)),
               "dequantizeInt8Silu")
    return out


@_on_tensor_device
def Int8quantize(src, scale):
    """cult.cu:1732-1771: dst = (int8) half2int_rn(hdiv(src, scale[row])) with a caller-supplied per-row scale."""
    _dev(src, scale)
    rows, cols = src.shape
    dst = torch.empty((rows, cols), dtype=torch.int8, device=src.device)
    _lib.check(_lib.load().mixq_int8_quantize_with_scale(rows, cols, _p(src), _p(scale), _p(dst), _st(src)),
               "Int8quantize")
    return dst


@_on_tensor_device
def FindRowScaleFusedExtracOutliers(x, scaleRow, ind, len_ind, rows, cols, q_layout=0):
    """cult.cu:2671-2709: returns [int8 rows, outliers fp16 [rows,len_ind]]; zeroes the outlier columns of x.
    ``q_layout`` (MI355X extension): write the rows in the consumer's preferred image (``qa_layout()``)."""
    _dev(x, scaleRow)
    _rows_fit(scaleRow, rows, "FindRowScaleFusedExtracOutliers")
    q = _alloc_q(rows, cols, q_layout, x.device)
    outl = torch.empty((rows, len_ind), dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().mixq_quant_extract_layout(rows, cols, _p(x), _p(q), _p(scaleRow), _p(outl),
                                                     _p(ind) if len_ind else None, len_ind, 1, q_layout, _st(x)),
               "FindRowScaleFusedExtracOutliers")
    return [q, outl]


@_on_tensor_device
def layernorm_forward_cuda(_input, _gamma, _out, eps):
    """layernorm.cu:100-117: T5-style RMSNorm of _input [b, n, c] (or [m, c]) into _out."""
    _dev(_input, _gamma, _out)
    c = _input.shape[-1]
    m = _input.numel() // c
    _lib.check(_lib.load().mixq_rmsnorm(m, c, _p(_input), _p(_gamma), _p(_out), ctypes.c_float(eps), _st(_input)),
               "layernorm_forward_cuda")


@_on_tensor_device
def layernorm_forward_cuda_extract_outliers(_input, _gamma, _out, eps, _ind, scaleRow, q_layout=0):
    """layernorm.cu:316-346: fused RMSNorm -> extract(+zero) outliers -> per-row int8 quantisation.
    Fills _out (normalised, outlier columns zeroed) and scaleRow; returns [outliers fp16 [m,len], quant int8 [m,c]].
    ``q_layout`` (MI355X extension): write the int8 rows in the consumer's preferred image (``qa_layout()``)."""
    _dev(_input, _gamma, _out, _ind, scaleRow)
    c = _input.shape[-1]
    m = _input.numel() // c
    _rows_fit(scaleRow, m, "layernorm_forward_cuda_extract_outliers")
    n = _ind.shape[0]
    outl = torch.zeros((m, n), dtype=torch.float16, device=_input.device)
    q = _alloc_q(m, c, q_layout, _input.device)
    _lib.check(_lib.load().mixq_rmsnorm_extract_quant_layout(m, c, _p(_input), _p(_gamma), _p(_out), ctypes.c_float(eps),
                                                             _p(_ind), n, _p(outl), _p(q), _p(scaleRow), q_layout,
                                                             _st(_input)),
               "layernorm_forward_cuda_extract_outliers")
    return [outl, q]


@_on_tensor_device
def layernorm_forward_cuda_extract_outliers_int4(_input, _gamma, _out, eps, _ind, scaleRow):
    """layernorm.cu:379-414: the same fused producer with packed 4-bit rows (scale = amax / 7).
    Returns [outliers fp16 [m,len], quant uint8 [m, c/2]]."""
    _dev(_input, _gamma, _out, _ind, scaleRow)
    c = _input.shape[-1]
    m = _input.numel() // c
    _rows_fit(scaleRow, m, "layernorm_forward_cuda_extract_outliers_int4")
    n = _ind.shape[0]
    outl = torch.zeros((m, n), dtype=torch.float16, device=_input.device)
    q = torch.empty((m, c // 2), dtype=torch.uint8, device=_input.device)
    _lib.check(_lib.load().mixq_rmsnorm_extract_quant4(m, c, _p(_input), _p(_gamma), _p(_out), ctypes.c_float(eps),
                                                       _p(_ind), n, _p(outl), _p(q), _p(scaleRow), _st(_input)),
               "layernorm_forward_cuda_extract_outliers_int4")
    return [outl, q]


@_on_tensor_device
def ExtractOutliers(ind, input):
    """cult.cu ExtractOutliers: input[:, ind] as fp16 [M, len]; `input` is left untouched (T-flavour gather)."""
    _dev(ind, input)
    m, k = input.shape
    n = ind.shape[0]
    out = torch.zeros((m, n), dtype=torch.float16, device=input.device)
    _lib.check(_lib.load().mixq_extract_outliers(m, k, _p(input), _p(out), _p(ind), n, _st(input)), "ExtractOutliers")
    return out


def int_to_half(int_ind):
    """cult.cu:3046-3056: bit reinterpretation int32 [n] -> fp16 [2n]."""
    return int_ind.contiguous().view(torch.float16).clone()


def int_matrix_to_half(int_ind):
    """cult.cu:3058-3070: int32 [m,n] -> fp16 [m,2n], same bytes."""
    return int_ind.contiguous().view(torch.float16).clone()


def int8_matrix_to_half(int_ind):
    """cult.cu:3072-3085: int8 [m,n] -> fp16 [m,n/2], same bytes."""
    return int_ind.contiguous().view(torch.float16).clone()


@_on_tensor_device
def w8_a16_gemm(input, weight, scale):
    """EETQ/csrc/eetpy.cpp:7-19 w8_a16_gemm: fp16 [m,k] x interleaved uint8 [k,n] -> fp16 [m,n].
    m <= 4: batched GEMV; m > 4: the fpA_intB MFMA GEMM (fpA_intB_gemm_wrapper.cu:45-70), with the per-stream scratch
    for its K split over workgroups."""
    _dev(input, weight, scale)
    m, k = input.shape
    n = scale.numel()
    out = torch.empty((m, n), dtype=torch.float16, device=input.device)
    lib = _lib.load()
    nws = int(lib.mixq_w8a16_gemm_workspace_size(m, n, k))
    # the two-pass form (m >= 1280) wants 16 KiB + N*K*2 bytes (the fp16 image of W: ~470 MB for 28672 x 8192); parking
    # that in the never-released per-stream scratch would pin up to > 1 GB per stream.  Large requests come from the caching
    # allocator per call instead (zeroed hand-over words; freed when the call's tensors die -- stream-ordered, so safe), unless a
    # graph is being captured (then the small persistent scratch is offered and the library picks a form that fits it).
    if nws > _W8A16_PERSISTENT_SCRATCH_LIMIT and not torch.cuda.is_current_stream_capturing():
        scr = torch.empty(nws, dtype=torch.uint8, device=input.device)
        scr[:16384].zero_()
    else:
        scr = _scratch_bytes(input, min(nws, _W8A16_PERSISTENT_SCRATCH_LIMIT)) if nws else None
    _lib.check(lib.mixq_w8a16_gemm_forward_ws(_p(input), _p(weight), _p(scale), _p(out), m, n, k, _p(scr),
                                              scr.numel() if scr is not None else 0, _st(input)), "w8_a16_gemm")
    return out


def preprocess_weights(row_major_int8):
    """EETQ preprocess_weights (cutlass_preprocessors.cc:536-545), int8: host tensor [K,N] -> interleaved uint8."""
    w = row_major_int8.contiguous().cpu()
    assert w.dtype == torch.int8
    out = torch.empty_like(w, dtype=torch.uint8)
    _lib.check(_lib.load().mixq_preprocess_weights_int8(ctypes.c_void_p(out.data_ptr()),
                                                        ctypes.c_void_p(w.data_ptr()), w.shape[0], w.shape[1]),
               "preprocess_weights")
    return out


@_on_tensor_device
def mixq_linear(A, W_int8, sW, fp_weight, ind, out=None, workspace=None):
    """The fused two-launch prefill path used inside enqueue, on plain tensors:
    A fp16 [M,K], W int8 [N,K], sW fp16 [N], fp_weight fp16 [N,O], ind int32 [O] -> fp16 [M,N]."""
    _dev(A, W_int8, sW, fp_weight, ind)
    M, K = A.shape
    N, O = fp_weight.shape
    lib = _lib.load()
    dev = A.device
    if workspace is None:
        qA = torch.empty((M, K), dtype=torch.int8, device=dev)
        sA = torch.empty((M,), dtype=torch.float16, device=dev)
        fpA = torch.empty((M, O), dtype=torch.float16, device=dev)
    else:
        qA, sA, fpA = workspace
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
    _lib.check(lib.mixq_quant_extract(M, K, _p(A), _p(qA), _p(sA), _p(fpA), _p(ind), O, 0, _st(A)), "quant_extract")
    scr = gemm_scratch(A, M, N, K)
    _lib.check(lib.mixq_gemm_mixed_scratch(_p(qA), _p(W_int8), _p(sA), _p(sW), _p(fpA), _p(fp_weight), _p(out), M, N, K, O,
                                           _p(scr), scr.numel() if scr is not None else 0, _st(A)), "gemm_mixed")
    return out


@_on_tensor_device
def mixlinear_forward(x, ind, q_weight, scale_col, weight_cache, x_scale, q_layout=0):
    """MixLinear_GEMM.forward (linear.py:163-286, bit = 8, static outlier set) in ONE library call and two launches (MI355X
    extension, include/mixq.h ``mixq_mixlinear_forward``): returns (out fp16 [M,N], q_x int8 [M,K], outliers fp16 [M,O]); zeroes
    the ``ind`` columns of ``x`` and fills ``x_scale`` like the reference's four-call sequence.  The wrapper does what one of
    the four wrappers does -- the host cost per linear drops ~4x (profiles/r03_mixlib_overhead.txt)."""
    _dev(x, ind, q_weight, scale_col, x_scale)   # (the same guards as the four wrappers this call replaces: raw pointers go to C)
    if weight_cache is not None:
        _dev(weight_cache)
    assert x.dtype == torch.float16 and ind.dtype == torch.int32 and q_weight.dtype == torch.int8
    assert scale_col.dtype == torch.float16 and x_scale.dtype == torch.float16
    M, K = x.shape
    N = q_weight.shape[0]
    O = int(ind.shape[0])
    assert q_weight.shape[1] == K and scale_col.numel() >= N, "mixlinear_forward: q_weight [N,K] / scale_col [N] do not match x [M,K]"
    assert weight_cache is None or O == 0 or (weight_cache.dtype == torch.float16 and weight_cache.numel() >= N * O)
    _rows_fit(x_scale, M, "mixlinear_forward")
    dev = x.device
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    q_x = _alloc_q(M, K, q_layout, dev)   # (q_layout = qa_layout(M, N, K): the opaque fragment-major image for decode batches)
    outliers = torch.empty((M, O), dtype=torch.float16, device=dev)
    lib = _lib.load()
    scratch = None if q_layout else gemm_scratch(x, M, N, K)
    _lib.check(lib.mixq_mixlinear_forward(M, N, K, O, x.data_ptr(), ind.data_ptr() if O else None, q_weight.data_ptr(),
                                          scale_col.data_ptr(), weight_cache.data_ptr() if (O and weight_cache is not None) else None,
                                          x_scale.data_ptr(),
                                          q_x.data_ptr(), outliers.data_ptr() if O else None, out.data_ptr(), q_layout,
                                          scratch.data_ptr() if scratch is not None else None,
                                          scratch.numel() if scratch is not None else 0,
                                          _st(x)), "mixlinear_forward")
    return out, q_x, outliers

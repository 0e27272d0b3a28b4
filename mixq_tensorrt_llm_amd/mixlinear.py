"""P-flavour MixQ linear with DYNAMIC outlier detection (reference: MixQ/src/mixquant/modules/linear.py:24-286,
MixQ/src/mixquant/Cache.py:5-24) on MI355X.

``MixLinear_GEMM`` keeps the reference's state machine: int8 ``q_weight`` [N,K] with per-row ``scale_col``; an outlier
index set ``ind`` that starts empty and grows while ``add_outliers`` is on (the first ``cache.stop`` calls, until more
than 256 columns are collected): whenever some row scale exceeds ``sigma / 127``, the columns holding an |a| > sigma are
found, moved out of the int8 path (zeroed in the activation, their weights dequantised into ``weight_cache``), and the
rows are re-quantised.  The product is  fp16(float(qA . qW^T) * (scale_col * x_scale) + outliers . weight_cache^T).

Every tensor op of the forward is a launch of libmixq_mi355x.so (no torch.mm / torch.where / torch.unique on the data
path): FindOutliers -> `mixq_find_outliers`, weight_cache columns -> `mixq_dequant_weight_columns`, the outlier product
-> `mixq_gemm_fp16`, the rest are the mixlib ops.  torch is used for allocation, `hstack` of the growing state and the
quantize-time `from_linear` (like the reference).  bit = 8 and bit = 4 (W4A4: packed int4 storage, sign-extended to
int8 for the MFMA kernels -- gfx950 has no int4 matrix instruction) and the weight-only W8A16 mode.
"""
import ctypes

import torch

from . import _lib, mixlib


from .mixlib import _p, _st  # noqa: E402  (raw pointers / the raw current stream: cheap on the host)


def find_outliers(activation: torch.Tensor, sigma: float, capacity: int = None) -> torch.Tensor:
    """linear.py:155-161 ``torch.unique(torch.where(A.abs() > sigma)[1]).to(int32)``: ascending column indices."""
    assert activation.is_cuda and activation.dtype == torch.float16 and activation.dim() == 2
    a = activation.contiguous()
    m, k = a.shape
    cap = k if capacity is None else capacity
    lib = _lib.load()
    ws = torch.empty((lib.mixq_find_outliers_workspace_size(k) + 3) // 4, dtype=torch.int32, device=a.device)
    ind = torch.empty(cap, dtype=torch.int32, device=a.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=a.device)
    _lib.check(lib.mixq_find_outliers(_p(a), m, k, ctypes.c_float(float(sigma)), _p(ws), _p(ind), _p(cnt), cap, _st(a)),
               "find_outliers")
    n = int(cnt.item())  # the reference's torch.unique synchronises too (data-dependent output size)
    return ind[: min(n, cap)]


def dequant_weight_columns(q_weight: torch.Tensor, scale_col: torch.Tensor, ind: torch.Tensor) -> torch.Tensor:
    """linear.py:207-209 ``q_weight[:, ind].to(float16) * scale_col.T`` -> fp16 [N, len(ind)]."""
    assert q_weight.is_cuda and q_weight.dtype == torch.int8 and ind.dtype == torch.int32
    n, k = q_weight.shape
    out = torch.empty((n, ind.shape[0]), dtype=torch.float16, device=q_weight.device)
    _lib.check(_lib.load().mixq_dequant_weight_columns(_p(q_weight), _p(scale_col), _p(ind), ind.shape[0], _p(out), n,
                                                       k, _st(q_weight)), "dequant_weight_columns")
    return out


def two_compl(x: torch.Tensor, bits: int) -> torch.Tensor:
    """linear.py:11-12."""
    return torch.where(x < 0, 2 ** bits + x, x)


def pack_to_i4(X: torch.Tensor) -> torch.Tensor:
    """linear.py:13-17: int8 values in [-8, 7] -> packed pairs, even column in the low nibble."""
    X_i8 = two_compl(X.to(dtype=torch.int8), 4).to(torch.uint8)
    return X_i8[:, 0::2] | (X_i8[:, 1::2] << 4)


def outlier_product(activation_outliers: torch.Tensor, weight_cache: torch.Tensor) -> torch.Tensor:
    """linear.py:241 ``torch.mm(activation_outliers, weight_cache.T)``: fp16 out, fp32 accumulate."""
    m, o = activation_outliers.shape
    n = weight_cache.shape[0]
    out = torch.empty((m, n), dtype=torch.float16, device=activation_outliers.device)
    _lib.check(_lib.load().mixq_gemm_fp16(_p(activation_outliers.contiguous()), _p(weight_cache.contiguous()), _p(out),
                                          m, n, o, _st(out)), "gemm_fp16")
    return out


class MixLibCache:
    """Cache.py:5-24: per-model scratch shared by the layers (row scales, sigma, the running outlier state)."""

    def __init__(self, inputdim=1024, sigma=6, bit=8, device="cuda"):
        self.device = device
        self.x_scale = torch.zeros((inputdim, 1), dtype=torch.float16, device=device)
        self.sigma = torch.zeros((1, 1), dtype=torch.float16, device=device)
        self.sigma[0] = sigma
        self.zeros = None  # (the reference keeps an [inputdim, 36864] zero matrix as the "no outliers" addend; a null
        #                     addend is passed to the fused GEMM instead)
        self.ind = None
        self.new_ind = None
        self.shape = None
        self.activation_outliers = None
        self.q_xcache = None
        self.q_layout = 0      # layout of q_xcache (mixlib.QA_ROW_MAJOR / QA_FRAGMENT_MAJOR: MI355X extension, decode batches)
        self.is_prefill = False
        self.bit = bit
        self.max_outliers = 256
        self.stop = 2


class MixLinear_GEMM:
    """linear.py:24-286: bit = 8, bit = 4 and the weight_only W8A16 mode."""

    def __init__(self, in_features, out_features, bias, dev, bit=8, weight_only=False, cache=None,
                 fp_features_num=256):
        assert bit in (8, 4)
        self.in_features, self.out_features, self.bit = in_features, out_features, bit
        self.fp_features_num = fp_features_num
        self.weight_only = weight_only
        self.cache = cache
        if weight_only:
            self.q_weight = torch.empty((in_features, out_features), dtype=torch.uint8, device=dev)
            self.scale_col = torch.empty((out_features,), dtype=torch.float16, device=dev)
        else:
            self.scale_col = torch.empty((1, out_features), dtype=torch.float16, device=dev)
            if bit == 8:
                self.q_weight = torch.empty((out_features, in_features), dtype=torch.int8, device=dev)
                self.ind = torch.zeros((0,), dtype=torch.int32, device=dev)
                self.weight_cache = None
            else:  # :44-54: packed int4 weights + a static set of fp_features_num fp16 outlier columns
                self.q_weight = torch.empty((out_features, in_features // 2), dtype=torch.uint8, device=dev)
                self.weight_cache = torch.empty((out_features, fp_features_num), dtype=torch.float16, device=dev)
                self.ind = torch.empty((fp_features_num,), dtype=torch.int32, device=dev)
        self.bias = torch.empty((out_features,), dtype=torch.float16, device=dev) if bias else None
        self.cnt = 0
        # linear.py:86 `self.arch = torch.cuda.get_device_capability()[0]`, read at :231 / :317: 9 (sm90) takes the UNFUSED route
        # mixlib.gemm -> outlier product -> mixlib.dequantizeInt8[Silu]; everything else the fused GEMM.  An MI355X reports no CUDA
        # capability of 9, so the default is the fused route; set `arch = 9` to run the reference's sm90 sequence op for op (bit = 8).
        self.arch = 0
        self.forward_without_precondition_len = fp_features_num if (bit == 4 and not weight_only) else -1  # :68-71
        self.add_outliers = True
        self.sigma = None
        if cache is not None:
            self.sigma = torch.ones((1, 1), dtype=torch.float16, device=dev)
            self.sigma[0] = cache.sigma[0]

    @classmethod
    def from_linear(cls, weight, bias=None, bit=8, weight_only=False, cache=None, dev="cuda", layer_scales=None,
                    fp_features_num=256):
        """linear.py:88-149 (``linear.weight`` fp16 [N,K], optional bias; ``layer_scales`` = per-input-channel
        activation scales, needed for bit = 4 to pick the static outlier columns)."""
        n, k = weight.shape
        q = cls(k, n, bias is not None, dev, bit=bit, weight_only=weight_only, cache=cache,
                fp_features_num=fp_features_num)
        if weight_only:  # :102-106 EETQ quant_weights of W^T
            from . import pack
            qweight, scales, _ = pack.eetq_quant_weights(weight.t().contiguous())
            q.q_weight.copy_(qweight.to(dev))
            q.scale_col.copy_(scales.to(dev))
        elif bit == 4:   # :121-143: top-k activation-scale columns stay fp16; scale = fp16(max|w_rest| / 10), clamp [-8, 7]
            assert layer_scales is not None
            ind = torch.sort(layer_scales.float().cpu(), stable=True)[1][-fp_features_num:]  # (stable: see pack.py)
            w = weight.to(dev).clone()
            q.weight_cache.copy_(w[:, ind.to(dev)])
            w[:, ind.to(dev)] = 0
            scale = (torch.max(torch.abs(w), dim=1)[0].unsqueeze(1) / 10).to(torch.float16).reshape((1, n))
            q.scale_col.copy_(scale)
            w /= q.scale_col.T
            w = torch.clamp(w.round(), -8, 7)
            q.q_weight.copy_(pack_to_i4(w.to(torch.int8).cpu()).to(dev))
            q.ind.copy_(ind.to(torch.int32).to(dev))
        else:            # :113-120: scale = fp16(max|w| / 127) per row; q = round(w / scale) (no clamp)
            w = weight.to(dev)
            scale = (torch.max(torch.abs(w), dim=1)[0].unsqueeze(1) / 127).to(torch.float16).reshape((1, n))
            q.scale_col.copy_(scale)
            tmp = w.clone()
            tmp /= q.scale_col.T
            q.q_weight.copy_(tmp.round().to(torch.int8))
        if bias is not None:
            q.bias.copy_(bias.to(dev).half())
        return q

    def prepare_decode_batches(self, enable=True):
        """MI355X extension: register a fragment-major streaming copy of ``q_weight`` (bit = 8) for decode batches of 5 .. 64 rows
        (mixlib.WeightImage; same bits, -10..-15 % per call, N * K more bytes).  Call after the weights are final."""
        if getattr(self, "weight_image", None) is not None:
            self.weight_image.close()
        self.weight_image = None
        if enable and self.bit == 8 and not self.weight_only:
            self.weight_image = mixlib.WeightImage(self.q_weight)
        return self

    def prepare_prefill(self, enable=True):
        """MI355X extension (bit = 4): keep the packed weight ALSO widened to int8 (+ N * K bytes) so that prefill-size calls
        (more than 64 rows: MFMA-bound, the int8 kernels) widen only the activation per call instead of both operands; decode
        batches stream the packed ``q_weight`` either way.  Call after the weights are final; ``enable=False`` drops the copy."""
        self.q_weight_i8 = mixlib.unpack_int4_to_int8(self.q_weight) if (enable and self.bit == 4 and not self.weight_only) else None
        return self

    def _q_weight_i8(self, M):
        return getattr(self, "q_weight_i8", None) if M > 64 else None

    def FindOutliers(self, activation):
        """linear.py:155-161."""
        return find_outliers(activation, float(self.sigma[0, 0]))

    @torch.no_grad()
    def forward(self, x, cache=None, unfused=False):
        """linear.py:163-286.  ``unfused=True`` quantises ``x`` here; with the default (``False``, as in the reference)
        q_xcache / x_scale / activation_outliers come through the cache from the preceding fused norm layer
        (`FasterTransformerRMSNorm` with ``next_layer`` set)."""
        cache = self.cache if cache is None else cache
        cache.shape = x.shape[:-1] + (self.out_features,)
        inputs = x.reshape(-1, x.shape[-1])
        M = inputs.shape[0]
        if self.weight_only:
            y = mixlib.w8_a16_gemm(inputs, self.q_weight, self.scale_col)
            if self.bias is not None:
                y += self.bias
            return y.reshape(cache.shape)

        if unfused and self.bit == 8 and not self.add_outliers and inputs.is_contiguous():
            # static outlier set: the reference's four mixlib calls (:186-189, :241-247) as ONE library call and two launches
            # (mixq_mixlinear_forward); a decode step is host-bound on those calls long before it is GPU-bound
            # (profiles/r03_mixlib_overhead.txt: 42.6 us of host time per linear through four wrappers, 16 us as four direct
            # C-ABI calls, 7.8 us as this one)
            assert cache.x_scale.numel() >= M
            cache.ind = self.ind
            wc = self.weight_cache if self.ind.shape[0] else None
            n_out = int(self.ind.shape[0])
            cache.q_layout = mixlib.qa_layout(M, self.out_features, self.in_features) if (n_out % 8 == 0 and n_out <= 128) else 0
            y1, cache.q_xcache, cache.activation_outliers = mixlib.mixlinear_forward(
                inputs, self.ind, self.q_weight, self.scale_col, wc, cache.x_scale, cache.q_layout)
            if self.bias is not None:
                y1 += self.bias
            return y1.reshape(cache.shape)
        if unfused:
            if self.ind.shape[0]:
                cache.activation_outliers = mixlib.ExtractOutliersAndSetToZeros(self.ind, inputs)
            cache.q_xcache = mixlib.FindRowScale(inputs, cache.x_scale, M, self.in_features, self.bit)
            cache.q_layout = 0
        cache.ind = self.ind

        if self.add_outliers:
            # :201 (a device->host read of one scalar, like the reference's `if tensor > tensor`)
            if bool(cache.x_scale[0:M].max() > self.sigma / (2 ** (self.bit - 1) - 1)):
                ind = self.FindOutliers(inputs)
                cache.new_ind = ind
                activation_outliers = mixlib.ExtractOutliersAndSetToZeros(ind, inputs)
                if self.bit == 8:
                    weight_cache = dequant_weight_columns(self.q_weight, self.scale_col, ind)
                else:  # :210-212 (fp16 x fp16 product, like the reference's tensor expression)
                    weight_cache = mixlib.unpack_int4_to_fp16(self.q_weight, ind) * self.scale_col.T
                if self.ind.shape[0] == 0:
                    cache.activation_outliers = activation_outliers
                    self.weight_cache = weight_cache
                else:
                    cache.activation_outliers = torch.hstack((cache.activation_outliers, activation_outliers))
                    self.weight_cache = torch.hstack((self.weight_cache, weight_cache))
                self.ind = torch.hstack((self.ind, ind))
                cache.ind = self.ind
                cache.q_xcache = mixlib.FindRowScale(inputs, cache.x_scale, M, self.in_features, self.bit)
                cache.q_layout = 0
            self.cnt += 1
            if self.cnt >= cache.stop or self.ind.shape[0] > 256:
                self.add_outliers = False

        y = outlier_product(cache.activation_outliers, self.weight_cache) if self.ind.shape[0] else None
        if self.arch == 9 and self.bit == 8:   # :231-238
            y1 = mixlib.dequantizeInt8(mixlib.gemm(self._row_major_q(cache, M), self.q_weight, M, self.out_features, self.in_features),
                                       cache.x_scale, self.scale_col, y if y is not None else self._zeros(M, inputs.device), 8, M, self.out_features)
        elif self.bit == 8:
            q, lay = self._cached_q(cache, M)
            y1 = mixlib.int8FusedDequantize(q, self.q_weight, cache.x_scale, self.scale_col, y, M,
                                            self.out_features, self.in_features, lay)
        else:
            y1 = mixlib.int4FusedDequantize(cache.q_xcache, self.q_weight, cache.x_scale, self.scale_col, y, M,
                                            self.out_features, self.in_features // 2, self._q_weight_i8(M))
        if self.bias is not None:
            y1 += self.bias
        return y1.reshape(cache.shape)

    __call__ = forward

    def _row_major_q(self, cache, M):
        """q_xcache as the reference's unfused route reads it (row-major int8 [M, K])."""
        if getattr(cache, "q_layout", 0):
            cache.q_xcache = mixlib.qa_to_row_major(cache.q_xcache, M, self.in_features)
            cache.q_layout = 0
        return cache.q_xcache

    def _zeros(self, M, device):
        """`self.cache.zeros` of the reference (Cache.py: an [inputdim, 36864] zero matrix handed to dequantizeInt8 as the addend when
        there are no outliers): here an [M, N] block, allocated on first use and kept."""
        z = getattr(self, "_zero_addend", None)
        if z is None or z.shape[0] < M or z.device != device:
            z = self._zero_addend = torch.zeros((M, self.out_features), dtype=torch.float16, device=device)
        return z[:M]

    def _cached_q(self, cache, M):
        """(q_xcache, layout) as THIS consumer's GEMM can read it.  The producer chose the image for one consumer's N (the
        fragment-major one only where the weight-streaming skinny GEMM serves that shape); a consumer that shares the cache with
        another N -- or runs after a selection knob changed -- may not be served by that kernel: it gets the row-major image
        back (one small permute, cached) instead of an MIXQ_E_SHAPE in the middle of a decode step."""
        lay = getattr(cache, "q_layout", 0)
        if lay and mixlib.qa_layout(M, self.out_features, self.in_features) != lay:
            cache.q_xcache = mixlib.qa_to_row_major(cache.q_xcache, M, self.in_features)
            cache.q_layout = lay = 0
        return cache.q_xcache, lay

    @torch.no_grad()
    def forward_without_preconditionFusedSilu(self, x, cache, mul=None):
        """linear.py:288-375: the gate projection of an MLP.  Re-uses the activation that the up projection has just
        quantised (cache.q_xcache / x_scale / activation_outliers / ind) -- no second quantisation pass -- adopts the
        outlier columns the up projection may have added, and applies SiLU in the GEMM epilogue.  ``mul`` (MI355X
        extension, bit = 8): a [.., N] fp16 tensor multiplied into the rounded result inside the same epilogue --
        `gate *= up` without another pass over [M, N]."""
        inputs = x.reshape(-1, x.shape[-1])
        M = inputs.shape[0]
        if self.forward_without_precondition_len != cache.ind.shape[0]:
            if cache.ind.shape[0]:
                ind = cache.new_ind
                if self.bit == 8:
                    weight_cache = dequant_weight_columns(self.q_weight, self.scale_col, ind)
                else:
                    weight_cache = mixlib.unpack_int4_to_fp16(self.q_weight, ind) * self.scale_col.T
                if self.ind.shape[0] == 0:
                    self.weight_cache = weight_cache
                else:
                    self.weight_cache = torch.hstack((self.weight_cache, weight_cache))
            self.ind = cache.ind
            self.forward_without_precondition_len = self.ind.shape[0]
        y = outlier_product(cache.activation_outliers, self.weight_cache) if self.ind.shape[0] else None
        fused_mul = mul is not None and self.bit == 8 and self.bias is None and self.arch != 9
        if self.arch == 9 and self.bit == 8:   # :317-324
            y1 = mixlib.dequantizeInt8Silu(mixlib.gemm(self._row_major_q(cache, M), self.q_weight, M, self.out_features, self.in_features),
                                           cache.x_scale, self.scale_col, y if y is not None else self._zeros(M, inputs.device), 8, M,
                                           self.out_features)
        elif fused_mul:
            q, lay = self._cached_q(cache, M)
            y1 = mixlib.int8FusedDequantizeSiluMul(q, self.q_weight, cache.x_scale, self.scale_col, y,
                                                   mul.reshape(M, self.out_features), M, self.out_features,
                                                   self.in_features, lay)
        elif self.bit == 8:
            q, lay = self._cached_q(cache, M)
            y1 = mixlib.int8FusedDequantizeSilu(q, self.q_weight, cache.x_scale, self.scale_col, y, M,
                                                self.out_features, self.in_features, lay)
        else:
            if y is None:
                raise RuntimeError("int4 mod should have outliers !")  # :364
            y1 = mixlib.int4FusedDequantizeSilu(cache.q_xcache, self.q_weight, cache.x_scale, self.scale_col, y, M,
                                                self.out_features, self.in_features // 2, self._q_weight_i8(M))
        if self.bias is not None:
            y1 += self.bias
        if mul is not None and not fused_mul:
            y1 = y1.reshape(cache.shape)
            y1 *= mul
        return y1.reshape(cache.shape)


class FasterTransformerRMSNorm:
    """MixQ/src/mixquant/modules/fused/norm.py:6-40: RMSNorm that, when ``next_layer`` is a MixLinear_GEMM, also
    extracts (+zeroes) that layer's outlier columns and quantises the rows in the same pass over the hidden state
    (`mixq_rmsnorm_extract_quant`), leaving q_xcache / x_scale / activation_outliers in the cache."""

    def __init__(self, weight, eps=1e-6, cache=None):
        self.weight = weight.to(torch.float16)
        self.variance_epsilon = eps
        self.cache = cache
        self.next_layer = None

    @torch.no_grad()
    def forward(self, x):
        output = torch.empty_like(x)
        if self.next_layer is None:
            mixlib.layernorm_forward_cuda(x, self.weight, output, self.variance_epsilon)
        elif self.next_layer.bit == 8:
            # decode batches: the int8 rows in the image the next layer's GEMM reads fastest (mixlib.qa_layout; the gate
            # projection that re-uses the cache has the same N).  A dynamic outlier set re-quantises row-major later.
            c = x.shape[-1]
            m = x.numel() // c
            n_out = int(self.next_layer.ind.shape[0])
            lay = mixlib.qa_layout(m, self.next_layer.out_features, c) if (n_out % 8 == 0 and n_out <= 128 and
                                                                            not self.next_layer.add_outliers) else 0
            self.cache.q_layout = lay
            self.cache.activation_outliers, self.cache.q_xcache = mixlib.layernorm_forward_cuda_extract_outliers(
                x, self.weight, output, self.variance_epsilon, self.next_layer.ind, self.cache.x_scale, lay)
        elif self.next_layer.bit == 4:
            self.cache.activation_outliers, self.cache.q_xcache = mixlib.layernorm_forward_cuda_extract_outliers_int4(
                x, self.weight, output, self.variance_epsilon, self.next_layer.ind, self.cache.x_scale)
        else:
            raise NotImplementedError
        return output

    __call__ = forward


class MixLlamaMLP:
    """MixQ/src/mixquant/modules/fused/mlp.py:37-68: down(silu(gate(x)) * up(x)) where the gate projection re-uses the
    up projection's quantised activation and has SiLU fused into its GEMM epilogue."""

    def __init__(self, gate_proj, down_proj, up_proj, MixGemmCache=None):
        self.down_proj_, self.gate_proj_, self.up_proj_ = down_proj, gate_proj, up_proj
        self.out_features = down_proj.out_features
        self.MLPCache = MixGemmCache

    @torch.no_grad()
    def forward(self, x):
        up_output = self.up_proj_(x, self.MLPCache)            # fused producer: the norm in front filled the cache
        # `gate_output *= up_output` (mlp.py:63) rides in the gate GEMM's epilogue: same bits, one pass less over [M, N]
        gate_output = self.gate_proj_.forward_without_preconditionFusedSilu(x, self.MLPCache, mul=up_output)
        return self.down_proj_(gate_output, None, True)

    __call__ = forward

"""Host-side mirror of the reference's operator boundary (the reference tree's top-level ``plugin.py``) for MI355X.

Same names, argument meaning and tensor contract as the reference:

* ``_load_lianxiang_plugin_lib()``  -- dlopen + ``initOpenAiTritonPlugins(None, b"tensorrt_llm")`` (plugin.py:34-43)
* ``mixgemm(m, n, k, inputs)``      -- create plugin ("MixQ","1","tensorrt_llm") with fields m/n/k and run it on the
                                      7 inputs (plugin.py:52-79).  The reference adds a layer to a TensorRT graph; here
                                      the plugin is executed eagerly on the current HIP stream (no TensorRT on MI355X).
* ``MixQLinear``                    -- parameter layout of plugin.py:86-135 and forward of :137-162.

PyTorch appears only as plumbing (device memory, current stream).  All compute is in libmixq_mi355x.so; there is no
fallback path.
"""
import ctypes
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import PluginField, TensorDesc

TRT_LLM_PLUGIN_NAMESPACE = "tensorrt_llm"   # plugin.py:29
LAYER_NAME = "MixQLayer"                    # plugin.py:30
NUM_OUTLIERS = 128                          # plugin.py:102-105


def _load_lianxiang_plugin_lib():
    handle = _lib.load()
    assert handle.initOpenAiTritonPlugins(None, TRT_LLM_PLUGIN_NAMESPACE.encode("utf-8"))
    return handle


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device) -> ctypes.c_void_p:
    if _raw_stream is not None and device.index is not None:   # ~0.2 us instead of ~1.8 us per call
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class MixQPlugin:
    """Owner of one ``mixq_handle`` (the reference's MixQPlugin object, TsinghuaMixQPlugin.h:34-89)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("MixQ plugin creation failed")
        self._h = ctypes.c_void_p(handle)
        self._lib = _lib.load()
        self._ws = {}
        self._retired = []

    # -- creator path (MixQPluginCreator::createPlugin / deserializePlugin) --
    @classmethod
    def create(cls, m: int, n: int, k: int) -> "MixQPlugin":
        lib = _load_lianxiang_plugin_lib()
        assert lib.mixq_registry_has_creator(b"MixQ", b"1", TRT_LLM_PLUGIN_NAMESPACE.encode()), \
            "plugin creator MixQ/1 not registered"
        vals = [np.array([v], np.int32) for v in (m, n, k)]
        fields = (PluginField * 3)()
        for f, name, v in zip(fields, (b"m", b"n", b"k"), vals):
            f.name, f.data, f.type, f.length = name, v.ctypes.data, _lib.MIXQ_FIELD_INT32, 1
        p = cls(lib.mixq_create_from_fields(fields, 3))
        lib.mixq_set_namespace(p._h, TRT_LLM_PLUGIN_NAMESPACE.encode())
        lib.mixq_initialize(p._h)
        return p

    @classmethod
    def deserialize(cls, blob: bytes) -> "MixQPlugin":
        lib = _load_lianxiang_plugin_lib()
        buf = ctypes.create_string_buffer(blob, len(blob))
        return cls(lib.mixq_deserialize(buf, len(blob)))

    def serialize(self) -> bytes:
        n = self._lib.mixq_serialization_size(self._h)
        buf = ctypes.create_string_buffer(n)
        self._lib.mixq_serialize(self._h, buf)
        return buf.raw

    def clone(self) -> "MixQPlugin":
        return MixQPlugin(self._lib.mixq_clone(self._h))

    @property
    def mnk(self):
        m, n, k = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._lib.mixq_get_mnk(self._h, ctypes.byref(m), ctypes.byref(n), ctypes.byref(k))
        return m.value, n.value, k.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mixq_terminate(self._h)
                self._lib.mixq_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def workspace_size(self, max_m: int, n: int, k: int) -> int:
        return int(self._lib.mixq_workspace_size(self._h, max_m, n, k))

    def _workspace(self, device, nbytes: int) -> torch.Tensor:
        # the caller (TensorRT in the reference) owns the workspace; enqueue never allocates
        key = (device.type, device.index)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                # a HIP graph captured earlier holds the old pointer (and the library parks split-K partial sums in it
                # on replay): superseded buffers stay alive for the life of the plugin, they are never handed back
                self._retired.append(ws)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def enqueue(self, inputs: List[torch.Tensor], output: Optional[torch.Tensor] = None,
                workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """MixQPlugin::enqueue (TsinghuaMixQPlugin.h:53-54) on the current stream of ``inputs[0].device``.

        ``inputs`` are the 7 tensors of plugin.py:141-151, every one declared fp16 exactly as TensorRT sees them
        (int8 / int32 payloads travel as fp16 views)."""
        assert len(inputs) == 7, "MixQ takes 7 inputs"
        A = inputs[0]
        if not A.is_cuda:
            raise _lib.MixQLibraryError("MixQ enqueue needs device tensors (no CPU path exists)")
        for t in inputs:
            assert t.dtype == torch.float16 and t.is_contiguous() and t.device == A.device
        in_desc = (TensorDesc * 7)(*[TensorDesc.make(t.shape) for t in inputs])
        out_desc = TensorDesc()
        _lib.check(self._lib.mixq_get_output_dimensions(self._h, 0, in_desc, 7, ctypes.byref(out_desc)),
                   "getOutputDimensions")
        out_shape = [out_desc.d[i] for i in range(out_desc.nbDims)]
        if output is None:
            output = torch.empty(out_shape, dtype=torch.float16, device=A.device)
        else:
            assert list(output.shape) == out_shape and output.dtype == torch.float16 and output.is_contiguous()
        K = A.shape[-1]
        M = A.numel() // K if K else 0
        N = inputs[1].shape[0]
        if workspace is None:
            workspace = self._workspace(A.device, self.workspace_size(max(M, 1), N, K))
        in_ptrs = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in inputs])
        out_ptrs = (ctypes.c_void_p * 1)(output.data_ptr())
        with torch.cuda.device(A.device):  # the library keys its per-device state on the CURRENT device
            rc = self._lib.mixq_enqueue(self._h, in_desc, ctypes.byref(out_desc), in_ptrs, out_ptrs,
                                        ctypes.c_void_p(workspace.data_ptr()), _stream_ptr(A.device))
        _lib.check(rc, "MixQPlugin::enqueue")
        return output


def mixgemm(m: int, n: int, k: int, inputs: List[torch.Tensor]) -> torch.Tensor:
    """plugin.py:52-79.  Creates the plugin through the registry exactly as the reference does and executes it."""
    plugin = MixQPlugin.create(m, n, k)
    return plugin.enqueue(inputs)


class MixQLinear:
    """plugin.py:86-162.  Parameters keep the reference's declared shapes/dtypes (all fp16 carriers):

    ``weight`` fp16 [N, K/2] (= int8 [N,K]) | ``fp_weight`` fp16 [N,128] | ``fp_ind`` fp16 [256] (= int32 [128]) |
    ``qweight`` fp16 [K, N/2] (= uint8 [K,N], EETQ-interleaved) | ``weights_scaling_factor`` fp16 [N] | optional ``bias``.
    """

    def __init__(self, in_features: int, out_features: int, bias: bool = False, dtype=None, tp_group=None,
                 tp_size: int = 1, gather_output: bool = True, device="cuda"):
        self.in_features = in_features
        self.out_features = out_features // tp_size          # plugin.py:97 (rows of W are sharded)
        self.tp_size, self.tp_group, self.gather_output = tp_size, tp_group, gather_output
        dev = torch.device(device)
        f16 = dict(dtype=torch.float16, device=dev)
        self.weight = torch.zeros(self.out_features, in_features // 2, **f16)
        self.fp_weight = torch.zeros(self.out_features, NUM_OUTLIERS, **f16)
        self.fp_ind = torch.zeros(NUM_OUTLIERS * 2, **f16)
        self.qweight = torch.zeros(in_features, self.out_features // 2, **f16)
        self.weights_scaling_factor = torch.zeros(self.out_features, **f16)
        self.bias = torch.zeros(self.out_features, dtype=dtype or torch.float16, device=dev) if bias else None
        self._plugin = None
        self.peer_gather = None   # optional parallel.PeerGather: one-sided peer writes instead of the RCCL all-gather
        self.peer_gather_alias = False  # True: forward() returns a VIEW of the gather buffer (valid until two more gathers)
        self.weight_image = None  # optional mixlib.WeightImage (prepare_decode_batches): a streaming-friendly copy of `weight`

    def prepare_decode_batches(self, enable: bool = True):
        """MI355X extension (include/mixq.h ``mixq_weight_image_*``): register a fragment-major copy of ``weight`` so that decode
        batches (5 .. 64 rows) stream it with contiguous reads -- same bits, -10..-15 % per call, N * K more bytes of HBM.  Call
        after the weights are loaded; ``enable=False`` drops the copy."""
        from . import mixlib
        if self.weight_image is not None:
            self.weight_image.close()
            self.weight_image = None
        if enable:
            self.weight_image = mixlib.WeightImage(self.weight.view(torch.int8))
        return self

    def load(self, packed: dict):
        """Install the tensors produced by ``pack.pack_linear_weights`` (true dtypes) as fp16 carriers."""
        dev = self.weight.device

        def carrier(a, shape):
            t = torch.from_numpy(np.ascontiguousarray(a)).view(torch.float16).reshape(shape)
            return t.to(dev)

        N, K = self.out_features, self.in_features
        had_image = self.weight_image is not None
        self.prepare_decode_batches(False)               # (a registered copy belongs to the OLD weight tensor)
        self.weight = carrier(packed["weight"], (N, K // 2))
        self.fp_weight = carrier(packed["fp_weight"], (N, NUM_OUTLIERS))
        self.fp_ind = carrier(packed["fp_ind"].astype(np.int32), (NUM_OUTLIERS * 2,))
        self.qweight = carrier(packed["qweight"], (K, N // 2))
        self.weights_scaling_factor = carrier(packed["weights_scaling_factor"], (N,))
        if self.bias is not None and packed.get("bias") is not None:
            b = torch.from_numpy(np.ascontiguousarray(packed["bias"])).to(dev)
            assert tuple(b.shape) == (N,), f"bias {tuple(b.shape)} vs this rank's {N} output features"
            self.bias = b.to(self.bias.dtype)
        if had_image:
            self.prepare_decode_batches(True)
        return self

    def forward(self, A: torch.Tensor) -> torch.Tensor:
        if self._plugin is None:
            self._plugin = MixQPlugin.create(A.shape[0], self.out_features, self.in_features)
        if self.tp_size > 1 and self.gather_output and self.peer_gather is not None and self.bias is None:
            # operator + all-gather in one pass: the GEMM's store path writes this rank's column block into every rank's
            # buffer (mixq_enqueue_tp); shapes that do not take the 256 x 256 ping-pong kernel fall through to
            # enqueue + push below
            g = self.peer_gather.enqueue_gather(self._plugin, [A.contiguous(), self.weight, self.weights_scaling_factor,
                                                               self.fp_weight, self.fp_ind, self.qweight,
                                                               self.weights_scaling_factor])
            if g is not None:
                g = g.reshape(*A.shape[:-1], self.out_features * self.tp_size)
                return g if self.peer_gather_alias else g.clone()
        x = self._plugin.enqueue([A, self.weight, self.weights_scaling_factor, self.fp_weight, self.fp_ind,
                                  self.qweight, self.weights_scaling_factor])   # plugin.py:141-151
        if self.bias is not None:
            # plugin.py:158-160 adds the bias after its collective; the bias of a row-sharded layer is this rank's
            # [N/tp] slice (parallel.shard_packed), so it is added to the shard BEFORE the gather -- the same fp16
            # additions, element for element, as adding the full bias to the gathered output
            x = x + self.bias.to(x.dtype)
        if self.tp_size > 1 and self.gather_output:
            # The reference calls allreduce here (plugin.py:155-156), which is shape-wrong for an N-split and is
            # guarded by assert tp_size==1 upstream; the row-sharded operator needs ONE all-gather of the fp16 output.
            if self.peer_gather is not None:   # the block lands in its column block of every rank's [M, N] buffer
                lead = x.shape[:-1]
                # gather() raises PeerGatherTimeout (sticky, no device sync) if an earlier wait gave up on a peer
                x = self.peer_gather.gather(x.reshape(-1, x.shape[-1])).reshape(*lead, self.out_features * self.tp_size)
                if not self.peer_gather_alias:
                    # the gathered tensor lives in one of TWO IPC buffers and is overwritten two gather calls later; a
                    # caller that keeps it (residual, KV projection) gets its own copy unless it opted into the alias
                    x = x.clone()
            else:
                from .parallel import all_gather_columns
                x = all_gather_columns(x, self.tp_group, self.tp_size)
        return x

    __call__ = forward

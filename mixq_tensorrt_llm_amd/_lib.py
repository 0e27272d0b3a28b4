"""ctypes binding of libmixq_mi355x.so (the C ABI declared in include/mixq.h).

The library is built in-tree by ``mixq_tensorrt_llm_amd/csrc/build.sh`` (``__graft_entry__.build()``).  There is no
CPU or PyTorch fallback: if the shared object is missing, every entry point raises ``MixQLibraryError``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmixq_mi355x.so")

MIXQ_MAX_DIMS = 8
MIXQ_TYPE_HALF = 1
MIXQ_FORMAT_LINEAR = 0
MIXQ_FIELD_INT32 = 3

ERRORS = {0: "ok", 1: "bad argument", 2: "unsupported shape", 3: "pointer not 16-byte aligned", 4: "HIP error",
          5: "workspace missing"}


class MixQLibraryError(RuntimeError):
    pass


class MixQError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = int(code)
        super().__init__(f"{where}: mixq error {self.code} ({ERRORS.get(self.code, 'unknown')})")


class TensorDesc(ctypes.Structure):
    """POD mirror of nvinfer1::PluginTensorDesc (include/mixq.h: mixq_tensor_desc)."""
    _fields_ = [("nbDims", ctypes.c_int32), ("d", ctypes.c_int64 * MIXQ_MAX_DIMS), ("type", ctypes.c_int32),
                ("format", ctypes.c_int32), ("scale", ctypes.c_float)]

    @classmethod
    def make(cls, shape, dtype=MIXQ_TYPE_HALF):
        t = cls()
        t.nbDims = len(shape)
        for i, s in enumerate(shape):
            t.d[i] = int(s)
        t.type, t.format, t.scale = dtype, MIXQ_FORMAT_LINEAR, 1.0
        return t


class TpEpilogue(ctypes.Structure):
    """include/mixq.h: mixq_tp_epilogue."""
    _fields_ = [("ndst", ctypes.c_int32), ("n_total", ctypes.c_int32), ("col0", ctypes.c_int32), ("seq", ctypes.c_uint32),
                ("dst_bases", ctypes.c_void_p * 8), ("dst_flags", ctypes.c_void_p * 8), ("counters", ctypes.c_void_p)]


class PluginField(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("type", ctypes.c_int32),
                ("length", ctypes.c_int32)]


_vp, _i, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t

# name -> (restype, argtypes).  Must list every symbol of include/mixq.h (tests/test_abi.py cross-checks).
SIGNATURES = {
    "initOpenAiTritonPlugins": (ctypes.c_bool, [_vp, ctypes.c_char_p]),
    "mixq_registry_has_creator": (_i, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]),
    "mixq_plugin_type": (ctypes.c_char_p, []),
    "mixq_plugin_version": (ctypes.c_char_p, []),
    "mixq_create": (_vp, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "mixq_create_from_fields": (_vp, [ctypes.POINTER(PluginField), ctypes.c_int32]),
    "mixq_get_field_names": (ctypes.POINTER(PluginField), [ctypes.POINTER(ctypes.c_int32)]),
    "mixq_deserialize": (_vp, [_vp, _sz]),
    "mixq_serialization_size": (_sz, [_vp]),
    "mixq_serialize": (None, [_vp, _vp]),
    "mixq_clone": (_vp, [_vp]),
    "mixq_destroy": (None, [_vp]),
    "mixq_initialize": (_i, [_vp]),
    "mixq_terminate": (None, [_vp]),
    "mixq_get_mnk": (_i, [_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                          ctypes.POINTER(ctypes.c_int32)]),
    "mixq_set_namespace": (_i, [_vp, ctypes.c_char_p]),
    "mixq_get_namespace": (ctypes.c_char_p, [_vp]),
    "mixq_get_nb_outputs": (_i, [_vp]),
    "mixq_get_output_dimensions": (_i, [_vp, _i, ctypes.POINTER(TensorDesc), _i, ctypes.POINTER(TensorDesc)]),
    "mixq_supports_format_combination": (_i, [_vp, _i, ctypes.POINTER(TensorDesc), _i, _i]),
    "mixq_get_output_data_type": (_i, [_vp, _i]),
    "mixq_workspace_size": (_sz, [_vp, _i64, _i64, _i64]),
    "mixq_reference_workspace_size": (_sz, [_i64, _i64, _i64]),
    "mixq_enqueue_scratch_size": (_sz, [_i64, _i64, _i64]),
    "mixq_enqueue": (_i, [_vp, ctypes.POINTER(TensorDesc), ctypes.POINTER(TensorDesc), ctypes.POINTER(_vp),
                          ctypes.POINTER(_vp), _vp, _vp]),
    "mixq_weight_image_bytes": (_sz, [_i64, _i64]),
    "mixq_weight_image_register": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "mixq_weight_image_unregister": (_i, [_vp]),
    "mixq_weight_image_verify": (_i, [_vp, _vp]),
    "mixq_weight_image_stale_count": (_i, []),
    "mixq_enqueue_profiled": (_i, [_vp, ctypes.POINTER(TensorDesc), ctypes.POINTER(TensorDesc), ctypes.POINTER(_vp),
                                   ctypes.POINTER(_vp), _vp, _vp, _vp, _vp]),
    "mixq_int8quant": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "mixq_extract_outliers": (_i, [_i, _i, _vp, _vp, _vp, _i, _vp]),
    "mixq_extract_outliers_set_zero": (_i, [_i, _i, _vp, _vp, _vp, _i, _vp]),
    "mixq_quant_extract": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mixq_int4quant": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "mixq_rmsnorm_extract_quant4": (_i, [_i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _i, _vp, _vp, _vp, _vp]),
    "mixq_int4_fused_workspace_size": (ctypes.c_size_t, [_i, _i, _i]),
    "mixq_int4_fused_dequantize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_int4_fused_dequantize_silu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_int4_fused_dequantize_w8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mixq_int4_linear_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mixq_unpack_int4_to_fp16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_unpack_int4_to_int8": (_i, [_vp, _vp, ctypes.c_size_t, _vp]),
    "mixq_find_outliers_workspace_size": (ctypes.c_size_t, [_i]),
    "mixq_find_outliers": (_i, [_vp, _i, _i, ctypes.c_float, _vp, _vp, _vp, _i, _vp]),
    "mixq_dequant_weight_columns": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "mixq_int8_quantize_with_scale": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "mixq_rmsnorm": (_i, [_i, _i, _vp, _vp, _vp, ctypes.c_float, _vp]),
    "mixq_rmsnorm_extract_quant": (_i, [_i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _i, _vp, _vp, _vp, _vp]),
    "mixq_int8_fused_dequantize": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_int8_fused_dequantize_silu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_gemm_scratch_size": (ctypes.c_size_t, [_i, _i, _i]),
    "mixq_gemm_scratch_bound": (ctypes.c_size_t, []),
    "mixq_gemm_mixed_scratch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "mixq_int8_fused_dequantize_silu_mul": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mixq_gemm_mixed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mixq_mixlinear_forward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, ctypes.c_size_t, _vp]),
    "mixq_qa_layout": (_i, [_i, _i, _i]),
    "mixq_qa_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "mixq_quant_extract_layout": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mixq_rmsnorm_extract_quant_layout": (_i, [_i, _i, _vp, _vp, _vp, ctypes.c_float, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "mixq_int8_fused_dequantize_layout": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mixq_gemm_mixed_layout": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "mixq_gemm_s8s8s32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "mixq_gemm_fp16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "mixq_dequantization": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mixq_dequantization_silu": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mixq_w8a16_gemm_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mixq_w8a16_gemm_workspace_size": (ctypes.c_size_t, [_i, _i, _i]),
    "mixq_w8a16_gemm_forward_ws": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "mixq_tp_buffer_alloc": (_i, [_sz, _i, ctypes.POINTER(_vp), _vp]),
    "mixq_tp_buffer_open": (_i, [_vp, ctypes.POINTER(_vp)]),
    "mixq_tp_buffer_close": (_i, [_vp]),
    "mixq_tp_buffer_free": (_i, [_vp]),
    "mixq_tp_status_alloc": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "mixq_tp_status_free": (_i, [_vp]),
    "mixq_tp_push_columns": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i, _i, ctypes.c_uint32, _i,
                                  _vp, _vp]),
    "mixq_tp_wait": (_i, [_vp, _i, _i, _i, ctypes.c_uint32, _vp, _i, ctypes.c_uint32, _vp]),
    "mixq_tp_arrive": (_i, [ctypes.POINTER(_vp), _vp, _i, _vp, _vp, _i, ctypes.c_uint32, _vp]),
    "mixq_tp_push_columns_seq": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mixq_tp_wait_seq": (_i, [_vp, _i, _vp, _vp, _i, ctypes.c_uint32, _vp]),
    "mixq_tp_fused_supported": (_i, [_i64, _i64, _i64]),
    "mixq_tp_flag_words": (_i, [_i64]),
    "mixq_enqueue_tp": (_i, [_vp, ctypes.POINTER(TensorDesc), ctypes.POINTER(_vp), _vp, ctypes.POINTER(TpEpilogue), _vp]),
    "mixq_preprocess_weights_int8": (_i, [_vp, _vp, _sz, _sz]),
    "mixq_unprocess_weights_int8": (_i, [_vp, _vp, _sz, _sz]),
    "mixq_debug_set_gemm_variant": (None, [_i]),
    "mixq_debug_reset": (None, []),
    "mixq_debug_knobs_enabled": (_i, []),
    "mixq_debug_set_stamp_buffer": (None, [_vp]),
    "mixq_debug_set_quant_stamp_buffer": (None, [_vp]),
    "mixq_debug_last_gemm_kernel": (ctypes.c_char_p, []),
    "mixq_describe_plan": (_i, [_i64, _i64, _i64, _i, ctypes.c_char_p, ctypes.c_size_t]),
    "mixq_version": (ctypes.c_char_p, []),
    "mixq_abi_version": (_i, []),
    "mixq_error_string": (ctypes.c_char_p, [_i]),
}

_lib = None
ABI_VERSION = 4   # include/mixq.h MIXQ_ABI_VERSION


def load():
    """dlopen the in-tree library (RTLD_GLOBAL like the reference loader, plugin.py:34-43) and type every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MixQLibraryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or mixq_tensorrt_llm_amd/csrc/build.sh).  There is no CPU fallback for the MixQ operator.")
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # e.g. libamdhip64 not found
        raise MixQLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MixQLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    # the measurement knobs are ignored by the library unless MIXQ_DEBUG_KNOBS=1 was in the environment (include/mixq.h): make a
    # forgotten opt-in loud on the Python side instead of a silently unchanged selection
    raw_set = lib.mixq_debug_set_gemm_variant

    def set_gemm_variant(v):
        if not lib.mixq_debug_knobs_enabled():
            raise MixQLibraryError("mixq_debug_set_gemm_variant: measurement knobs are off in this process -- export "
                                   "MIXQ_DEBUG_KNOBS=1 before the first knob call (tests/conftest.py and the scripts under tools/ do)")
        raw_set(v)
    lib.mixq_debug_set_gemm_variant = set_gemm_variant
    if lib.mixq_abi_version() != ABI_VERSION:   # a stale .so next to newer bindings would take shifted arguments
        raise MixQLibraryError(f"{LIB_PATH} has ABI revision {lib.mixq_abi_version()}, these bindings need {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(rc, where):
    if rc != 0:
        raise MixQError(rc, where)

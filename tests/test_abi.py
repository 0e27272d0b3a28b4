"""CPU: the C-ABI library loads, exports every symbol include/mixq.h declares, and its host-only entry points
(registry, plugin lifecycle, shape negotiation, workspace sizing, qweight layout importer) behave like the reference's
host code.  No compute launches here (no GPU in the build container)."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT

from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import PluginField, TensorDesc


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mixq.h")).read()
    return sorted(set(re.findall(r"MIXQ_API\s+[\w\s\*]+?\b(\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_typed(lib):
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mixq.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES out of sync with include/mixq.h"


def test_registry_matches_reference_loader(lib):
    # plugin.py:34-43: initOpenAiTritonPlugins(None, b"tensorrt_llm") must return true, idempotently
    assert lib.mixq_registry_has_creator(b"MixQ", b"1", b"never_registered_ns") == 0
    assert lib.initOpenAiTritonPlugins(None, b"tensorrt_llm") is True
    assert lib.initOpenAiTritonPlugins(None, b"tensorrt_llm") is True
    assert lib.mixq_registry_has_creator(b"MixQ", b"1", b"tensorrt_llm") == 1
    assert lib.mixq_registry_has_creator(b"MixQ", b"2", b"tensorrt_llm") == 0
    assert lib.mixq_plugin_type() == b"MixQ" and lib.mixq_plugin_version() == b"1"


def test_creator_advertises_the_reference_field_table(lib):
    """MixQPluginCreator::getFieldNames (TsinghuaMixQPlugin.cpp:890-893) returns the table its constructor fills (:868-878): three INT32
    fields "mm", "mn", "mk" with no data and length -1 -- while createPlugin parses "m", "n", "k" (:906-919).  Both are mirrored."""
    n = ctypes.c_int32(0)
    f = lib.mixq_get_field_names(ctypes.byref(n))
    assert n.value == 3
    assert [(f[i].name, f[i].data, f[i].type, f[i].length) for i in range(3)] == [(b"mm", None, _lib.MIXQ_FIELD_INT32, -1), (b"mn", None, _lib.MIXQ_FIELD_INT32, -1),
                                                                                   (b"mk", None, _lib.MIXQ_FIELD_INT32, -1)]
    assert lib.mixq_get_field_names(None)   # a NULL count pointer is tolerated


def test_lifecycle_serialize_clone(lib):
    vals = [np.array([v], np.int32) for v in (8192, 12288, 4096)]
    fields = (PluginField * 4)()
    for f, name, v in zip(fields, (b"m", b"n", b"k"), vals):
        f.name, f.data, f.type, f.length = name, v.ctypes.data, _lib.MIXQ_FIELD_INT32, 1
    junk = np.array([7], np.int32)
    fields[3].name, fields[3].data, fields[3].type, fields[3].length = b"mm", junk.ctypes.data, 3, 1  # ignored
    h = lib.mixq_create_from_fields(fields, 4)
    assert h
    assert lib.mixq_serialization_size(h) == 12           # TsinghuaMixQPlugin.cpp:808-811
    buf = ctypes.create_string_buffer(12)
    lib.mixq_serialize(h, buf)
    assert struct.unpack("<iii", buf.raw) == (8192, 12288, 4096)
    h2 = lib.mixq_deserialize(buf, 12)
    h3 = lib.mixq_clone(h2)
    m, n, k = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.mixq_get_mnk(h3, ctypes.byref(m), ctypes.byref(n), ctypes.byref(k)) == 0
    assert (m.value, n.value, k.value) == (8192, 12288, 4096)
    assert lib.mixq_deserialize(buf, 11) is None           # short blob
    assert lib.mixq_set_namespace(h, b"tensorrt_llm") == 0 and lib.mixq_get_namespace(h) == b"tensorrt_llm"
    assert lib.mixq_initialize(h) == 0
    lib.mixq_terminate(h)
    for x in (h, h2, h3):
        lib.mixq_destroy(x)
    lib.mixq_destroy(None)


def test_shape_negotiation(lib):
    h = lib.mixq_create(1, 2, 3)
    descs = (TensorDesc * 8)(
        TensorDesc.make((4, 2048, 4096)), TensorDesc.make((12288, 2048)), TensorDesc.make((12288,)),
        TensorDesc.make((12288, 128)), TensorDesc.make((256,)), TensorDesc.make((4096, 6144)),
        TensorDesc.make((12288,)), TensorDesc.make((4, 2048, 12288)))
    out = TensorDesc()
    assert lib.mixq_get_output_dimensions(h, 0, descs, 7, ctypes.byref(out)) == 0
    assert [out.d[i] for i in range(out.nbDims)] == [4, 2048, 12288]     # .cpp:244-261
    assert lib.mixq_get_output_dimensions(h, 1, descs, 7, ctypes.byref(out)) != 0
    for pos in range(8):
        assert lib.mixq_supports_format_combination(h, pos, descs, 7, 1) == 1
    descs[3].type = 0  # float
    assert lib.mixq_supports_format_combination(h, 3, descs, 7, 1) == 0
    assert lib.mixq_get_nb_outputs(h) == 1 and lib.mixq_get_output_data_type(h, 0) == _lib.MIXQ_TYPE_HALF
    lib.mixq_destroy(h)


def test_workspace_sizes_are_size_t_clean(lib):
    h = lib.mixq_create(0, 0, 0)
    M, N, K = 8192, 12288, 4096
    ws = lib.mixq_workspace_size(h, M, N, K)
    need = M * K + 2 * M + 2 * 128 * M
    # the K-split exchange scratch is sized for THIS N, K and every M <= maxM (it was a flat 56 MiB in round 1):
    # at most one 256-KiB slot per CU + the hand-over words, and nothing at all for most shapes
    bound = lib.mixq_gemm_scratch_bound()
    assert bound == 256 * 262144 + 16384   # one 256-KiB slot per CU (S shares of a tile's 1/S-th) + the hand-over words
    worst = max(lib.mixq_gemm_scratch_size(m, N, K) for m in range(256, M + 1, 256))
    assert need + worst <= ws <= need + worst + 5 * 128 + 128 and worst <= bound
    assert lib.mixq_workspace_size(h, 64, 4096, 4096) <= 64 * 4096 + 2 * 64 + 2 * 64 * 128 + 5 * 128  # no split form: no scratch
    assert lib.mixq_workspace_size(h, M, 0, K) >= need + bound           # N unknown: the shape-independent bound
    xsplit = 16384 + 256 * 64 * 64 * 4            # below 256 rows: the small-tile form's scratch (256 workgroups x 16 KiB)
    deep = 16384 + 96 * 2 * 128 * 128 * 4         # round 5: 97..128 rows on this shape take the deep form, 2 workgroups per 128 x 128 tile
    assert lib.mixq_workspace_size(h, 128, N, K) <= 128 * K + 2 * 128 + 2 * 128 * 128 + max(xsplit, deep) + 5 * 128 + 128
    assert lib.mixq_workspace_size(h, 96, N, K) <= 96 * K + 2 * 96 + 2 * 96 * 128 + xsplit + 5 * 128 + 128
    assert lib.mixq_workspace_size(h, 4, N, K) <= 4 * K + 8 + 8 * 128 + 4 * 128 + 128     # decode: none
    assert lib.mixq_gemm_scratch_size(1024, 4096, 11008) == 16384 + 64 * 4 * 262144   # 4 workgroups per tile, a 256-KiB slot each
    assert lib.mixq_gemm_scratch_size(2048, 4096, 4096) == 0          # 256 tiles of 128 x 256: no K split, no scratch
    assert lib.mixq_gemm_scratch_size(2048, 4096, 12288) == 16384 + 128 * 2 * 262144  # longer K: 2 workgroups per tile
    assert lib.mixq_gemm_scratch_size(8192, 12288, 4096) == 0 and lib.mixq_gemm_scratch_size(64, 4096, 4096) == 0
    # 1M tokens x 11008: the reference's int arithmetic overflows here
    big = lib.mixq_workspace_size(h, 1 << 20, 4096, 11008)
    assert big > (1 << 20) * 11008 > 2**31
    assert lib.mixq_reference_workspace_size(M, N, K) == max(M * K + 2 * M + 2 * K * N, 16 * M * N)
    assert lib.mixq_reference_workspace_size(0, 0, 0) == 33554432
    lib.mixq_destroy(h)


@pytest.mark.parametrize("N,K", [(7168, 7168), (6144, 6144), (4096, 5120), (12288, 4096), (4096, 11008), (1024, 28672),
                                 (3584, 18944), (4608, 3584), (8192, 28672), (5120, 13824)])
@pytest.mark.parametrize("maxM", [255, 256, 1024, 2048, 3000])
def test_workspace_sized_once_covers_every_smaller_m(lib, N, K, maxM):
    """the split plans are not monotone in M (the 128 x 256 tiles gate the K split through ceil(M / 128)),
    so a caller that sizes the workspace once for maxM (getWorkspaceSize, TsinghuaMixQPlugin.cpp:351-378) must still be
    covered for EVERY M <= maxM: the per-M carve of enqueue_impl, byte for byte, against the bound."""
    h = lib.mixq_create(0, 0, 0)
    ws = lib.mixq_workspace_size(h, maxM, N, K)

    def up(v, a=128):
        return (v + a - 1) // a * a

    def qa(m):   # qA region: row-major, or (decode batches) whole 16-row tiles x whole 64-byte k-steps of the fragment-major image
        return max(m * K, up(m, 16) * up(K, 64)) if 4 < m <= 64 else m * K

    worst = 0
    for m in range(5, maxM + 1):
        scratch = lib.mixq_enqueue_scratch_size(m, N, K)
        carve = 127 + up(qa(m)) + up(2 * m) + up(2 * 128 * m) + scratch   # unaligned base, 128-B carve of each region
        assert carve <= ws, (m, scratch, ws)
        worst = max(worst, scratch)
    # and the bound is tight: it is the largest per-M scratch, not the shape-independent 64 MiB
    assert ws <= 128 + up(maxM * K) + up(2 * maxM) + up(2 * 128 * maxM) + up(worst) + 128
    lib.mixq_destroy(h)


def test_workspace_bound_on_random_shapes_up_to_512_rows(lib):
    """Round 4: decode batches of 129..255 rows take the K split of the 256 x 256 tiles too; the plan sees M through ceil(M / 64 | 128 | 256)
    and, inside gemm_pp128_wins, through M itself -- 120 seeded random (N, K, maxM): the workspace sized once covers every smaller M."""
    import random
    rng = random.Random(5)
    h = lib.mixq_create(0, 0, 0)

    def up(v, a=128):
        return (v + a - 1) // a * a

    for _ in range(120):
        N, K = rng.randrange(16, 2000) * 16, rng.randrange(8, 1900) * 16
        maxM = rng.choice([130, 160, 191, 192, 200, 255, 256, 300, 384, 512])
        ws = lib.mixq_workspace_size(h, maxM, N, K)
        for m in range(5, maxM + 1):
            qa = max(m * K, up(m, 16) * up(K, 64)) if m <= 64 else m * K
            carve = 127 + up(qa) + up(2 * m) + up(2 * 128 * m) + lib.mixq_enqueue_scratch_size(m, N, K)
            assert carve <= ws, (N, K, maxM, m)
    lib.mixq_destroy(h)


def test_bench_line_guard_writes_the_line_when_the_process_aborts_and_keeps_its_status():
    """bench.py guards its JSON line with a CHILD PROCESS during the tp = N leg (the operator library installs no
    signal handlers): a process that dies on SIGABRT (a GPU memory fault ends in the runtime's abort()) still gets its line out,
    and -- unlike the handler the library used to carry -- dies with its signal, so the driver sees a failed run."""
    import signal
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys\nsys.argv = ['bench.py']\nimport bench\n"
            "g = bench.LineGuard(1, b'{\"ok\": 1}\\n')\nDISARM\nos.abort()\n")
    r = subprocess.run([sys.executable, "-c", code.replace("DISARM", "")], capture_output=True, cwd=root, timeout=300)
    assert r.returncode == -signal.SIGABRT and r.stdout == b'{"ok": 1}\n', (r.returncode, r.stdout, r.stderr[-300:])
    r = subprocess.run([sys.executable, "-c", code.replace("DISARM", "g.disarm()")], capture_output=True, cwd=root, timeout=300)
    assert r.returncode == -signal.SIGABRT and r.stdout == b""   # disarmed: nothing is written
    from mixq_tensorrt_llm_amd import _lib
    assert not hasattr(_lib.load(), "mixq_debug_arm_crash_line")


def test_scratch_plan_is_not_monotone_in_m(lib):
    """The shapes ADVICE r2 named: M = 257 needs a K-split scratch that M = 1024 / 2048 does not."""
    assert lib.mixq_enqueue_scratch_size(257, 7168, 7168) > 0 == lib.mixq_enqueue_scratch_size(1024, 7168, 7168) or \
        lib.mixq_enqueue_scratch_size(257, 6144, 6144) > 0
    h = lib.mixq_create(0, 0, 0)
    for n, k in ((7168, 7168), (6144, 6144)):
        need = lib.mixq_enqueue_scratch_size(257, n, k)
        assert lib.mixq_workspace_size(h, 2048, n, k) >= 2048 * k + need
    lib.mixq_destroy(h)


def test_enqueue_rejects_bad_arguments_without_touching_the_gpu(lib):
    h = lib.mixq_create(0, 0, 0)
    d = (TensorDesc * 7)(*[TensorDesc.make((8, 64))] * 7)
    assert lib.mixq_enqueue(h, d, None, None, None, None, None) == 1
    d[0].nbDims = 0
    ins = (ctypes.c_void_p * 7)()
    outs = (ctypes.c_void_p * 1)()
    assert lib.mixq_enqueue(h, d, None, ins, outs, None, None) == 1
    assert lib.mixq_error_string(2) == b"unsupported shape"
    assert lib.mixq_gemm_s8s8s32(None, None, None, 0, 0, 16, None) == 0      # empty problem is a no-op
    assert lib.mixq_gemm_s8s8s32(None, None, None, 8, 16, 16, None) == 1     # null pointers
    assert lib.mixq_int8quant(4, 12, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None) == 2
    lib.mixq_destroy(h)


def test_qweight_layout_importer_matches_oracle(lib, oracle):
    rng = np.random.default_rng(0)
    K, N = 192, 96
    q = rng.integers(-128, 128, size=(K, N), dtype=np.int8)
    want = oracle.eetq_preprocess(q)
    got = np.empty((K, N), np.uint8)
    assert lib.mixq_preprocess_weights_int8(got.ctypes.data, q.ctypes.data, K, N) == 0
    assert np.array_equal(got, want)
    back = np.empty((K, N), np.int8)
    assert lib.mixq_unprocess_weights_int8(back.ctypes.data, got.ctypes.data, K, N) == 0
    assert np.array_equal(back, q)
    assert lib.mixq_preprocess_weights_int8(got.ctypes.data, q.ctypes.data, 100, N) == 2


def test_argument_validation_returns_codes_and_never_touches_the_device():
    """Bad arguments are rejected before any HIP call (return codes, no exceptions across the C boundary), so these run
    without a GPU: null pointers, negative sizes, K / N not multiples of 16, misaligned pointers."""
    import ctypes
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    OK, BADARG, SHAPE, ALIGN = 0, 1, 2, 3
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = ctypes.c_void_p((p + 15) & ~15)
    odd = ctypes.c_void_p(((p + 15) & ~15) + 2)
    # quantiser: null pointers, K % 8, misaligned A
    assert lib.mixq_quant_extract(4, 64, None, p16, p16, None, None, 0, 0, None) == BADARG
    assert lib.mixq_quant_extract(4, 60, p16, p16, p16, None, None, 0, 0, None) == SHAPE
    assert lib.mixq_quant_extract(4, 64, odd, p16, p16, None, None, 0, 0, None) == ALIGN
    assert lib.mixq_quant_extract(-1, 64, p16, p16, p16, None, None, 0, 0, None) == BADARG
    assert lib.mixq_quant_extract(0, 64, None, None, None, None, None, 0, 0, None) == OK      # empty input is fine
    # fused GEMM: K / N multiples of 16, alignment, nulls
    assert lib.mixq_gemm_mixed(p16, p16, p16, p16, None, None, p16, 8, 64, 40, 0, None) == SHAPE
    assert lib.mixq_gemm_mixed(p16, p16, p16, p16, None, None, p16, 8, 60, 64, 0, None) == SHAPE
    assert lib.mixq_gemm_mixed(odd, p16, p16, p16, None, None, p16, 8, 64, 64, 0, None) == ALIGN
    assert lib.mixq_gemm_mixed(None, p16, p16, p16, None, None, p16, 8, 64, 64, 0, None) == BADARG
    assert lib.mixq_gemm_mixed(p16, p16, p16, p16, None, None, p16, 8, 64, 64, 128, None) == BADARG  # O > 0 without operands
    assert lib.mixq_gemm_mixed(p16, p16, p16, p16, None, None, p16, 0, 64, 64, 0, None) == OK
    assert lib.mixq_gemm_mixed_scratch(p16, p16, p16, p16, None, None, p16, 8, 64, 40, 0, None, 0, None) == SHAPE
    assert lib.mixq_gemm_mixed_scratch(None, p16, p16, p16, None, None, p16, 8, 64, 64, 0, p16, 64, None) == BADARG
    assert lib.mixq_int8_fused_dequantize(p16, p16, p16, p16, None, p16, 8, 64, 24, None, None) == SHAPE
    assert lib.mixq_int8_fused_dequantize_silu_mul(p16, p16, p16, p16, None, None, p16, 8, 64, 64, None, None) == BADARG
    # 4-bit flavour and outlier helpers
    assert lib.mixq_int4quant(4, 60, p16, p16, p16, None) == SHAPE
    assert lib.mixq_int4_fused_dequantize(p16, p16, p16, p16, None, p16, 8, 64, 24, p16, None) == SHAPE
    assert lib.mixq_int4_fused_dequantize(p16, p16, p16, p16, None, p16, 80, 64, 32, None, None) == 5  # M > 64 needs the workspace (MIXQ_E_WORKSPACE)
    assert lib.mixq_int4_fused_dequantize(p16, p16, None, p16, None, p16, 8, 64, 32, None, None) == BADARG  # decode batch: no scales
    assert lib.mixq_int4_fused_dequantize_w8(p16, None, p16, p16, None, p16, 80, 64, 32, 0, p16, None) == BADARG
    assert lib.mixq_int4_fused_dequantize_w8(p16, p16, p16, p16, None, p16, 80, 64, 32, 2, p16, None) == BADARG  # epilogue 0 | 1
    assert lib.mixq_find_outliers(p16, 4, 60, ctypes.c_float(6.0), p16, p16, p16, 8, None) == SHAPE
    assert lib.mixq_find_outliers(p16, 4, 64, ctypes.c_float(6.0), None, p16, p16, 8, None) == BADARG
    assert lib.mixq_find_outliers_workspace_size(4096) == 512 and lib.mixq_int4_fused_workspace_size(65, 64, 16) == 65 * 32 + 64 * 32 and lib.mixq_int4_fused_workspace_size(32, 64, 16) == 0 and lib.mixq_int4_fused_workspace_size(32, 0, 16) == 32 * 32
    for code in (1, 2, 3, 4, 5):
        assert lib.mixq_error_string(code)


def test_scratch_of_any_shape_fits_the_plugin_workspace(lib):
    """mixq_enqueue carves the K-split scratch from the plugin workspace behind qA / sA / fpA: for every shape the scratch
    the GEMM asks for must fit what mixq_workspace_size reserved (host arithmetic only)."""
    h = lib.mixq_create(0, 0, 0)
    rng = np.random.default_rng(7)
    shapes = [(int(m), int(n) * 16, int(k) * 16) for m, n, k in
              zip(rng.integers(1, 20000, 400), rng.integers(1, 2000, 400), rng.integers(1, 2000, 400))]
    shapes += [(32, 4096, 11008), (24, 1024, 28672), (64, 4096, 11008), (128, 4096, 16384), (200, 2048, 8192),
               (1024, 4096, 11008), (2048, 4096, 11008), (1536, 11008, 4096), (4096, 1024, 28672), (256, 65536, 4096),
               (255, 65536, 4096), (200, 80000, 4096), (65536, 12288, 4096)]
    used = 0
    for M, N, K in shapes:
        need = lib.mixq_gemm_scratch_size(M, N, K)
        al = lambda x: (x + 127) & ~127
        carved = 128 + al(M * K) + al(2 * M) + al(2 * 128 * M)
        ws = lib.mixq_workspace_size(h, M, N, K)
        if M <= 4:
            continue     # decode path: no quantised operand, no scratch
        assert carved + need <= ws, (M, N, K, need, ws)
        used += need > 0
        # a workspace sized for maxM also serves every smaller call of the same layer
        for m in {max(5, M // 2), max(5, M - 1), min(M, 300), min(M, 129)}:
            carved_m = 128 + al(m * K) + al(2 * m) + al(2 * 128 * m)
            assert carved_m + lib.mixq_gemm_scratch_size(m, N, K) <= ws, (m, M, N, K)
    assert used > 20
    lib.mixq_destroy(h)


def test_asm_load_registers_are_not_moved_before_they_land():
    """The fpA_intB GEMM prefetches weights with asm-form loads the compiler does not track (mixq_device.h gload16_sbase); its
    loops are written so that hipcc has no reason to copy a destination register while the load is in flight.  Checked on the
    ISA this toolchain produces (tools/asm_load_check.py; ~12 s of hipcc): every kernel with such loads, no register move on
    their destination registers between the top of the main loop and the last load."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("asm_load_check", os.path.join(ROOT, "tools", "asm_load_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.check_isa(mod.compile_to_isa(mod.DEFAULT[0]))
    assert len(res) >= 8, [r[0] for r in res]          # narrow form x 4, register-weight wide forms x 4
    for name, n_loads, bad in res:
        assert n_loads > 0 and not bad, (name, bad[:4])


def test_headers_are_valid_c99_and_cxx11_and_the_c_and_cxx_hosts_link(tmp_path):
    """include/mixq.h is the product's boundary: it must compile, warning-free and pedantic, as C99 and as C++11 (the
    reference's plugin host is C++), and the plain C host of tests/c_abi/ must build and link against the library with
    nothing but gcc, the HIP runtime and libm (it RUNS in tests/test_gpu_c_host.py)."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for compiler, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        src = tmp_path / f"hdr.{ext}"
        src.write_text('#include "mixq.h"\nint main(void) { return mixq_plugin_type() == 0; }\n')
        r = subprocess.run([compiler, std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # the C++ mirror of the reference's two plugin classes (include/mixq_plugin.hpp) is header-only C++11
    src = tmp_path / "hpp.cpp"
    src.write_text('#include "mixq_plugin.hpp"\nint main() { mixq_plugin::MixQPluginCreator c; return c.getPluginName() == nullptr; }\n')
    r = subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    link = ["-L", os.path.join(ROOT, "mixq_tensorrt_llm_amd"), "-l:libmixq_mi355x.so", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
            "-Wl,-rpath,/opt/rocm/lib"]
    for compiler, std, name in (("gcc", "-std=c99", "host_example.c"), ("g++", "-std=c++17", "host_example.cpp")):  # (HIP's headers want C++17)
        exe = tmp_path / (name + ".out")
        r = subprocess.run([compiler, std, "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I", inc, "-I", "/opt/rocm/include",
                            os.path.join(ROOT, "tests", "c_abi", name), "-o", str(exe)] + link, capture_output=True, text=True)
        assert r.returncode == 0 and exe.exists(), r.stderr


def test_measurement_knobs_need_an_opt_in():
    """mixq_debug_set_* act only in a process that exported MIXQ_DEBUG_KNOBS=1 (include/mixq.h): a production process cannot have
    its kernel selection changed by a stray call; the Python binding makes a forgotten opt-in loud."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mixq_tensorrt_llm_amd import _lib\n"
            "lib = _lib.load()\n"
            "print(lib.mixq_debug_knobs_enabled())\n"
            "try:\n"
            "    lib.mixq_debug_set_gemm_variant(1); print('set')\n"
            "except _lib.MixQLibraryError: print('refused')\n"
            "lib.mixq_debug_reset(); print('reset ok')\n") % ROOT
    for env_val, want in ((None, ["0", "refused", "reset ok"]), ("1", ["1", "set", "reset ok"]), ("0", ["0", "refused", "reset ok"])):
        env = {k: v for k, v in os.environ.items() if k != "MIXQ_DEBUG_KNOBS"}
        if env_val is not None:
            env["MIXQ_DEBUG_KNOBS"] = env_val
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split("\n")[:3] == want, (env_val, r.stdout)


def test_selection_table_is_the_pinned_one(lib):
    """The kernel selection is a set of tables fitted on measurements; mixq_describe_plan (host only) says what mixq_enqueue
    would launch, and tests/golden/selection_table.json pins its answers on every BASELINE (N, K) x a ladder of row counts, with and
    without a registered weight image: a change of any rule must come with a regenerated table (tools/selection_table.py), so that it is
    visible as a diff -- and docs/SELECTION_TABLE.md, its readable form, stays true."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "selection_table.py"), "--check"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))   # (the 256-CU tables, wherever this runs)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    buf = ctypes.create_string_buffer(8)
    assert lib.mixq_describe_plan(32, 4096, 4096, 0, buf, 8) == 0 and len(buf.value) == 7      # truncated, terminated
    assert lib.mixq_describe_plan(0, 4096, 4096, 0, buf, 8) != 0 and lib.mixq_describe_plan(32, 4096, 4096, 0, None, 8) != 0

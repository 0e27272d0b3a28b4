"""GPU (-m gpu): the C ABI driven by a plain C99 host (tests/c_abi/host_example.c: no Python, no torch, no C++ in the caller) and by
a C++ host through the mirror of the reference's plugin classes (tests/c_abi/host_example.cpp, include/mixq_plugin.hpp) -- what a
maintainer of the reference's C++ runtime links against.  One prefill call and one decode call (M <= 4: the W8A16 path
on `qweight`), inputs and the oracle's expected output handed over as raw files."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_layer

pytestmark = pytest.mark.gpu


def _build(tmp_path, lang):
    exe = str(tmp_path / ("host_example_" + lang))
    compiler, std, src = ("gcc", "-std=c99", "host_example.c") if lang == "c" else ("g++", "-std=c++17", "host_example.cpp")
    cmd = [compiler, std, "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I", os.path.join(ROOT, "include"), "-I",
           "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", src), "-o", exe, "-L",
           os.path.join(ROOT, "mixq_tensorrt_llm_amd"), "-l:libmixq_mi355x.so", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + os.path.join(ROOT, "mixq_tensorrt_llm_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("lang", ["c", "cxx"])
@pytest.mark.parametrize("M,N,K", [(300, 512, 1024), (3, 384, 1024), (40, 256, 2048)])
def test_plain_c_and_cxx_hosts_run_the_plugin_lifecycle(tmp_path, oracle, lang, M, N, K):
    """`c`: the C ABI itself (include/mixq.h); `cxx`: the reference's two classes by their own method names
    (include/mixq_plugin.hpp: MixQPluginCreator::createPlugin / deserializePlugin, MixQPlugin::supportsFormatCombination /
    getOutputDimensions / configurePlugin / getWorkspaceSize / enqueue / clone / serialize / destroy)."""
    exe = _build(tmp_path, lang)
    A, W, act = make_layer(M, N, K, seed=3 * M + N, outlier_gain=1.0 if M <= 4 else 20.0)
    p = oracle.pack_linear_weights(W, act)
    d = tmp_path / "data"
    d.mkdir()
    A.astype(np.float16).tofile(d / "A.f16")
    np.ascontiguousarray(p["weight"]).astype(np.int8).tofile(d / "weight.i8")
    np.ascontiguousarray(p["weights_scaling_factor"]).astype(np.float16).tofile(d / "weights_scaling_factor.f16")
    np.ascontiguousarray(p["fp_weight"]).astype(np.float16).tofile(d / "fp_weight.f16")
    np.ascontiguousarray(p["fp_ind"]).astype(np.int32).tofile(d / "fp_ind.i32")
    np.ascontiguousarray(p["qweight"]).astype(np.uint8).tofile(d / "qweight.u8")
    if M <= 4:   # decode: the W8A16 path on the un-zeroed weights, max/127 scales reused (TsinghuaMixQPlugin.cpp:641-647)
        want = oracle.w8a16_gemv(A, oracle.eetq_symmetric_quantize(W.T.copy())[0], p["weights_scaling_factor"])
    else:
        want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    np.ascontiguousarray(want).astype(np.float16).tofile(d / "want.f16")
    r = subprocess.run([exe, str(d), str(M), str(N), str(K)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "bit-identical: 1" in r.stdout, r.stdout

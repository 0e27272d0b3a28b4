"""CPU: the product's quantize-time producer (pack.py) against the reference golden vectors and the oracle; and the
rule that the product never reaches into oracle/ or /root/reference."""
import os
import re

import numpy as np
import torch

from conftest import GOLDEN, ROOT, make_layer, assert_same_outlier_columns

from mixq_tensorrt_llm_amd import pack


def test_pack_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "pack_small.npz"))
    W = torch.from_numpy(g["W"])
    sW = pack.weight_scales(W)
    assert np.array_equal(sW.numpy().view(np.uint16), g["weights_scaling_factor"].view(np.uint16))
    Wz = W.clone()
    Wz[:, torch.from_numpy(g["fp_ind"]).long()] *= 0
    assert np.array_equal(pack.quantize_weight(Wz, sW).numpy(), g["weight_int8"])


def test_pack_matches_oracle_on_all_seven_tensors(oracle):
    A, W, act = make_layer(4, 128, 256, seed=2)
    p = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    o = oracle.pack_linear_weights(W, act)
    for key in ("weight", "fp_ind", "qweight"):
        assert np.array_equal(p[key], o[key]), key
    for key in ("weights_scaling_factor", "fp_weight", "scales"):
        assert np.array_equal(p[key].view(np.uint16), o[key].view(np.uint16)), key
    # declared carrier shapes of plugin.py:99-123
    N, K = W.shape
    assert p["weight"].view(np.float16).shape == (N, K // 2)
    assert p["fp_ind"].view(np.float16).shape == (256,)
    assert p["qweight"].view(np.float16).shape == (K, N // 2)


def test_real_llama_act_scales_select_same_columns(oracle):
    a = np.load(os.path.join(GOLDEN, "act_scales_llama.npz"))
    for i in range(3):
        mine = pack.select_outlier_columns(torch.from_numpy(a[f"scales_{i}"])).numpy()
        assert_same_outlier_columns(mine, a[f"fp_ind_{i}"])   # the reference's own order, ties included (its very torch.sort call; under the goldens' torch build)
        stable = pack.select_outlier_columns(torch.from_numpy(a[f"scales_{i}"]), stable=True).numpy()
        assert np.array_equal(stable, oracle.select_outliers(a[f"scales_{i}"]))   # the oracle orders tie groups by index
        assert set(mine.tolist()) == set(stable.tolist())


def test_product_never_touches_oracle_or_reference():
    pkg = os.path.join(ROOT, "mixq_tensorrt_llm_amd")
    bad = re.compile(r"(^|\s)(import|from)\s+oracle\b|/root/reference|libmixq_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                for ln, line in enumerate(open(os.path.join(dirpath, f), errors="ignore"), 1):
                    code = line.split("#")[0] if f.endswith(".py") else line
                    assert not bad.search(code) or "reference:" in line.lower(), f"{f}:{ln}: {line.strip()}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mixq_tensorrt_llm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.MixQLibraryError):
        _lib.load()


def test_cpu_tensors_are_rejected_not_emulated():
    import pytest
    from mixq_tensorrt_llm_amd import _lib, mixlib
    x = torch.zeros(8, 64, dtype=torch.float16)
    with pytest.raises(_lib.MixQLibraryError):
        mixlib.FindRowScale(x, torch.zeros(8, dtype=torch.float16), 8, 64, 8)

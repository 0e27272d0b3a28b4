#!/usr/bin/env python3
"""Randomised soak of the weight-image registry: several layers of random shapes registered / unregistered / re-registered in random
order while decode-batch calls (5..64 rows, random) run through mixq_enqueue; every output must equal the same call with the registry
ignored (knob 883).  usage: python tools/wimg_soak.py [--iters 400] [--seed 0]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(a.seed)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    layers = []
    for _ in range(10):
        N = int(rng.integers(1, 400)) * 16
        K = int(rng.integers(4, 130)) * 64
        t = bench.synth_layer(N, K, dev, gen)
        layers.append(dict(N=N, K=K, t=t, img=torch.empty(N * K, dtype=torch.int8, device=dev), reg=False))
    bad = served = 0
    for it in range(a.iters):
        L = layers[int(rng.integers(len(layers)))]
        act = rng.random()
        if act < 0.25:
            assert lib.mixq_weight_image_register(p(L["t"]["weight"]), L["N"], L["K"], p(L["img"]), st) == 0
            L["reg"] = True
        elif act < 0.35 and L["reg"]:
            assert lib.mixq_weight_image_unregister(p(L["t"]["weight"])) == 0
            L["reg"] = False
        M = int(rng.integers(5, 65))
        N, K, t = L["N"], L["K"], L["t"]
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
        in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
        ptrs = (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins])
        h = ctypes.c_void_p(lib.mixq_create(M, N, K))
        ws = torch.empty(max(lib.mixq_workspace_size(h, 64, N, K), 16), dtype=torch.uint8, device=dev)
        outs = []
        for knob in (883, 880):
            lib.mixq_debug_set_gemm_variant(knob)
            o = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            od = TensorDesc.make(o.shape)
            op = (ctypes.c_void_p * 1)(o.data_ptr())
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(od), ptrs, op, p(ws), st) == 0
            torch.cuda.synchronize()
            outs.append(o)
        served += int(L["reg"] and b"skinny" in lib.mixq_debug_last_gemm_kernel())
        if not torch.equal(outs[0], outs[1]):
            bad += 1
            print(f"MISMATCH it={it} M={M} N={N} K={K} registered={L['reg']}")
        lib.mixq_destroy(h)
    lib.mixq_debug_reset()
    for L in layers:
        lib.mixq_weight_image_unregister(p(L["t"]["weight"]))
    print(f"{a.iters} calls over {len(layers)} layers, {served} served from an image, {bad} mismatches")


if __name__ == "__main__":
    main()

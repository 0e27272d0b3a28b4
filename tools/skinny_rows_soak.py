#!/usr/bin/env python3
"""Soak of the decode-batch GEMM's 256-byte-run weight route (csrc/gemm_skinny_kernels.hip WFRAG == 3; packed-int4 twin WROWS): seeded
random shapes (5..64 rows, N % 16 == 0, K % 256 == 0 for int8 / any K % 32 == 0 for int4), every call once with the route and once
with the 64-byte fragment loads (knobs 885 / 873), outputs compared bit for bit.  usage: python tools/skinny_rows_soak.py [--n 300]"""
import argparse
import ctypes
import os
import random

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    rng = random.Random(11)
    st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    bad = served = 0
    for it in range(a.n):
        M, N, K = rng.randrange(5, 65), rng.randrange(4, 420) * 16, rng.randrange(1, 49) * 256
        t = bench.synth_layer(N, K, dev, gen)
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
        in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
        outs = []
        for knob in (885, 884):
            lib.mixq_debug_reset()
            lib.mixq_debug_set_gemm_variant(knob)
            o = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            out_desc = TensorDesc.make(o.shape)
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]),
                                    (ctypes.c_void_p * 1)(o.data_ptr()), p(ws), st0) == 0
            torch.cuda.synchronize(dev)
            outs.append(o)
            kern = lib.mixq_debug_last_gemm_kernel().decode()
            lib.mixq_destroy(h)
        served += "skinny" in kern
        if not torch.equal(outs[0], outs[1]) or torch.isnan(outs[1]).any():
            bad += 1
            print(f"MISMATCH int8 M={M} N={N} K={K} [{kern}]")
        # the packed-int4 twin on a shape of its own (K: elements; packed row = K / 2 bytes, any multiple of 16)
        M4, N4, K4 = rng.randrange(1, 65), rng.randrange(2, 300) * 16, rng.randrange(8, 400) * 32
        q4 = torch.randint(0, 256, (M4, K4 // 2), dtype=torch.uint8, device=dev, generator=gen)
        w4 = torch.randint(0, 256, (N4, K4 // 2), dtype=torch.uint8, device=dev, generator=gen)
        sa = (torch.rand(M4, device=dev, generator=gen) * 0.05 + 0.01).to(torch.float16)
        sw = (torch.rand(N4, device=dev, generator=gen) * 4e-4 + 4e-4).to(torch.float16)
        o4 = []
        for knob in (873, 874):
            lib.mixq_debug_reset()
            lib.mixq_debug_set_gemm_variant(knob)
            o = torch.full((M4, N4), float("nan"), dtype=torch.float16, device=dev)
            assert lib.mixq_int4_fused_dequantize(p(q4), p(w4), p(sa), p(sw), None, p(o), M4, N4, K4 // 2, None, st0) == 0
            torch.cuda.synchronize(dev)
            o4.append(o)
        if not torch.equal(o4[0], o4[1]) or torch.isnan(o4[1]).any():
            bad += 1
            print(f"MISMATCH int4 M={M4} N={N4} K={K4}")
    lib.mixq_debug_reset()
    print(f"{a.n} int8 shapes ({served} served by the skinny kernel) + {a.n} int4 shapes: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

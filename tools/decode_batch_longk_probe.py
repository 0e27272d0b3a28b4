#!/usr/bin/env python3
"""Decode batches on long K: the automatic selection ('auto') against the small-tile K split wherever its plan applies ('tiles': skinny
range off via knob 1 is not used -- the K split is forced by leaving the skinny form only its non-scratch range) and the fragment-major
skinny GEMM forced up to 64 rows with the tile K split off ('skinny'), with and without a registered weight image.  Operator
(mixq_enqueue) us per call, COLD weights, HIP graph of 100 calls; bit identity against the automatic path."""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402

CASES = [(1024, 28672), (1280, 8192), (3584, 18944), (3584, 8192), (4096, 11008), (4096, 16384), (2048, 8192), (8192, 8192),
         (512, 16384), (512, 8192), (1024, 8192), (2560, 12288), (5120, 13824), (8192, 16384)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--Ms", default="8,16,32,48,64")
    ap.add_argument("--cases", default="", help='"N K;N K;..." instead of the built-in long-K list')
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    cases = [tuple(int(x) for x in c.split()) for c in a.cases.split(";") if c.strip()] or CASES
    for N, K in cases:
        t = bench.synth_layer(N, K, dev, gen)
        copies = (320 << 20) // (N * K) + 2
        ws = [t["weight"]] + [t["weight"].clone() for _ in range(copies - 1)]
        imgs = [torch.empty(N * K, dtype=torch.int8, device=dev) for _ in ws]
        for M in [int(x) for x in a.Ms.split(",")]:
            A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
            o = torch.empty((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
            out_desc = TensorDesc.make(o.shape)
            out_ptrs = (ctypes.c_void_p * 1)(o.data_ptr())
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            sets = []
            for w in ws:
                v = [x.data_ptr() for x in ins]
                v[1] = w.data_ptr()
                sets.append((ctypes.c_void_p * 7)(*v))
            turn = [0]
            cells, ref = [], None
            for label, image, knobs in (("auto", False, ()), ("img+auto", True, ()), ("skinny", False, (60, 897)), ("img+skinny", True, (60, 897)),
                                        ("two-barrier tiles (variant 1)", False, (1,))):
                lib.mixq_debug_reset()
                st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for w, im in zip(ws, imgs):
                    if image:
                        assert lib.mixq_weight_image_register(p(w), N, K, p(im), st0) == 0
                    else:
                        lib.mixq_weight_image_unregister(p(w))
                for k in knobs:
                    lib.mixq_debug_set_gemm_variant(k)
                wsp = torch.zeros(max(lib.mixq_workspace_size(h, 64, N, K), 16), dtype=torch.uint8, device=dev)

                def cold(st):
                    assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), sets[turn[0] % copies], out_ptrs, p(wsp), st) == 0
                    turn[0] += 1
                turn[0] = 0
                o.zero_()
                cold(st0)
                torch.cuda.synchronize()
                kern = lib.mixq_debug_last_gemm_kernel().decode().split(" ")[0].replace("gemm_", "").replace("w8a8o16_", "")
                if ref is None:
                    ref = o.clone()
                ok = torch.equal(o, ref)
                if "skinny" in label and "skinny" not in kern:
                    continue   # the forced form does not apply to this shape
                tc = bench.graph_time_us(cold, dev)
                cells.append(f"{label} {tc:5.1f}{'' if ok else ' MISMATCH'}{' [' + kern + ']' if 'auto' in label or 'tiles' in label else ''}")
            lib.mixq_debug_reset()
            for w in ws:
                lib.mixq_weight_image_unregister(p(w))
            lib.mixq_destroy(h)
            print(f"M={M:3d} N={N:5d} K={K:6d} | " + " | ".join(cells), flush=True)
        del ws, imgs


if __name__ == "__main__":
    main()

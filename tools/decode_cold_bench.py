#!/usr/bin/env python3
"""Decode (M <= 4) and decode-batch calls through mixq_enqueue with WARM weights (the same layer every call: the Infinity Cache
serves them) and COLD weights (the calls cycle through enough distinct copies to exceed it: every call streams from HBM, as a
model's decode step does), HIP graph of 100 calls, per forced kernel form.  usage:
  python tools/decode_cold_bench.py [--shapes "12288 4096;4096 4096;4096 11008"] [--Ms 1,2,4] [--knobs "850,858;856;857;..."]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="12288 4096;4096 4096;4096 11008;11008 4096")
    ap.add_argument("--Ms", default="1,2,4")
    ap.add_argument("--knobs", default="0;857;856;856,852;856,853;856,854;856,855;856,859;856,8590")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    for shape in a.shapes.split(";"):
        N, K = (int(x) for x in shape.split())
        t = bench.synth_layer(N, K, dev, gen)
        for M in [int(x) for x in a.Ms.split(",")]:
            A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
            o = torch.empty((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
            out_desc = TensorDesc.make(o.shape)
            out_ptrs = (ctypes.c_void_p * 1)(o.data_ptr())
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, max(M, 64), N, K), 16), dtype=torch.uint8, device=dev)
            widx = 1 if M > 4 else 5
            copies = (320 << 20) // (N * K) + 2
            alts = [ins[widx]] + [ins[widx].clone() for _ in range(copies - 1)]
            ptr_sets = []
            for w in alts:
                v = [x.data_ptr() for x in ins]
                v[widx] = w.data_ptr()
                ptr_sets.append((ctypes.c_void_p * 7)(*v))
            cells = []
            for ks in a.knobs.split(";"):
                lib.mixq_debug_reset()
                for k in [int(x) for x in ks.split(",") if x and int(x) != 0]:
                    lib.mixq_debug_set_gemm_variant(k)
                turn = [0]

                def warm(st):
                    assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), ptr_sets[0], out_ptrs, ctypes.c_void_p(ws.data_ptr()), st) == 0

                def cold(st):
                    assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), ptr_sets[turn[0] % copies], out_ptrs,
                                            ctypes.c_void_p(ws.data_ptr()), st) == 0
                    turn[0] += 1
                tw, tc = bench.graph_time_us(warm, dev), bench.graph_time_us(cold, dev)
                cells.append(f"[{ks}] {tw:5.2f}/{tc:5.2f}")
            lib.mixq_debug_reset()
            lib.mixq_destroy(h)
            print(f"M={M} N={N} K={K} warm/cold us (cold floor at 6.3 TB/s: {N * K / 6.3e6:.2f}): " + "  ".join(cells), flush=True)
            del alts


if __name__ == "__main__":
    main()

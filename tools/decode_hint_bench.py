#!/usr/bin/env python3
"""Next-layer weight hint (mixq_enqueue_hint) on one decode step of Llama-2-7B's 96 MixQ linears (bench.decode_step_points): us per
step without the hint and with 25 / 50 / 100 % of the next layer's weight bytes hinted, batch 1 / 4 / 32, plus a bit-identity check
of hinted against plain calls.  usage: python tools/decode_hint_bench.py [--fracs 0.25,0.5,1.0]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib, parallel  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fracs", default="0.25,0.5,1.0")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    model = bench.Model(lib, TensorDesc, parallel, dev, gen, 64, 1, 0)   # (the prefill buffers stay tiny: 64-token chunks)
    # bit identity: hinted == plain, on the three shapes, M = 1 / 4 / 32
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (t, ins) in model.keep[:3]:
        N, K = t["weight"].shape[0], ins[0].shape[1]
        for M in (1, 4, 32):
            A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
            v = [A] + list(ins[1:])
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in v])
            ptrs = (ctypes.c_void_p * 7)(*[x.data_ptr() for x in v])
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            outs = []
            for hint in (False, True):
                o = torch.zeros((M, N), dtype=torch.float16, device=dev)
                od = TensorDesc.make(o.shape)
                op = (ctypes.c_void_p * 1)(o.data_ptr())
                nxt = model.keep[5][1][5 if M <= 4 else 1]
                if hint:
                    assert lib.mixq_enqueue_hint(h, in_desc, ctypes.byref(od), ptrs, op, ctypes.c_void_p(ws.data_ptr()), st,
                                                 ctypes.c_void_p(nxt.data_ptr()), nxt.numel() * 2) == 0
                else:
                    assert lib.mixq_enqueue(h, in_desc, ctypes.byref(od), ptrs, op, ctypes.c_void_p(ws.data_ptr()), st) == 0
                torch.cuda.synchronize()
                outs.append(o)
            print(f"identity M={M} N={N} K={K}: {torch.equal(outs[0], outs[1])}  [{lib.mixq_debug_last_gemm_kernel().decode()}]", flush=True)
            lib.mixq_destroy(h)
    for frac in [float(x) for x in a.fracs.split(",")]:
        r = bench.decode_step_points(lib, TensorDesc, model, dev, gen, hint_frac=frac)
        for bs in (1, 4, 32):
            d = r[f"bs{bs}"]
            print(f"hint {frac:4.2f}  bs{bs:<3d} plain {d['us_per_step']:8.1f} us ({d['hbm_frac']:.3f} of 8 TB/s)   hinted "
                  f"{d['with_next_weight_hint']['us_per_step']:8.1f} us ({d['with_next_weight_hint']['hbm_frac']:.3f})   "
                  f"{(d['with_next_weight_hint']['us_per_step'] / d['us_per_step'] - 1) * 100:+.1f} %", flush=True)


if __name__ == "__main__":
    main()

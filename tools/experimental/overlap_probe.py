#!/usr/bin/env python3
"""Does the quantiser of one token chunk overlap the GEMM of another?  One MixQ linear (N x K), chunks of M tokens,
mixq_enqueue on ONE stream vs alternating TWO streams (own workspace / activations / output each): wall time per call.
usage: python tools/overlap_probe.py [--M 65536 --N 12288 --K 4096 --calls 64]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=65536)
    ap.add_argument("--N", type=int, default=12288)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--calls", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(1)
    t = bench.synth_layer(a.N, a.K, dev, gen)
    acts = [bench.synth_activation(a.M, a.K, t["ind_i32"], dev, gen) for _ in range(4)]
    outs = [torch.empty((a.M, a.N), dtype=torch.float16, device=dev) for _ in range(2)]
    h = lib.mixq_create(a.M, a.N, a.K)
    ws_bytes = lib.mixq_workspace_size(h, a.M, a.N, a.K)
    wss = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    ins0 = [acts[0], t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
    in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins0])
    out_desc = TensorDesc.make(outs[0].shape)
    in_ptrs = [(ctypes.c_void_p * 7)(*([x.data_ptr()] + [y.data_ptr() for y in ins0[1:]])) for x in acts]
    out_ptrs = [(ctypes.c_void_p * 1)(o.data_ptr()) for o in outs]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def run(nstreams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.calls):
            s = i % nstreams
            rc = lib.mixq_enqueue(ctypes.c_void_p(h), in_desc, ctypes.byref(out_desc), in_ptrs[i % 4], out_ptrs[s],
                                  ctypes.c_void_p(wss[s].data_ptr()), ctypes.c_void_p(streams[s].cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.calls * 1e3

    for n in (1, 2, 1, 2):
        run(n)  # warm
        print(f"{n} stream(s): {run(n):.3f} ms per call of {a.M} x {a.N} x {a.K}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Fused RMSNorm -> extract -> quant producer (mixq_rmsnorm_extract_quant) and plain RMSNorm (mixq_rmsnorm) at prefill sizes:
us per call and algorithmic HBM rate (fused: 2MK in, 2MK + MK + 2M + 2*len*M out; plain: 2MK in, 2MK out), against the
quantiser alone (mixq_quant_extract, zeroing flavour).  usage: python tools/norm_bench.py [--Ms 2048,16384,65536] [--K 4096]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--Ms", default="64,2048,16384,65536")
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    K, O = a.K, 128
    g = torch.Generator(device=dev).manual_seed(0)
    gamma = (torch.rand(K, device=dev, generator=g) + 0.5).to(torch.float16)
    ind = torch.randperm(K, device=dev, generator=g)[:O].to(torch.int32)
    for M in [int(x) for x in a.Ms.split(",")]:
        x = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
        out = torch.empty_like(x)
        q = torch.empty((M, K), dtype=torch.int8, device=dev)
        s = torch.empty(M, dtype=torch.float16, device=dev)
        f = torch.empty((M, O), dtype=torch.float16, device=dev)
        xq = x.clone()

        def fused():
            assert lib.mixq_rmsnorm_extract_quant(M, K, p(x), p(gamma), p(out), 1e-6, p(ind), O, p(f), p(q), p(s), st) == 0

        def plain():
            assert lib.mixq_rmsnorm(M, K, p(x), p(gamma), p(out), 1e-6, st) == 0

        def quant():
            assert lib.mixq_quant_extract(M, K, p(xq), p(q), p(s), p(f), p(ind), O, 1, st) == 0
        cells = []
        for name, fn, nbytes in (("fused norm+extract+quant", fused, 2 * M * K + 2 * M * K + M * K + 2 * M + 2 * O * M),
                                 ("plain rmsnorm", plain, 4 * M * K), ("quantiser (zeroing)", quant, 2 * M * K + M * K + 2 * M + 2 * O * M)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            cells.append(f"{name} {us:8.1f} us {nbytes / us / 1e6:6.2f} TB/s")
        print(f"M={M:6d} K={K}: " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()

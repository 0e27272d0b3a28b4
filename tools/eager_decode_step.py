#!/usr/bin/env python3
"""Torch-side boundary cost where it matters: an EAGER (no HIP graph) decode step of Llama-2-7B's 96
MixLinear_GEMM linears -- the reference's own published metric is such a loop (MixQ/src/benchflops.py:311-315, eager
`MixLinear_GEMM.forward` at bs 32..512) -- through

  (i)   the reference's four mixlib calls per linear (ExtractOutliersAndSetToZeros, FindRowScale, the outlier product,
        int8FusedDequantize) via the Python wrappers of mixq_tensorrt_llm_amd/mixlib.py,
  (ii)  MixLinear_GEMM.forward(unfused=True) = ONE library call / two launches per linear (mixq_mixlinear_forward via ctypes),
  (iii) a LOWER BOUND for any compiled (pybind11 / torch-extension) binding of (ii): the same C entry called straight through ctypes
        on PRE-ALLOCATED outputs with prepared pointers -- no tensor allocation, no argument conversion, no Python-level checks;
        a pybind11 module has to do all of that (at::empty x 3, tensor -> pointer x 10) on top of the same C call,
  (gpu) the same 96 calls replayed as one HIP graph: the GPU's own time for the step (what an infinitely fast host would get).

usage: python tools/eager_decode_step.py [--bs 32,8] [--steps 30]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib, mixlib, mixlinear  # noqa: E402

SHAPES = [(12288, 4096), (11008, 4096), (4096, 11008)]   # attention.qkv, mlp.gate, mlp.proj (Llama-2-7B), x 32 layers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", default="32,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--layers", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    cache = mixlinear.MixLibCache(inputdim=1024, device=dev)
    layers = []
    for _ in range(a.layers):
        for N, K in SHAPES:
            L = mixlinear.MixLinear_GEMM(K, N, False, dev, bit=8, cache=cache)
            L.q_weight.copy_(torch.randint(-127, 128, (N, K), device=dev, generator=g, dtype=torch.int32).to(torch.int8))
            L.scale_col.copy_((torch.rand((1, N), device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16))
            L.ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
            L.weight_cache = (torch.randn((N, 128), device=dev, generator=g) * 0.02).to(torch.float16)
            L.add_outliers = False
            layers.append(L)
    for bs in [int(x) for x in a.bs.split(",")]:
        xs = {K: torch.randn((bs, K), device=dev, generator=g).to(torch.float16) for K in (4096, 11008)}

        def step_four():
            for L in layers:
                x = xs[L.in_features]
                o = mixlib.ExtractOutliersAndSetToZeros(L.ind, x)
                q = mixlib.FindRowScale(x, cache.x_scale, bs, L.in_features, 8)
                y = mixlinear.outlier_product(o, L.weight_cache)
                mixlib.int8FusedDequantize(q, L.q_weight, cache.x_scale, L.scale_col, y, bs, L.out_features, L.in_features)

        def step_one():
            for L in layers:
                L.forward(xs[L.in_features], cache, True)

        # (iii) prepared direct calls
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        prepared = []
        keep = []
        for L in layers:
            N, K = L.out_features, L.in_features
            lay = int(lib.mixq_qa_layout(bs, N, K))
            q = torch.empty(int(lib.mixq_qa_bytes(bs, K, lay)), dtype=torch.int8, device=dev)
            outl = torch.empty((bs, 128), dtype=torch.float16, device=dev)
            D = torch.empty((bs, N), dtype=torch.float16, device=dev)
            scr = None if lay else mixlib.gemm_scratch(xs[K], bs, N, K)
            keep.append((q, outl, D, scr))
            prepared.append((bs, N, K, 128, xs[K].data_ptr(), L.ind.data_ptr(), L.q_weight.data_ptr(), L.scale_col.data_ptr(),
                             L.weight_cache.data_ptr(), cache.x_scale.data_ptr(), q.data_ptr(), outl.data_ptr(), D.data_ptr(), lay,
                             scr.data_ptr() if scr is not None else None, scr.numel() if scr is not None else 0))
        fwd = lib.mixq_mixlinear_forward

        def step_direct(stp=st):
            for args in prepared:
                fwd(*args, stp)

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / a.steps * 1e6

        t4, t1, td = timeit(step_four), timeit(step_one), timeit(step_direct)
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                step_direct(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        tg = timeit(gr.replay)
        n = len(layers)
        print(f"bs {bs:3d}, {n} linears, eager step (us; per linear):  (i) four wrappers {t4:8.0f} ({t4 / n:5.1f})   (ii) one call "
              f"{t1:8.0f} ({t1 / n:5.1f})   (iii) bound for a compiled binding {td:8.0f} ({td / n:5.1f})   (gpu) graph replay "
              f"{tg:8.0f} ({tg / n:5.1f})   (ii)/(iii) = {t1 / td:.2f}x   (i)/(ii) = {t4 / t1:.2f}x", flush=True)


if __name__ == "__main__":
    main()

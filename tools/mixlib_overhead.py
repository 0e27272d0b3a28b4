#!/usr/bin/env python3
"""Host-side cost of the mixlib wrappers: the four mixlib calls of MixLinear_GEMM.forward(unfused=True)
(linear.py:163-286: ExtractOutliersAndSetToZeros, FindRowScale, the outlier product, int8FusedDequantize) on a tiny shape,
eager, no graph -- the GPU work is ~2 us per launch, so the wall time per call is what the HOST spends per call -- (a) through
the Python wrappers of mixq_tensorrt_llm_amd/mixlib.py (decorators, asserts, torch.empty per call), (b) as direct ctypes
calls into the C ABI on preallocated outputs, (c) the one-call entry mixq_mixlinear_forward.  The reference binds these ops
with pybind11 (quantkernel/mix_cuda/pybind_mix.cpp:256-335)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib, mixlib, mixlinear  # noqa: E402


def bench(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()        # host time to ISSUE n calls (the queue never fills: the kernels are shorter than the host)
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    M, N, K, O = 8, 256, 256, 32
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
    ind = torch.randperm(K, device=dev, generator=g)[:O].to(torch.int32)
    qw = torch.randint(-127, 128, (N, K), device=dev, generator=g, dtype=torch.int32).to(torch.int8)
    sc = (torch.rand((N, 1), device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
    wc = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    xs = torch.empty((M, 1), dtype=torch.float16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    res = {}
    # (a) wrappers, op by op
    res["wrapper ExtractOutliersAndSetToZeros"] = bench(lambda: mixlib.ExtractOutliersAndSetToZeros(ind, x))
    res["wrapper FindRowScale"] = bench(lambda: mixlib.FindRowScale(x, xs, M, K, 8))
    outl = mixlib.ExtractOutliersAndSetToZeros(ind, x)
    q = mixlib.FindRowScale(x, xs, M, K, 8)
    res["wrapper outlier_product"] = bench(lambda: mixlinear.outlier_product(outl, wc))
    y = mixlinear.outlier_product(outl, wc)
    res["wrapper int8FusedDequantize"] = bench(lambda: mixlib.int8FusedDequantize(q, qw, xs, sc, y, M, N, K))

    def seq_wrapped():
        o = mixlib.ExtractOutliersAndSetToZeros(ind, x)
        qq = mixlib.FindRowScale(x, xs, M, K, 8)
        yy = mixlinear.outlier_product(o, wc)
        return mixlib.int8FusedDequantize(qq, qw, xs, sc, yy, M, N, K)
    res["wrapper: the 4-call sequence"] = bench(seq_wrapped)
    # (b) direct ctypes calls, preallocated outputs, prepared pointers
    D = torch.empty((M, N), dtype=torch.float16, device=dev)
    px, pind, poutl, pq, pxs, pwc, py, pqw, psc, pD = (p(t) for t in (x, ind, outl, q, xs, wc, y, qw, sc, D))

    def seq_direct():
        lib.mixq_extract_outliers_set_zero(M, K, px, poutl, pind, O, st)
        lib.mixq_int8quant(M, K, px, pq, pxs, st)
        lib.mixq_gemm_fp16(poutl, pwc, py, M, N, O, st)
        lib.mixq_int8_fused_dequantize(pq, pqw, pxs, psc, py, pD, M, N, K, None, st)
    res["direct ctypes: the 4-call sequence"] = bench(seq_direct)
    res["direct ctypes: one launch (mixq_int8quant)"] = bench(lambda: lib.mixq_int8quant(M, K, px, pq, pxs, st))
    res["ctypes call without a launch (mixq_gemm_scratch_size)"] = bench(lambda: lib.mixq_gemm_scratch_size(M, N, K))
    res["torch.empty((M, N), fp16, cuda)"] = bench(lambda: torch.empty((M, N), dtype=torch.float16, device=dev))
    res["torch.cuda.current_stream().cuda_stream"] = bench(lambda: torch.cuda.current_stream(dev).cuda_stream)
    if hasattr(lib, "mixq_mixlinear_forward"):
        res["one-call entry mixq_mixlinear_forward (2 launches), direct"] = bench(
            lambda: lib.mixq_mixlinear_forward(M, N, K, O, px, pind, pqw, psc, pwc, pxs, pq, poutl, pD, 0, None, 0, st))
        if hasattr(mixlib, "mixlinear_forward"):
            res["one-call entry through mixlib.mixlinear_forward"] = bench(
                lambda: mixlib.mixlinear_forward(x, ind, qw, sc, wc, xs))
    for k, v in res.items():
        print(f"{k:62s} {v:7.2f} us / call (host)")
    per_call = (res["wrapper: the 4-call sequence"] - res["direct ctypes: the 4-call sequence"]) / 4
    print(f"wrapper overhead: {per_call:.2f} us per mixlib call over a direct C-ABI call")


if __name__ == "__main__":
    main()

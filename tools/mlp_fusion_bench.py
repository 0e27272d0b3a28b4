#!/usr/bin/env python3
"""gate * up fusion: int8FusedDequantizeSilu + `*= up` (the reference's two steps) vs int8FusedDequantizeSiluMul."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mixq_tensorrt_llm_amd import mixlib

M, N, K = 16384, 11008, 4096
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randint(-20, 21, (M, K), dtype=torch.int8, device=dev, generator=g)
b = torch.randn((N, K), device=dev, generator=g).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
sa = (torch.rand((M, 1), device=dev, generator=g) * 1e-2 + 1e-3).half()
sb = (torch.rand((1, N), device=dev, generator=g) * 1e-3 + 1e-4).half()
y = torch.randn((M, N), device=dev, generator=g).half()
up = torch.randn((M, N), device=dev, generator=g).half()


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def two_step():
    o = mixlib.int8FusedDequantizeSilu(a, b, sa, sb, y, M, N, K)
    o *= up
    return o


t_silu = timed(lambda: mixlib.int8FusedDequantizeSilu(a, b, sa, sb, y, M, N, K))
t_two = timed(two_step)
t_fused = timed(lambda: mixlib.int8FusedDequantizeSiluMul(a, b, sa, sb, y, up, M, N, K))
print(f"M={M} N={N} K={K}: silu GEMM {t_silu:.0f} us | + torch `*= up` {t_two:.0f} us | fused silu*up GEMM {t_fused:.0f} us")

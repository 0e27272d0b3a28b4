#!/usr/bin/env python3
"""Yardstick (not part of the product): the vendor int8 GEMM reachable from PyTorch (torch._int_mm -> hipBLASLt,
s8 x s8 -> s32, no dequant epilogue, no outlier side GEMM, int32 output) on the bench's GEMM shapes."""
import time
import torch

dev = "cuda:0"
for (M, N, K) in [(8192, 12288, 4096), (8192, 11008, 4096), (8192, 4096, 11008), (65536, 12288, 4096)]:
    a = torch.randint(-20, 21, (M, K), dtype=torch.int8, device=dev)
    b = torch.randn((N, K), device=dev).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
    bt = b.t()  # [K, N] column-major view: the same operand layout the MixQ kernel reads
    try:
        for _ in range(20):
            torch._int_mm(a, bt)
        torch.cuda.synchronize()
        iters = 300 if M <= 8192 else 60
        t0 = time.perf_counter()
        for _ in range(iters):
            torch._int_mm(a, bt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f"torch._int_mm M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.0f} TOPS (int32 out, no epilogue)")
    except Exception as e:  # noqa: BLE001
        print(f"torch._int_mm M={M} N={N} K={K}: not available ({type(e).__name__}: {e})")

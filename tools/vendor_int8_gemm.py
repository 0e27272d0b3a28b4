#!/usr/bin/env python3
"""Yardstick (not part of the product): the vendor int8 GEMM reachable from PyTorch (torch._int_mm -> hipBLASLt,
s8 x s8 -> s32, no dequant epilogue, no outlier side GEMM, int32 output) on the bench's GEMM shapes.

--mid: decode-batch shapes (M = 96..1024 on the Llama-2-7B linears) in steady state, warm and with weights cycled through
> 320 MiB of copies (cold), next to the fused MixQ GEMM (which also does the 128-column fp16 side product and the fp16 epilogue).
usage: python tools/vendor_int8_gemm.py [--mid] [--secs 0.25]"""
import argparse
import ctypes
import os
import sys
import time

import torch

dev = "cuda:0"


def steady(fn, secs):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
    est = max(e0.elapsed_time(e1) / 2 * 1e-3, 1e-6)
    for _ in range(max(3, int(0.1 / est))):
        fn()
    n = max(5, int(secs / est))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def mid(secs, all_shapes=False):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
    O = 128
    print("# us per launch, steady state; vendor = torch._int_mm (int32 out, no epilogue, no outlier product); mixq = the fused GEMM")
    shapes = [(12288, 4096), (11008, 4096), (4096, 11008)]
    if all_shapes:   # every (N, K) of BASELINE.json's configs: + Qwen2-7B and one GPU's shard of Llama-2-70B at TP = 8
        shapes += [(4608, 3584), (18944, 3584), (3584, 18944), (1280, 8192), (3584, 8192), (1024, 28672)]
    for (N, K) in shapes:
        W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        Ws = [W] + [W.clone() for _ in range((320 << 20) // (N * K) + 1)]
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
        fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
        for M in ((64, 96, 128, 192, 256, 384, 512, 768, 1024) if all_shapes else (96, 128, 256, 512, 1024)):
            qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
            sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
            fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
            turn = [0]

            def ours(cold):
                w = Ws[turn[0] % len(Ws)] if cold else W
                turn[0] += 1
                assert lib.mixq_gemm_mixed_scratch(p(qA), p(w), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                                   p(scr) if nscr else None, nscr, st) == 0

            def vend(cold):
                w = Ws[turn[0] % len(Ws)] if cold else W
                turn[0] += 1
                torch._int_mm(qA, w.t())

            row = []
            for cold in (False, True):
                try:
                    tv = steady(lambda: vend(cold), secs)
                except Exception as e:  # noqa: BLE001  (torch._int_mm has shape constraints)
                    tv = float("nan")
                    print(f"  torch._int_mm M={M}: {type(e).__name__}: {e}")
                to = steady(lambda: ours(cold), secs)
                row.append(f"{'cold' if cold else 'warm'}: vendor {tv:6.1f}  mixq {to:6.1f}  ({(to / tv - 1) * 100:+5.1f} %)")
            print(f"M={M:5d} N={N:6d} K={K:6d}  " + "   ".join(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mid", action="store_true")
    ap.add_argument("--secs", type=float, default=0.25)
    ap.add_argument("--all", action="store_true", help="--mid on all nine BASELINE (N, K), 64..1024 rows")
    a = ap.parse_args()
    if a.mid:
        return mid(a.secs, a.all)
    for (M, N, K) in [(8192, 12288, 4096), (8192, 11008, 4096), (8192, 4096, 11008), (65536, 12288, 4096)]:
        a_ = torch.randint(-20, 21, (M, K), dtype=torch.int8, device=dev)
        b = torch.randn((N, K), device=dev).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
        bt = b.t()  # [K, N] column-major view: the same operand layout the MixQ kernel reads
        try:
            for _ in range(20):
                torch._int_mm(a_, bt)
            torch.cuda.synchronize()
            iters = 300 if M <= 8192 else 60
            t0 = time.perf_counter()
            for _ in range(iters):
                torch._int_mm(a_, bt)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            print(f"torch._int_mm M={M} N={N} K={K}: {dt*1e6:.1f} us  {2*M*N*K/dt/1e12:.0f} TOPS (int32 out, no epilogue)")
        except Exception as e:  # noqa: BLE001
            print(f"torch._int_mm M={M} N={N} K={K}: not available ({type(e).__name__}: {e})")


if __name__ == "__main__":
    main()

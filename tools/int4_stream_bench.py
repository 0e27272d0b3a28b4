#!/usr/bin/env python3
"""Packed-int4 weight stream (csrc/int4_gemm_kernels.hip) against the unpack-to-int8 route it replaces (knob 871) and against the int8
decode-batch operator's GEMM on the same shape: us per call, HIP graph of 100 calls, warm (one weight) and cold (weights cycled through
> 320 MiB).  usage: python tools/int4_stream_bench.py [--shapes "4096 4096;12288 4096;4096 11008"] [--Ms 1,8,16,32,48,64]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096 4096;12288 4096;4096 11008")
    ap.add_argument("--Ms", default="1,8,16,32,48,64")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    print("# us per call (HIP graph of 100): s4 stream (256-byte runs) | s4 stream with 64-byte fragment loads (knob 873) | unpack route (2 unpack launches + int8 GEMM) | int8 GEMM alone (row-major qA); warm / cold")
    for shape in a.shapes.split(";"):
        N, K = (int(x) for x in shape.split())
        copies = (320 << 20) // (N * K // 2) + 2
        w4 = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev, generator=gen) for _ in range(copies)]
        c8 = (320 << 20) // (N * K) + 2
        w8 = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=gen) for _ in range(c8)]
        sw = (torch.rand(N, device=dev, generator=gen) * 4e-4 + 4e-4).to(torch.float16)
        for M in [int(x) for x in a.Ms.split(",")]:
            q4 = torch.randint(0, 256, (M, K // 2), dtype=torch.uint8, device=dev, generator=gen)
            q8 = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=gen)
            sa = (torch.rand(M, device=dev, generator=gen) * 0.05 + 0.01).to(torch.float16)
            o = torch.empty((M, N), dtype=torch.float16, device=dev)
            ws = torch.empty(max(16, lib.mixq_int4_fused_workspace_size(M, N, K // 2)), dtype=torch.uint8, device=dev)
            turn = [0]

            def s4(cold):
                def f(st):
                    w = w4[turn[0] % copies] if cold else w4[0]
                    turn[0] += 1
                    assert lib.mixq_int4_fused_dequantize(p(q4), p(w), p(sa), p(sw), None, p(o), M, N, K // 2, p(ws), st) == 0
                return f

            def i8(cold):
                def f(st):
                    w = w8[turn[0] % c8] if cold else w8[0]
                    turn[0] += 1
                    assert lib.mixq_int8_fused_dequantize(p(q8), p(w), p(sa), p(sw), None, p(o), M, N, K, None, st) == 0
                return f

            cells = []
            for knobs, mk in (((870, 872), s4), ((870, 873), s4), ((871, 872), s4), ((870, 872), i8)):   # 873: the 64-byte fragment loads of the first build
                for knob in knobs:
                    lib.mixq_debug_set_gemm_variant(knob)
                cells.append(f"{bench.graph_time_us(mk(False), dev):6.2f} / {bench.graph_time_us(mk(True), dev):6.2f}")
            lib.mixq_debug_set_gemm_variant(870)
            lib.mixq_debug_set_gemm_variant(872)
            bw = N * K / 2 / (float(cells[0].split('/')[1]) * 1e-6) / 1e12
            print(f"M={M:3d} N={N:6d} K={K:6d}  s4 {cells[0]}  s4 (64-B loads) {cells[1]}  unpack {cells[2]}  int8 {cells[3]}   (s4 cold: {bw:.2f} TB/s of packed weight)")


if __name__ == "__main__":
    main()

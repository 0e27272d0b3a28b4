#!/usr/bin/env python3
"""fpA_intB GEMM (mixq_w8a16_gemm_forward_ws) timing over M on one shape: us per call, weight GB/s, fp16 TFLOP/s.
usage: python tools/w8a16_bench.py --N 12288 --K 4096 [--Ms 5,16,32,64,128,256,512] [--no-scratch] [--iters 200]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=12288)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--Ms", default="1,4,5,16,32,64,128,256,512,1024")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--no-scratch", action="store_true")
    ap.add_argument("--variant", default="", help="comma list of mixq_debug_set_gemm_variant knobs (81 narrow | 82 / 84 wide; 86..89 K split)")
    ap.add_argument("--sweep", default="", help="';'-separated knob sets timed side by side per M, e.g. '81;82,86;82,87;84,86' (compact output)")
    ap.add_argument("--vendor", action="store_true", help="also time torch fp16 matmul on pre-dequantised weights")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    N, K = a.N, a.K
    for v in [int(x) for x in a.variant.split(",") if x]:
        lib.mixq_debug_set_gemm_variant(v)
    g = torch.Generator(device=dev).manual_seed(0)
    Wq = torch.randint(0, 256, (K, N), dtype=torch.uint8, device=dev, generator=g)   # any bytes are valid weights
    sc = (torch.rand(N, device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    Wf = None
    if a.vendor:
        Wf = torch.randn((N, K), device=dev, generator=g).to(torch.float16)
    if a.sweep:
        sets = [[int(x) for x in part.split(",") if x] for part in a.sweep.split(";")]
        for M in [int(x) for x in a.Ms.split(",")]:
            A = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            cells = []
            for knobs in [sets[0]] + sets:   # (the first set runs once untimed: clocks / caches / function attributes warm)
                lib.mixq_debug_set_gemm_variant(80)   # (also: ablations off, two-pass form automatic)
                lib.mixq_debug_set_gemm_variant(85)
                for v in knobs:
                    lib.mixq_debug_set_gemm_variant(v)
                nws = int(lib.mixq_w8a16_gemm_workspace_size(M, N, K))
                ws = torch.zeros(max(nws, 16384), dtype=torch.uint8, device=dev)

                def run():
                    assert lib.mixq_w8a16_gemm_forward_ws(A.data_ptr(), Wq.data_ptr(), sc.data_ptr(), out.data_ptr(), M, N,
                                                          K, ws.data_ptr() if nws else None, nws, st) == 0
                for _ in range(20):
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    run()
                torch.cuda.synchronize()
                cells.append(f"{'+'.join(map(str, knobs))}={(time.perf_counter() - t0) / a.iters * 1e6:.1f}")
            cells = cells[1:]
            print(f"sweep N={N} K={K} M={M}: " + " ".join(cells), flush=True)
        return
    for M in [int(x) for x in a.Ms.split(",")]:
        A = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        nws = 0 if a.no_scratch else int(lib.mixq_w8a16_gemm_workspace_size(M, N, K))
        ws = torch.zeros(max(nws, 16384), dtype=torch.uint8, device=dev)

        def run():
            rc = lib.mixq_w8a16_gemm_forward_ws(A.data_ptr(), Wq.data_ptr(), sc.data_ptr(), out.data_ptr(), M, N, K,
                                                ws.data_ptr() if nws else None, nws, st)
            assert rc == 0, rc
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        line = (f"w8a16 M={M:5d} N={N} K={K}: {dt*1e6:8.1f} us  weights {N*K/dt/1e9:7.0f} GB/s  "
                f"{2.0*M*N*K/dt/1e12:7.1f} TFLOP/s  scratch {nws >> 10} KiB")
        if Wf is not None:
            for _ in range(5):
                torch.matmul(A, Wf.t())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                torch.matmul(A, Wf.t())
            torch.cuda.synchronize()
            dv = (time.perf_counter() - t0) / a.iters
            line += f"  | torch fp16 matmul {dv*1e6:8.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()

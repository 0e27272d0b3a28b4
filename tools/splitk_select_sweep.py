#!/usr/bin/env python3
"""Selection data for the K split over workgroups (gemm_pp_kernels.hip, SPLITK): one process, many shapes; per shape the
fused GEMM time of the plain launch (variant 70), the forced 2-way / 4-way split (72 / 74) and torch._int_mm (the vendor's
plain s8 GEMM, for scale), plus a bit-identity check of every split result against the plain launch.
--secs S times every cell in STEADY STATE (0.3 s of warm-up launches, then >= S seconds timed: the power-capped clock the
bench runs at; 100-iteration cells read the boost clock of a chip that was idle a moment ago and favour the split forms);
--hybrid = the grid of whole rounds + a partial one (N = 4096: tiles = 16 x M / 256) behind the "solo + split tail" rule.
usage: python tools/splitk_select_sweep.py [--iters 100 | --secs 0.4] [--vendor] [--hybrid]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402

NK = [(4096, 4096), (12288, 4096), (11008, 4096), (22016, 4096), (4096, 11008), (5120, 5120), (15360, 5120),
      (13824, 5120), (5120, 13824), (8192, 8192), (10240, 8192), (28672, 8192), (8192, 28672), (6144, 4096), (4096, 16384)]
MS = [192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--vendor", action="store_true")
    ap.add_argument("--secs", type=float, default=0.0)
    ap.add_argument("--hybrid", action="store_true")
    ap.add_argument("--shapes", default="", help='explicit list "M,N,K;M,N,K;..." instead of the built-in grid')
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    O = 128
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    scr = torch.zeros(72 << 20, dtype=torch.uint8, device=dev)
    print("M N K tiles nk | plain_us s2_us s4_us s8_us vendor_us | best")
    grid = [(N, K, MS) for N, K in NK]
    if a.hybrid:   # rounds r in {1, 2, 5}, tail f in {1/8 .. 1/2} of the CUs: M = 256 * 16 * (r + f)
        grid = [(4096, K, [int(4096 * (r + f)) for r in (1, 2, 5) for f in (0.125, 0.25, 0.375, 0.5)])
                for K in (3584, 4096, 8192, 11008, 18944)]
    if a.shapes:
        grid = [(int(t.split(",")[1]), int(t.split(",")[2]), [int(t.split(",")[0])]) for t in a.shapes.split(";") if t]
    for N, K, ms in grid:
        W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
        fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
        for M in ms:
            qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
            sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
            fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            res, ref = {}, None
            for v in (70, 72, 74, 78):
                lib.mixq_debug_set_gemm_variant(v)
                nscr = lib.mixq_gemm_scratch_size(M, N, K)
                if v != 70 and nscr == 0:
                    res[v] = float("nan")
                    continue
                assert nscr <= scr.numel()
                fn = lambda: lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                                         p(scr) if nscr else None, nscr, st)
                out.zero_()
                for _ in range(3):
                    assert fn() == 0
                torch.cuda.synchronize()
                if v == 70:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    print(f"MISMATCH M={M} N={N} K={K} variant={v}")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = a.iters
                if a.secs > 0:
                    e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
                    est = e0.elapsed_time(e1) / 2 * 1e-3
                    for _ in range(max(3, int(0.3 / est))):
                        fn()
                    iters = max(5, int(a.secs / est))
                e0.record()
                for _ in range(iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[v] = e0.elapsed_time(e1) / iters * 1e3
            ven = float("nan")
            if a.vendor and M > 16:
                Wt = W.t()
                for _ in range(3):
                    torch._int_mm(qA, Wt)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    torch._int_mm(qA, Wt)
                e1.record()
                torch.cuda.synchronize()
                ven = e0.elapsed_time(e1) / a.iters * 1e3
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            best = min((t, v) for v, t in res.items() if t == t)[1]
            print(f"{M} {N} {K} {tiles} {(K + 127) // 128} | {res[70]:.1f} {res[72]:.1f} {res[74]:.1f} {res[78]:.1f} {ven:.1f} | {best}", flush=True)
    lib.mixq_debug_set_gemm_variant(0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel-level A/B harness: time mixq_gemm_mixed (and the quant pre-pass) on one shape with HIP events.
usage: python tools/gemm_bench.py [--M 8192 --N 12288 --K 4096 --O 128 --variant 0|1|2 --iters 20 --what gemm|quant|both]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=8192)
    ap.add_argument("--N", type=int, default=12288)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--O", type=int, default=128)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--variant2", type=int, default=None, help="a second knob set after --variant (e.g. 1 then 64)")
    ap.add_argument("--variant3", type=int, default=None, help="a third knob")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--what", default="both")
    ap.add_argument("--scratch-bound", action="store_true", help="hand the GEMM a scratch of mixq_gemm_scratch_bound() bytes")
    ap.add_argument("--check", action="store_true", help="compare the output with the plain launch bit for bit (20 rounds)")
    ap.add_argument("--zero", action="store_true", help="zero-filled operands (DVFS ceiling probe)")
    ap.add_argument("--qa-mode", default="", help="power probe (wrong results): replace qA after the quantiser -- 'offset8' = "
                    "q + 8 clamped (all small positives), 'abs' = |q|, 'zero_outl' = outlier columns zeroed, 'full' = uniform int8")
    ap.add_argument("--stamps", action="store_true", help="print the per-tile timeline from in-kernel s_memtime stamps")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.mixq_debug_set_gemm_variant(a.variant)
    if a.variant2 is not None:
        lib.mixq_debug_set_gemm_variant(a.variant2)
    if a.variant3 is not None:
        lib.mixq_debug_set_gemm_variant(a.variant3)
    g = torch.Generator(device=dev).manual_seed(0)
    M, N, K, O = a.M, a.N, a.K, a.O
    A = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
    W = torch.randn((N, K), device=dev, generator=g).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
    ind = torch.randperm(K, device=dev, generator=g)[:O].to(torch.int32)
    A[:, ind.long()] *= 20
    if a.zero:
        A.zero_(), W.zero_()
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    qA = torch.empty((M + 16, K), dtype=torch.int8, device=dev)[:M]   # (slack: the stride probe of the skinny kernel reads past M x K)
    sA = torch.empty(M, dtype=torch.float16, device=dev)
    fpA = torch.empty((M, O), dtype=torch.float16, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def quant():
        assert lib.mixq_quant_extract(M, K, p(A), p(qA), p(sA), p(fpA), p(ind), O, 0, st) == 0

    nscr = lib.mixq_gemm_scratch_size(M, N, K)   # > 0 only with --variant 72 / 74 / 79 (K split over workgroups)
    if a.scratch_bound:                          # (measurement configurations that split K without a plan of their own)
        nscr = max(nscr, lib.mixq_gemm_scratch_bound())
    scr = torch.zeros(max(nscr, 16), dtype=torch.uint8, device=dev)

    def gemm():
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                           p(scr) if nscr else None, nscr, st) == 0

    if a.qa_mode:
        quant()
        torch.cuda.synchronize()
        q32 = qA.to(torch.int32)
        if a.qa_mode == "offset8":
            q32 = (q32 + 8).clamp_(-128, 127)
        elif a.qa_mode == "abs":
            q32 = q32.abs().clamp_(max=127)
        elif a.qa_mode == "zero_outl":
            q32[:, ind.long()] = 0
        elif a.qa_mode == "offset8_zero_outl":
            q32 = (q32 + 8).clamp_(-128, 127)
            q32[:, ind.long()] = 8
        elif a.qa_mode == "full":
            q32 = torch.randint(-127, 128, q32.shape, device=dev, generator=g, dtype=torch.int32)
        qA.copy_(q32.to(torch.int8))
        print("qA mode", a.qa_mode, "mean |q|", float(qA.float().abs().mean()), "frac negative", float((qA < 0).float().mean()))
        a.what = "gemm"
    if a.what == "decode":
        Wq = torch.randint(0, 256, (K, N), dtype=torch.uint8, device=dev, generator=g)
        for m in (1, 2, 4):
            x = torch.randn((m, K), device=dev, generator=g).to(torch.float16)
            o = torch.empty((m, N), dtype=torch.float16, device=dev)
            fn = lambda: lib.mixq_w8a16_gemm_forward(p(x), p(Wq), p(sW), p(o), m, N, K, st)
            for _ in range(5):
                assert fn() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"decode m={m} N={N} K={K}: {ms*1e3:.1f} us  {N*K/ms/1e6:.0f} GB/s (weight bytes)")
        return
    quant()
    gemm()
    torch.cuda.synchronize()
    if a.check:
        bad = 0
        for _ in range(20):
            quant(); gemm(); torch.cuda.synchronize()
            ref = out.clone()
            out.zero_()
            assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, st) == 0
            torch.cuda.synchronize()
            bad += 0 if torch.equal(ref, out) else 1
        print("bit-identical to the plain launch (20 rounds):", bad == 0, "scratch bytes", nscr)
    if a.stamps:
        timeline(lib, gemm, M, N, dev, a.variant)
    for name, fn in (("quant", quant), ("gemm", gemm)):
        if a.what not in (name, "both"):
            continue
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        if name == "gemm":
            tops = (2.0 * M * N * K + 2.0 * M * N * O) / ms / 1e9
            print(f"gemm  variant={a.variant} M={M} N={N} K={K}: {ms*1e3:.1f} us  {tops:.0f} TOPS  ({tops/5033*100:.1f}% of 5033)")
        else:
            gb = (2.0 * M * K + M * K + 2 * M + 2.0 * M * O) / 1e9
            print(f"quant M={M} K={K}: {ms*1e3:.1f} us  {gb/ms*1e3:.0f} GB/s algorithmic")


def timeline(lib, gemm, M, N, dev, variant=0):
    import numpy as np
    nblk = 4 * ((M + 255) // 256) * ((N + 255) // 256) + 32   # (up to 4 workgroups per tile in the split form)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    lib.mixq_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    gemm()
    torch.cuda.synchronize()
    lib.mixq_debug_set_stamp_buffer(None)
    t = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
    t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]   # only workgroups that ran
    nblk = len(t)
    t0 = t[:, 0].min()
    names = ["prologue", "main loop", "outlier stage", "dequant math", "tile->LDS", "issue stores", "drain stores"]
    if variant == 3:  # persistent kernel: stamps of the second tile of every resident workgroup
        names = ["slices 0-1", "steady slices", "tail slices", "fpW half 1", "dequant math", "next slice 0 + LDS", "stores"]
    if variant in (72, 74, 79):   # K split over workgroups
        names = ["prologue", "main loop", "park", "outlier stage", "arrival wait", "fetch + add", "epilogue"]
    d = np.diff(t, axis=1)
    print(f"stamps: {nblk} blocks; kernel span {(t[:, -1].max() - t0):.0f} ticks; per-block total mean {(t[:, -1] - t[:, 0]).mean():.0f}")
    for i, nme in enumerate(names):
        print(f"   {nme:20s} mean {d[:, i].mean():9.0f}  min {d[:, i].min():9.0f}  max {d[:, i].max():9.0f} ticks")
    starts = np.sort(t[:, 0] - t0)
    print("   block start times (ticks): " + " ".join(f"{starts[int(q * (nblk - 1))]:.0f}" for q in (0, .1, .2, .4, .6, .8, 1.0)))


if __name__ == "__main__":
    main()

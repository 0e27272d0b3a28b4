#!/usr/bin/env python3
"""Round 6: in-kernel timeline of the mid-M GEMM builds (gemm_mid_kernels.hip) on the 100 MHz wall clock, warm and cold weights:
where do the microseconds of a 256 x 12288 x 4096 launch go (entry -> first slices handed over -> half of K -> last MFMA -> halves
swapped -> stores issued -> acknowledged), per workgroup: min / mean / max.  usage: python tools/experimental/r06_mid_timeline.py [--M 256 --N 12288 --K 4096 --knobs 1272,1401]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402

NAMES = ["entry", "first slices handed over", "(qA loader's wait words)", "last MFMA issued", "halves swapped / outliers staged", "stores issued",
         "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=256)
    ap.add_argument("--N", type=int, default=12288)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--knobs", default="1272")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    M, N, K, O = a.M, a.N, a.K, 128
    g = torch.Generator(device=dev).manual_seed(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
    fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
    for k in [int(x) for x in a.knobs.split(",") if x]:
        lib.mixq_debug_set_gemm_variant(k)
    nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
    for cold in (False, True):
        Ws = [W] + ([W.clone() for _ in range((320 << 20) // (N * K) + 1)] if cold else [])
        stamps = [torch.zeros(4096 * 8, dtype=torch.int64, device=dev) for _ in range(2)]
        turn = [0]

        def fn(par=None):
            if par is not None:
                lib.mixq_debug_set_stamp_buffer(p(stamps[par]))
            w = Ws[turn[0] % len(Ws)]
            turn[0] += 1
            assert lib.mixq_gemm_mixed_scratch(p(qA), p(w), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr) if nscr else None, nscr, st) == 0
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        kern = lib.mixq_debug_last_gemm_kernel().decode()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"# M={M} N={N} K={K} knobs {a.knobs} {'COLD' if cold else 'warm'}: {e0.elapsed_time(e1) * 5:.2f} us per launch, no stamps   [{kern}]")
        for i in range(40):
            fn(i & 1)
        lib.mixq_debug_set_stamp_buffer(None)
        torch.cuda.synchronize()
        raw = stamps[1].cpu().numpy().reshape(-1, 8)
        raw = raw[raw[:, 0] > 0]
        for slot, who in ((7, "W loader"), (2, "qA loader")):
            v = raw[:, slot].astype(np.uint64)
            w, b = (v >> np.uint64(32)).astype(np.float64) * 0.01, (v & np.uint64(0xffffffff)).astype(np.float64) * 0.01
            if w.max() < 1e4:
                print(f"   {who:10s}: us in `s_waitcnt vmcnt` (its copies landing) {w.min():6.2f} {w.mean():6.2f} {w.max():6.2f} | us at the barriers {b.min():6.2f} {b.mean():6.2f} {b.max():6.2f}")
        rec = lambda b: (lambda t: t[t[:, 0] > 0][:, :7] * 0.01)(b.cpu().numpy().reshape(-1, 8).astype(np.float64))
        last, prev = rec(stamps[1]), rec(stamps[0])
        t0 = last[:, 0].min()
        print(f"   previous launch's last acknowledged store -> first entry: {t0 - prev[:, 6].max():6.2f} us; {len(last)} workgroups; us after the first entry (min / mean / max)")
        for i, n in enumerate(NAMES):
            c = last[:, i] - t0
            print(f"   {n:36s} {c.min():6.2f} {c.mean():6.2f} {c.max():6.2f}")
        d = last[:, 1:] - last[:, :-1]
        print("   per-workgroup spans (mean): " + " | ".join(f"{NAMES[i + 1]}: {d[:, i].mean():5.2f}" for i in range(6)))
    lib.mixq_debug_reset()


if __name__ == "__main__":
    main()

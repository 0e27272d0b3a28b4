#!/usr/bin/env python3
"""Partial-round table of the fused int8 GEMM (VERDICT r3, next #1a): every BASELINE (N, K) at M in {8192, 16384, 65536}.

Per case: 256 x 256 tiles, rounds = tiles / CUs, the kernel the selection takes, steady-state us per launch (>= --secs of
back-to-back launches after a warm-up that brings the chip to its power-capped clock), TOP/s, fraction of the 5033 TOP/s
int8 peak, the same with the K split over workgroups switched off (variant 70), and -- from one stamped launch -- the CU
occupancy of the launch = sum of workgroup life times / (CUs x kernel span).  `wre` = the rate of the same (N, K) at
M = 65536 (tiles_m = 256 = the CU count: whole rounds whatever N), the "whole-round-equivalent" a shorter chunk is held to.

usage: python tools/round_table.py [--secs 0.5] [--Ms 8192,16384,65536] [--shapes llama,qwen,l70,tp8] [--out FILE]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402

SHAPES = {
    "llama": [("L7B qkv", 12288, 4096), ("L7B gate", 11008, 4096), ("L7B proj", 4096, 11008)],
    "qwen": [("Q7B qkv", 4608, 3584), ("Q7B gate", 18944, 3584), ("Q7B proj", 3584, 18944)],
    "l70": [("70B qkv", 10240, 8192), ("70B gate", 28672, 8192), ("70B proj", 8192, 28672)],
    "tp8": [("70B/8 qkv", 1280, 8192), ("70B/8 gate", 3584, 8192), ("70B/8 proj", 1024, 28672)],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", type=float, default=0.5)
    ap.add_argument("--Ms", default="8192,16384,65536")
    ap.add_argument("--shapes", default="llama,qwen,l70,tp8")
    ap.add_argument("--variants", default="0,70", help="knobs to time per case: 0 = automatic, 70 = K split off, 79 = auto again")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.mixq_debug_last_gemm_kernel.restype = ctypes.c_char_p
    lib.mixq_gemm_scratch_size.restype = ctypes.c_size_t
    lib.mixq_gemm_scratch_bound.restype = ctypes.c_size_t
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device=dev).manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    Ms = [int(x) for x in a.Ms.split(",")]
    variants = [int(x) for x in a.variants.split(",")]
    O = 128
    lines = []

    def emit(s):
        print(s, flush=True)
        lines.append(s)

    emit(f"# round_table: {cus} CUs, >= {a.secs} s of back-to-back launches per cell; us / TOPS / frac of 5033")
    emit("# shape | M | tiles | rounds | kernel | us | TOPS | frac | (K split off) us frac | occupancy | frac / wre")
    scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
    rows = []
    for grp in a.shapes.split(","):
        for name, N, K in SHAPES[grp]:
            W = torch.randn((N, K), device=dev, generator=g).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
            ind = torch.randperm(K, device=dev, generator=g)[:O].to(torch.int32)
            sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
            fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
            per_m = {}
            for M in sorted(Ms, reverse=True):
                A = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
                A[:, ind.long()] *= 20
                qA = torch.empty((M + 16, K), dtype=torch.int8, device=dev)[:M]
                sA = torch.empty(M, dtype=torch.float16, device=dev)
                fpA = torch.empty((M, O), dtype=torch.float16, device=dev)
                out = torch.empty((M, N), dtype=torch.float16, device=dev)
                assert lib.mixq_quant_extract(M, K, p(A), p(qA), p(sA), p(fpA), p(ind), O, 0, st) == 0
                del A
                res = {}
                for v in variants:
                    lib.mixq_debug_set_gemm_variant(v if v != 0 else 79)
                    nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
                    assert nscr <= scr.numel()

                    def gemm():
                        assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                                           p(scr) if nscr else None, nscr, st) == 0

                    gemm()
                    kern = lib.mixq_debug_last_gemm_kernel().decode()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gemm(); gemm(); e1.record(); torch.cuda.synchronize()
                    est = e0.elapsed_time(e1) / 2 * 1e-3
                    n_warm = max(3, int(0.3 / est))
                    n = max(5, int(a.secs / est))
                    for _ in range(n_warm):
                        gemm()
                    e0.record()
                    for _ in range(n):
                        gemm()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) / n * 1e3
                    tops = (2.0 * M * N * K + 2.0 * M * N * O) / us / 1e6
                    # one stamped launch: occupancy of the launch
                    tiles = ((M + 255) // 256) * ((N + 255) // 256)
                    nblk = 4 * tiles + 32
                    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
                    lib.mixq_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
                    gemm()
                    torch.cuda.synchronize()
                    lib.mixq_debug_set_stamp_buffer(None)
                    t = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
                    ran = (t[:, 0] > 0) & (t[:, 7] > 0)
                    occ, occs = float("nan"), []
                    for x in range(8):   # the stamps are the shader clock of the block's XCD (block b -> XCD b % 8): one time base per XCD
                        sel = ran & (np.arange(nblk) % 8 == x)
                        if sel.any():
                            span = t[sel, 7].max() - t[sel, 0].min()
                            occs.append((t[sel, 7] - t[sel, 0]).sum() / (cus / 8 * span))
                    if occs:
                        occ = float(np.mean(occs))
                    t = t[ran]
                    res[v] = (us, tops, kern, occ, len(t))
                    del buf
                per_m[M] = res
                tiles = ((M + 255) // 256) * ((N + 255) // 256)
                rows.append((name, N, K, M, tiles, res))
                del qA, sA, fpA, out
            wre = per_m.get(65536, {}).get(variants[0], (None, None))[1]
            for (nm, N_, K_, M, tiles, res) in [r for r in rows if r[0] == name]:
                us, tops, kern, occ, nb = res[variants[0]]
                short = kern.split(" ")[0].replace("gemm_w8a8o16_", "")
                alt = " ".join(f"[{v}: {res[v][0]:.1f} us {res[v][1] / 5033:.3f}]" for v in variants[1:])
                emit(f"{nm:11s} {N_:6d}x{K_:<6d} M={M:6d} tiles={tiles:5d} rounds={tiles / cus:6.2f} {short:22s} blocks={nb:5d} "
                     f"{us:8.1f} us {tops:6.0f} TOPS frac={tops / 5033:.3f} {alt} occ={occ:.3f} "
                     f"vs_wre={(tops / wre if wre else float('nan')):.3f}")
            del W
            torch.cuda.empty_cache()
    lib.mixq_debug_reset()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

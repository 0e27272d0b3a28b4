#!/usr/bin/env python3
"""Soak test of the K split over workgroups: random shapes for which the automatic rule (or a forced factor) selects the
split form, each launched several times on one shared scratch and compared bit for bit with the one-workgroup kernels.
usage: python tools/splitk_soak.py [--shapes 200] [--seed 0]"""
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
    ap.add_argument("--shapes", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(a.seed)
    cg = torch.Generator().manual_seed(a.seed)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    scr = torch.zeros(72 << 20, dtype=torch.uint8, device=dev)
    done = bad = tried = 0
    hist = {}
    while done < a.shapes and tried < 100 * a.shapes:
        tried += 1
        small = int(torch.randint(0, 2, (1,), generator=cg)) == 1   # half of the shapes: decode batches / short chunks
        M = int(torch.randint(5, 300, (1,), generator=cg)) if small else int(torch.randint(129, 6000, (1,), generator=cg))
        N = int(torch.randint(1, 1200, (1,), generator=cg)) * 16
        K = int(torch.randint(8, 1800, (1,), generator=cg)) * 16
        O = [128, 128, 0, 64][int(torch.randint(0, 4, (1,), generator=cg))]
        mode = [79, 79, 72, 74, 78][int(torch.randint(0, 5, (1,), generator=cg))]
        xmode = [69, 69, 62, 64, 68, 66][int(torch.randint(0, 6, (1,), generator=cg))]   # the small-tile form's knob
        lib.mixq_debug_set_gemm_variant(mode)
        lib.mixq_debug_set_gemm_variant(xmode)
        n = lib.mixq_gemm_scratch_size(M, N, K)
        if n == 0 or M * N > (1 << 27) or M * K > (1 << 27) or N * K > (1 << 28):
            continue
        qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 1e-4).to(torch.float16)
        fpA = torch.randn((M, max(O, 8)), device=dev, generator=g).to(torch.float16)[:, :O].contiguous() if O else None
        fpW = (torch.randn((N, max(O, 8)), device=dev, generator=g) * 0.02).to(torch.float16)[:, :O].contiguous() if O else None
        ref = torch.empty((M, N), dtype=torch.float16, device=dev)
        lib.mixq_debug_set_gemm_variant(70)
        assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(ref), M, N, K, O, st) == 0
        lib.mixq_debug_set_gemm_variant(mode)
        lib.mixq_debug_set_gemm_variant(xmode)
        ok = True
        for _ in range(4):
            o2 = torch.empty((M, N), dtype=torch.float16, device=dev)
            assert lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(o2), M, N, K, O, p(scr), n,
                                               st) == 0
            torch.cuda.synchronize()
            if ok and not torch.equal(o2, ref):
                ok, out = False, o2
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        hist[(mode, xmode)] = hist.get((mode, xmode), 0) + 1
        if not ok:
            bad += 1
            # which side is wrong?  recompute a few differing outputs exactly
            diff = (out != ref).nonzero()
            m, nn = int(diff[0, 0]), int(diff[0, 1])
            acc = int((qA[m].to(torch.int64) * W[nn].to(torch.int64)).sum())
            side = float((fpA[m].float() * fpW[nn].float()).sum()) if O else 0.0
            want = acc * float(sW[nn]) * float(sA[m]) + side
            print(f"MISMATCH M={M} N={N} K={K} (K%128={K % 128}) O={O} mode={mode}/{xmode} tiles={tiles} ndiff={len(diff)} at ({m},{nn}): "
                  f"split={float(out[m, nn]):.4f} plain={float(ref[m, nn]):.4f} exact={want:.4f} "
                  f"rows {int(diff[:,0].min())}..{int(diff[:,0].max())} cols {int(diff[:,1].min())}..{int(diff[:,1].max())}", flush=True)
        elif os.environ.get("SOAK_VERBOSE"):
            print(f"ok M={M} N={N} K={K} (K%128={K % 128}) O={O} mode={mode} tiles={tiles}", flush=True)
        done += 1
    lib.mixq_debug_set_gemm_variant(79)
    lib.mixq_debug_set_gemm_variant(69)
    print(f"{done} shapes x 4 launches (modes {hist}), {bad} mismatches; hand-over words left zero: "
          f"{int(scr[:16384].to(torch.int32).sum()) == 0}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""128 x 256 ping-pong tiles with K split over 2 / 4 / 8 workgroups per tile (knobs 922 / 923 / 924) against the automatic
selection with the form off (921): bit-identical output?  us per call (HIP events, 300 iterations), one box.
usage: python tools/experimental/pp128_xsplit_probe.py [--shapes "M N K;..."]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402

DEFAULT = ("256 4096 11008;384 4096 11008;512 4096 11008;768 4096 11008;1024 4096 11008;"
           "256 4096 4096;512 4096 4096;1024 4096 4096;2048 4096 4096;"
           "256 3584 18944;512 3584 18944;1024 3584 18944;256 8192 28672;512 8192 28672;256 1024 28672;512 1024 28672;1024 1024 28672;"
           "256 4096 8192;512 4096 8192;1024 4096 8192;256 8192 8192;512 8192 8192;256 5120 5120;512 5120 5120;1024 5120 5120;"
           "256 12288 4096;384 12288 4096;256 11008 4096;200 4096 11008;300 3584 18944;640 4096 11008")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=DEFAULT)
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    nscr = int(lib.mixq_gemm_scratch_bound())
    scr = torch.zeros(nscr, dtype=torch.uint8, device=dev)
    for shape in a.shapes.split(";"):
        M, N, K = (int(x) for x in shape.split())
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        O = 128
        qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        sA = (torch.rand(M, device=dev, generator=g) * 1e-2 + 1e-3).to(torch.float16)
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
        fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
        fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)

        def run():
            rc = lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr), nscr, st)
            assert rc == 0, rc
        cells, ref = [], None
        for knob in (921, 922, 923, 924):
            lib.mixq_debug_reset()
            lib.mixq_debug_set_gemm_variant(knob)
            out.zero_()
            run()
            torch.cuda.synchronize()
            kern = lib.mixq_debug_last_gemm_kernel().decode()
            if knob == 921:
                ref, refk = out.clone(), kern
                tag = "off"
            else:
                if "XSP" not in kern:
                    cells.append(f"x{2 << (knob - 922)}=n/a")
                    continue
                same = torch.equal(out, ref)
                tag = f"x{2 << (knob - 922)}" + ("" if same else "(DIFFERS)")
            for _ in range(10):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            cells.append(f"{tag}={e0.elapsed_time(e1) / a.iters * 1e3:.1f}")
        assert int(scr[:16384].view(torch.int32).abs().sum()) == 0, "hand-over words not left zero"
        print(f"M={M:5d} N={N:6d} K={K:6d}: " + " ".join(cells) + f"   [off: {refk}]", flush=True)
    lib.mixq_debug_reset()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Whole-operator timing through mixq_enqueue (quantiser + GEMM), back-to-back calls.
usage: python tools/enqueue_bench.py --M 32 --N 4096 --K 4096 [--variant 80|81] [--iters 2000]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--variant2", type=int, default=None, help="a second knob set after --variant")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--stamps", action="store_true", help="one-launch operator (variant 81): print the in-kernel timeline "
                    "(wall-clock ticks of 10 ns) of one call")
    ap.add_argument("--graph", type=int, default=0, help="capture this many calls into one HIP graph and time replays "
                    "(device-paced: no host launch cost per call)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    if a.variant is not None:
        lib.mixq_debug_set_gemm_variant(a.variant)
    if a.variant2:
        lib.mixq_debug_set_gemm_variant(a.variant2)
    M, N, K = a.M, a.N, a.K
    g = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn((N, K), device=dev, generator=g).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
    ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
    W[:, ind.long()] = 0
    A = torch.randn((M, K), device=dev, generator=g)
    A[:, ind.long()] *= 20
    A = A.to(torch.float16)
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpW = (torch.randn((N, 128), device=dev, generator=g) * 0.02).to(torch.float16)
    qw = torch.zeros((K, N), dtype=torch.uint8, device=dev)
    ins = [A, W.view(torch.float16), sW, fpW, ind.view(torch.float16), qw.view(torch.float16), sW]
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    in_desc = (TensorDesc * 7)(*[TensorDesc.make(t.shape) for t in ins])
    out_desc = TensorDesc.make(out.shape)
    in_ptrs = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in ins])
    out_ptrs = (ctypes.c_void_p * 1)(out.data_ptr())
    h = ctypes.c_void_p(lib.mixq_create(M, N, K))
    ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():  # (`st` is read at call time: the graph branch rebinds it to the capture stream)
        return lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), in_ptrs, out_ptrs, ctypes.c_void_p(ws.data_ptr()), st)
    for _ in range(50):
        assert run() == 0
    torch.cuda.synchronize()
    if a.stamps:
        import numpy as np
        nblk = (N + 15) // 16
        buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        lib.mixq_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        lib.mixq_debug_set_stamp_buffer(None)
        t = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
        t0 = t[:, 0].min()
        names = ["start->W issued", "quantise+publish", "wait for flags", "GEMM (qA loads + MFMA)", "LDS hand-over",
                 "epilogue", "re-arm"]
        print(f"kernel span {(t[:, 7].max() - t0) * 10:.0f} ns; first start -> last start {(t[:, 0].max() - t0) * 10:.0f} ns")
        d = np.diff(t, axis=1) * 10
        for i, nme in enumerate(names):
            print(f"   {nme:26s} mean {d[:, i].mean():8.0f}  min {d[:, i].min():8.0f}  max {d[:, i].max():8.0f} ns")
        q = t[:M]
        print(f"   quantiser workgroups: publish done at {((q[:, 2] - t0) * 10).mean():.0f} ns (max {((q[:, 2] - t0) * 10).max():.0f});"
              f" all workgroups through the wait at {((t[:, 3] - t0) * 10).mean():.0f} ns (max {((t[:, 3] - t0) * 10).max():.0f})")
    if a.graph:
        side = torch.cuda.Stream()
        st = ctypes.c_void_p(side.cuda_stream)
        with torch.cuda.stream(side):
            run()
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(a.graph):
                    run()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        reps = max(1, a.iters // a.graph)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * a.graph)
        print(f"{dt*1e6:.2f} us/call (graph of {a.graph})  weights {N*K/dt/1e9:.0f} GB/s  kernel: "
              f"{lib.mixq_debug_last_gemm_kernel().decode()}")
        return
    t0 = time.perf_counter()
    for _ in range(a.iters):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"{dt*1e6:.2f} us/call  weights {N*K/dt/1e9:.0f} GB/s  kernel: {lib.mixq_debug_last_gemm_kernel().decode()}")


if __name__ == "__main__":
    main()

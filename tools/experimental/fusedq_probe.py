#!/usr/bin/env python3
"""One-launch decode-batch operator (csrc/gemm_fusedq_kernels.hip) against the two-launch form: same bits, time per call
(HIP graph of 100 mixq_enqueue calls, device-paced) and the in-kernel timeline of the one-launch kernel.
usage: python tools/fusedq_probe.py [--shapes "32 4096 4096;16 4096 4096"]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402

FN = ["entry", "weights + row requested", "own row(s) published", "through the flag wait", "last MFMA", "stores issued",
      "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="32 4096 4096;16 4096 4096;8 4096 4096;32 4096 1024;24 1024 4096;32 5120 5120")
    ap.add_argument("--calls", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for shp in a.shapes.split(";"):
        M, N, K = (int(v) for v in shp.split())
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        W = torch.randn((N, K), device=dev, generator=g).mul_(32).round_().clamp_(-127, 127).to(torch.int8)
        ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
        W[:, ind.long()] = 0
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
        fpW = (torch.randn((N, 128), device=dev, generator=g) * 0.02).to(torch.float16)
        qw = torch.zeros((K, N), dtype=torch.uint8, device=dev)
        As = []
        for i in range(4):
            A = torch.randn((M, K), device=dev, generator=g)
            A[:, ind.long()] *= 20
            As.append(A.to(torch.float16))
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        h = ctypes.c_void_p(lib.mixq_create(M, N, K))
        ws = torch.full((max(lib.mixq_workspace_size(h, M, N, K), 16),), 0xAB, dtype=torch.uint8, device=dev)
        out_desc = TensorDesc.make(out.shape)
        out_ptrs = (ctypes.c_void_p * 1)(out.data_ptr())

        def call(A, st):
            ins = [A, W.view(torch.float16), sW, fpW, ind.view(torch.float16), qw.view(torch.float16), sW]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(t.shape) for t in ins])
            in_ptrs = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in ins])
            rc = lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), in_ptrs, out_ptrs, ctypes.c_void_p(ws.data_ptr()), st)
            assert rc == 0, rc

        st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        res = {}
        for mode in (880, 881, 882):
            lib.mixq_debug_set_gemm_variant(mode)
            outs = []
            for rep in range(3):
                for A in As:                      # changing data on one workspace: flags / epochs must follow
                    out.fill_(float("nan"))
                    call(A, st0)
                    torch.cuda.synchronize()
                    outs.append(out.clone())
            res[mode] = (outs, lib.mixq_debug_last_gemm_kernel().decode())
        same1 = all(torch.equal(x, y) for x, y in zip(res[880][0], res[881][0]))
        same2 = all(torch.equal(x, y) for x, y in zip(res[880][0], res[882][0]))
        print(f"## M={M} N={N} K={K}: one-launch == two-launch bits: {same1} (help path: {same2})   [{res[881][1]}]")

        def timed(mode):
            lib.mixq_debug_set_gemm_variant(mode)
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                call(As[0], ctypes.c_void_p(s.cuda_stream))
                s.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    stp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                    for i in range(a.calls):
                        call(As[i & 3], stp)
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (30 * a.calls)

        t2, t1 = timed(880), timed(881)
        print(f"   two launches {t2:6.2f} us / call   one launch {t1:6.2f} us / call   ({(t1 / t2 - 1) * 100:+.1f} %)")
        # timeline of the one-launch kernel (last call of a stamped graph)
        lib.mixq_debug_set_gemm_variant(881)
        G = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        lib.mixq_debug_set_stamp_buffer(ctypes.c_void_p(G.data_ptr()))
        for i in range(20):
            call(As[i & 3], st0)
        torch.cuda.synchronize()
        lib.mixq_debug_set_stamp_buffer(None)
        t = G.cpu().numpy().reshape(-1, 8).astype(np.float64)
        t = t[t[:, 0] > 0][:, :7] * 0.01
        if len(t):
            t0 = t[:, 0].min()
            for i, n in enumerate(FN):
                c = t[:, i] - t0
                print(f"      {n:28s} {c.min():6.2f} {c.mean():6.2f} {c.max():6.2f}")
        lib.mixq_debug_set_gemm_variant(881)
        lib.mixq_destroy(h)


if __name__ == "__main__":
    main()

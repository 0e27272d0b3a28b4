#!/usr/bin/env python3
"""In-kernel timeline of the small-M operator (BASELINE configs[0]: 4096 x 4096 at bs = 32) on the 100 MHz wall clock:
where the microseconds of the two launches of mixq_enqueue go.

A HIP graph of 100 mixq_enqueue calls is replayed (device-paced); calls alternate between two stamp-buffer sets, so the last
two calls of the last replay survive: gap GEMM(98) -> quantiser(99), the quantiser's stamps, gap quantiser -> GEMM, the
GEMM's stamps.  usage: python tools/small_m_timeline.py [--M 32 --N 4096 --K 4096]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402

QN = ["entry", "row loads issued", "gather issued", "amax reduced (row data here)", "stores issued", "stores acknowledged"]
GN = ["entry", "epilogue operands requested", "first 16 k-steps multiplied", "last MFMA", "LDS hand-over", "stores issued",
      "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--calls", type=int, default=100)
    ap.add_argument("--knobs", default="", help="comma list of mixq_debug_set_gemm_variant knobs to set first")
    ap.add_argument("--cold", action="store_true", help="cycle through enough weight copies to exceed the Infinity Cache")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    for _k in [int(x) for x in a.knobs.split(",") if x]:
        lib.mixq_debug_set_gemm_variant(_k)
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
    copies = ((320 << 20) // (N * K) + 2) if a.cold else 1
    alts = [W] + [W.clone() for _ in range(copies - 1)]
    ptr_sets = []
    for w in alts:
        v = [t.data_ptr() for t in ins]
        v[1] = w.data_ptr()
        ptr_sets.append((ctypes.c_void_p * 7)(*v))
    turn = [0]
    out_ptrs = (ctypes.c_void_p * 1)(out.data_ptr())
    h = ctypes.c_void_p(lib.mixq_create(M, N, K))
    ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
    nb = 4096
    Q = [torch.zeros(nb * 8, dtype=torch.int64, device=dev) for _ in range(2)]
    G = [torch.zeros(nb * 8, dtype=torch.int64, device=dev) for _ in range(2)]

    def run(st, par=None):
        if par is not None:
            lib.mixq_debug_set_quant_stamp_buffer(ctypes.c_void_p(Q[par].data_ptr()))
            lib.mixq_debug_set_stamp_buffer(ctypes.c_void_p(G[par].data_ptr()))
        rc = lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), ptr_sets[turn[0] % copies], out_ptrs, ctypes.c_void_p(ws.data_ptr()), st)
        turn[0] += 1
        assert rc == 0

    st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        run(st0)
    torch.cuda.synchronize()
    print(f"# M={M} N={N} K={K}  kernel: {lib.mixq_debug_last_gemm_kernel().decode()}")

    def graph_of(stamped):
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                stp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for i in range(a.calls):
                    run(stp, (i & 1) if stamped else None)
        lib.mixq_debug_set_quant_stamp_buffer(None)
        lib.mixq_debug_set_stamp_buffer(None)
        return gr

    def time_graph(gr, reps=30):
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (reps * a.calls)

    plain = graph_of(False)
    print(f"operator, graph of {a.calls} calls, no stamps : {time_graph(plain):6.2f} us / call")
    stamped = graph_of(True)
    print(f"operator, graph of {a.calls} calls, stamped   : {time_graph(stamped):6.2f} us / call")
    torch.cuda.synchronize()

    def rec(buf, nslots):
        t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
        return t[t[:, 0] > 0][:, :nslots] * 0.01   # 100 MHz ticks -> us

    last, prev = (a.calls - 1) & 1, a.calls & 1
    q, gm, gprev = rec(Q[last], 6), rec(G[last], 7), rec(G[prev], 7)
    t_prev_end = gprev[:, 6].max()
    q0 = q[:, 0].min()
    print(f"\nGEMM(call {a.calls - 2}) last store acknowledged -> quantiser(call {a.calls - 1}) first workgroup entry: "
          f"{q0 - t_prev_end:6.2f} us   [kernel boundary]")
    print(f"quantiser: {len(q)} workgroups; us after its first entry (min / mean / max over workgroups)")
    for i, n in enumerate(QN):
        c = q[:, i] - q0
        print(f"   {n:34s} {c.min():6.2f} {c.mean():6.2f} {c.max():6.2f}")
    g0 = gm[:, 0].min()
    print(f"quantiser last store acknowledged -> GEMM first workgroup entry: {g0 - q[:, 5].max():6.2f} us   [kernel boundary]")
    print(f"GEMM: {len(gm)} workgroups; us after its first entry (min / mean / max over workgroups)")
    for i, n in enumerate(GN):
        c = gm[:, i] - g0
        print(f"   {n:34s} {c.min():6.2f} {c.mean():6.2f} {c.max():6.2f}")
    print(f"call period (GEMM end to GEMM end): {gm[:, 6].max() - t_prev_end:6.2f} us")
    lib.mixq_destroy(h)


if __name__ == "__main__":
    main()

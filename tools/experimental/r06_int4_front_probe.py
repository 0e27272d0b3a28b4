#!/usr/bin/env python3
"""Round 6 probe: what does the fused int4 quantiser front cost?  One shape, cold weights (rotation over > 320 MiB of packed copies), device-paced graphs:
the packed-stream GEMM alone, the two launches, mixq_int4_linear_forward, and the latter without the weight touches / without the quantiser body."""
import ctypes, os, sys
os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for (N, K) in ((12288, 4096), (4096, 11008)):
    for bs in (1, 2, 4, 8):
        ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev, generator=g) for _ in range((320 << 20) // (N * K // 2) + 1)]
        sw = (torch.rand(N, device=dev, generator=g) * 1e-2 + 1e-3).to(torch.float16)
        x = torch.randn((bs, K), device=dev, generator=g).to(torch.float16)
        q4 = torch.empty((bs, K // 2), dtype=torch.uint8, device=dev); sa = torch.empty(bs, dtype=torch.float16, device=dev)
        out = torch.empty((bs, N), dtype=torch.float16, device=dev)
        st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        assert lib.mixq_int4quant(bs, K, p(x), p(q4), p(sa), st0) == 0
        turn = [0]
        def gemm(st):
            w = ws[turn[0] % len(ws)]; turn[0] += 1
            assert lib.mixq_int4_fused_dequantize(p(q4), p(w), p(sa), p(sw), None, p(out), bs, N, K // 2, None, st) == 0
        def two(st):
            assert lib.mixq_int4quant(bs, K, p(x), p(q4), p(sa), st) == 0
            gemm(st)
        def one(st):
            w = ws[turn[0] % len(ws)]; turn[0] += 1
            assert lib.mixq_int4_linear_forward(p(x), p(w), p(sa), p(q4), p(sw), None, p(out), bs, N, K // 2, 0, None, st) == 0
        res = {}
        for name, fn, knob in (("gemm", gemm, 877), ("two", two, 877), ("one", one, 877), ("one-notouch", one, 878), ("one-nobody", one, 879), ("one-neither", one, 880)):
            lib.mixq_debug_set_gemm_variant(knob)
            turn[0] = 0
            res[name] = min(bench.graph_time_us(fn, dev, calls=len(ws) * 4, reps=10) for _ in range(2))
        lib.mixq_debug_set_gemm_variant(877)
        print(f"{N}x{K} bs {bs}: " + "  ".join(f"{k} {v:6.2f}" for k, v in res.items()), flush=True)

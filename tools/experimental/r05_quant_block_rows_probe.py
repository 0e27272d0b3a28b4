import ctypes, os, sys
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mixq_tensorrt_llm_amd import _lib
MS = [int(x) for x in os.environ.get("MS", "2048,8192,16384,65536").split(",")]; KNOBS = [int(x) for x in os.environ.get("KNOBS", "1301,1306,1312").split(",")]
lib = _lib.load(); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for K in (4096, 8192, 11008):
    ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
        outl = torch.empty((M, 128), dtype=torch.float16, device=dev)
        q = torch.empty((M, K), dtype=torch.int8, device=dev); sc = torch.empty(M, dtype=torch.float16, device=dev)
        cells = []
        for knob in KNOBS:
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
            def f(st):
                assert lib.mixq_quant_extract(M, K, p(x), p(q), p(sc), p(outl), p(ind), 128, 0, st) == 0
            cells.append(bench.graph_time_us(f, dev, calls=20 if M >= 16384 else 100, reps=10))
        gb = (3 * M * K + 258 * M) / 1e3
        print(f"quant_extract M={M:6d} K={K:5d}: " + " | ".join(f"knob {k}: {c:8.2f} us ({gb / c:5.0f} GB/s)" for k, c in zip(KNOBS, cells)), flush=True)
lib.mixq_debug_reset()

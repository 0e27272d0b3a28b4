import ctypes, os, sys
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mixq_tensorrt_llm_amd import _lib
MS = [int(x) for x in os.environ.get("MS", "64,96,128,256,512,1024,2048,4096").split(",")]; KNOBS = [int(x) for x in os.environ.get("KNOBS", "1301,1306").split(",")]
lib = _lib.load(); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for K in (4096, 8192):
    gamma = (torch.rand(K, device=dev, generator=g) + 0.5).to(torch.float16)
    ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
    for M in MS:
        x = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
        out = torch.empty_like(x); outl = torch.empty((M, 128), dtype=torch.float16, device=dev)
        q = torch.empty((M, K), dtype=torch.int8, device=dev); sc = torch.empty(M, dtype=torch.float16, device=dev)
        cells = []
        for knob in KNOBS:
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
            def f(st):
                assert lib.mixq_rmsnorm_extract_quant(M, K, p(x), p(gamma), p(out), ctypes.c_float(1e-5), p(ind), 128, p(outl), p(q), p(sc), st) == 0
            cells.append(bench.graph_time_us(f, dev))
        print(f"rmsnorm_extract_quant M={M:5d} K={K}: " + " | ".join(f"knob {k}: {c:7.2f} us" for k, c in zip(KNOBS, cells)), flush=True)
lib.mixq_debug_reset()

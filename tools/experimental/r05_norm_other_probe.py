import ctypes, os, sys
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for K in (4096, 8192):
    gamma = (torch.rand(K, device=dev, generator=g) + 0.5).to(torch.float16)
    ind = torch.randperm(K, device=dev, generator=g)[:128].to(torch.int32)
    for M in (128, 2048, 16384, 65536):
        x = torch.randn((M, K), device=dev, generator=g).to(torch.float16)
        out = torch.empty_like(x); outl = torch.empty((M, 128), dtype=torch.float16, device=dev)
        q4 = torch.empty((M, K // 2), dtype=torch.uint8, device=dev); sc = torch.empty(M, dtype=torch.float16, device=dev)
        row = []
        for name, fn in (("rmsnorm", lambda st: lib.mixq_rmsnorm(M, K, p(x), p(gamma), p(out), ctypes.c_float(1e-5), st)),
                         ("rmsnorm_extract_quant4", lambda st: lib.mixq_rmsnorm_extract_quant4(M, K, p(x), p(gamma), p(out), ctypes.c_float(1e-5), p(ind), 128, p(outl), p(q4), p(sc), st))):
            cells = []
            for knob in (1301, 1300):
                lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
                def f(st, fn=fn):
                    assert fn(st) == 0
                cells.append(bench.graph_time_us(f, dev, calls=20 if M >= 16384 else 100, reps=10))
            row.append(f"{name}: 64-row rule {cells[0]:8.2f} us, block per row {cells[1]:8.2f} us")
        print(f"M={M:6d} K={K}: " + " | ".join(row), flush=True)
lib.mixq_debug_reset()

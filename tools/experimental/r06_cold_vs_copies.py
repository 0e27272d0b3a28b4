"""round 6: what makes a mid-M GEMM 'cold'?  Time per launch against the NUMBER of weight copies taken in rotation: a step between 4 and 6 copies of 48 MiB (192 -> 288 MiB, across the 256-MiB Infinity
Cache) means residency; a gradual rise from 2 copies on would mean address translation (every copy lives on its own pages)."""
import ctypes, os, sys, torch
os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
O = 128
for (M, N, K) in ((256, 12288, 4096), (256, 4096, 11008), (32, 12288, 4096)):
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    Ws = [W] + [W.clone() for _ in range(15)]
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
    fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
    def gemm(w):
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(w), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr) if nscr else None, nscr, st) == 0
    row = []
    for ncopy in (1, 2, 3, 4, 5, 6, 8, 12, 16):
        for i in range(3 * ncopy + 10): gemm(Ws[i % ncopy])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 600
        e0.record()
        for i in range(n): gemm(Ws[i % ncopy])
        e1.record(); torch.cuda.synchronize()
        row.append((ncopy, e0.elapsed_time(e1) / n * 1e3))
    print(f"M={M} N={N} K={K} ({N * K / 2**20:.0f} MiB per copy) [{lib.mixq_debug_last_gemm_kernel().decode().split(' ')[0]}]: " + "  ".join(f"{c} x: {t:.1f}" for c, t in row), flush=True)

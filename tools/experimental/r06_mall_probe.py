"""round 6: is the cold penalty of a mid-M GEMM (25.9 vs 20.6 us at 256 x 12288 x 4096) about the Infinity Cache at all?  Per iteration: a full-chip read of
weight copy i (torch reduction) then the GEMM on copy i (rotation over > 320 MiB), against the read alone and the GEMM alone; also the GEMM on copy i after a read of copy i + 1
(control: the same extra launch, no residency)."""
import ctypes, os, sys, torch
os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
for (M, N, K) in ((256, 12288, 4096), (256, 11008, 4096), (256, 4096, 11008), (512, 4096, 4096)):
    O = 128
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    Ws = [W] + [W.clone() for _ in range((320 << 20) // (N * K) + 1)]
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
    fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
    sink = torch.empty((), dtype=torch.int64, device=dev)
    def gemm(w):
        assert lib.mixq_gemm_mixed_scratch(p(qA), p(w), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, p(scr) if nscr else None, nscr, st) == 0
    def read(w):
        torch.sum(w.view(torch.int32).view(-1), dim=(0,), dtype=torch.int64, out=sink)
    def bench(fn, n=300):
        for i in range(20): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    L = len(Ws)
    t_g_cold = bench(lambda i: gemm(Ws[i % L]))
    t_g_warm = bench(lambda i: gemm(Ws[0]))
    t_r_cold = bench(lambda i: read(Ws[i % L]))
    t_r_warm = bench(lambda i: read(Ws[0]))
    t_rg_same = bench(lambda i: (read(Ws[i % L]), gemm(Ws[i % L])))
    t_rg_other = bench(lambda i: (read(Ws[(i + 1) % L]), gemm(Ws[i % L])))
    t_rg_prev = bench(lambda i: (read(Ws[(i + 1) % L]), gemm(Ws[i % L])) if False else (gemm(Ws[i % L]), read(Ws[(i + 1) % L])))
    print(f"M={M} N={N} K={K} ({N * K / 2**20:.0f} MiB x {L}): GEMM cold {t_g_cold:.1f} warm {t_g_warm:.1f} | read cold {t_r_cold:.1f} warm {t_r_warm:.1f} | "
          f"read(i) + GEMM(i) {t_rg_same:.1f} (GEMM ~ {t_rg_same - t_r_cold:.1f}) | read(i+1) + GEMM(i) {t_rg_other:.1f} (GEMM ~ {t_rg_other - t_r_cold:.1f}) | GEMM(i) then read(i+1): {t_rg_prev:.1f}", flush=True)

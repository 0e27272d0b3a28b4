"""round 6: ONE structural change to the 256 x 256 epilogue, measured: the outlier operand whose LDS region is the slice buffer the
last K slice does not use is copied UNDER that slice (knob 1491) instead of behind the loop (1490, the library).  Bit check, then interleaved timing."""
import ctypes, os, sys, torch
os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
O = 128
def operands(M, N, K):
    qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
    sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
    fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
    fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
    return qA, W, sA, sW, fpA, fpW
def run(ops, out, M, N, K):
    qA, W, sA, sW, fpA, fpW = ops
    assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, st) == 0
# ---- bits: even and odd slice counts (fpW / fpA early), ragged tiles, a partial last slice
for (M, N, K) in ((8192, 4096, 4096), (8192, 4096, 4224), (5000, 4000, 4160), (8192, 4096, 4208), (8192, 2048, 128), (8192, 2048, 256)):
    ops = operands(M, N, K)
    ref, got = torch.empty((M, N), dtype=torch.float16, device=dev), torch.empty((M, N), dtype=torch.float16, device=dev)
    lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(2); lib.mixq_debug_set_gemm_variant(1490); run(ops, ref, M, N, K)
    k0 = lib.mixq_debug_last_gemm_kernel()
    lib.mixq_debug_set_gemm_variant(1491)
    ok = True
    for _ in range(5):
        got.zero_(); run(ops, got, M, N, K); torch.cuda.synchronize(); ok &= torch.equal(ref, got)
    print(f"bits {M} x {N} x {K}: {'identical' if ok else 'DIFFERENT'}   [{k0.decode().split(' ')[0]}]", flush=True)
# ---- time: the bench's three shapes at its chunk size, interleaved
for (M, N, K) in ((65536, 12288, 4096), (65536, 11008, 4096), (65536, 4096, 11008)):
    ops = operands(M, N, K)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    res = {1490: [], 1491: []}
    lib.mixq_debug_reset()
    for rnd in range(4):
        for knob in (1490, 1491):
            lib.mixq_debug_set_gemm_variant(knob)
            for _ in range(3): run(ops, out, M, N, K)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): run(ops, out, M, N, K)
            e1.record(); torch.cuda.synchronize()
            res[knob].append(e0.elapsed_time(e1) / 40 * 1e3)
    a, b = sum(res[1490]) / 4, sum(res[1491]) / 4
    print(f"time {M} x {N} x {K}: behind the loop {a:.1f} us ({' '.join(f'{x:.0f}' for x in res[1490])}) | under the last slice {b:.1f} us ({' '.join(f'{x:.0f}' for x in res[1491])})  {100 * (b / a - 1):+.2f} %", flush=True)
lib.mixq_debug_reset()

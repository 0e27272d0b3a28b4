#!/usr/bin/env python3
"""Randomised shape sweep (GPU): for many (M, N, K, O) the default kernel selection must give int32 sums equal to the
integer matrix product and fused outputs bit-identical to a fixed reference configuration (variant 13 = 64x128 tiles of
the two-barrier kernel, one workgroup per tile); the default selection includes the K splits over workgroups (the mixlib
wrappers bring their per-stream scratch).  Not part of pytest (minutes, not seconds): python tools/stress_shapes.py [count] [seed]"""
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from mixq_tensorrt_llm_amd import _lib, mixlib

count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
lib = _lib.load()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(seed)
bad = 0
t0 = time.time()
for it in range(count):
    M = int(rng.choice([rng.integers(5, 70), rng.integers(60, 300), rng.integers(250, 1100), rng.integers(1000, 4200)]))
    N = int(rng.integers(1, 300)) * 16
    K = int(rng.integers(1, 260)) * 16 if rng.random() < 0.6 else int(rng.integers(500, 1400)) * 16   # long K: the K splits
    if M * K > (1 << 26):
        M = max(5, (1 << 26) // K)
    O = int(rng.choice([0, 8, 64, 128]))
    if O > K:
        O = 0
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    lib.mixq_debug_set_gemm_variant(0)
    acc = mixlib.gemm(a, b, M, N, K)
    want = (a.double() @ b.double().T).to(torch.int32) if M * N * K < 4e9 else None
    ok_int = want is None or torch.equal(acc, want)
    # fused operator: quantise + extract + GEMM, default selection vs a fixed configuration
    A = (torch.randn((M, K), device=dev, generator=g) * 2).half()
    sW = (torch.rand(N, device=dev, generator=g) * 1e-3 + 1e-4).half()
    fpw = (torch.randn((N, max(O, 8)), device=dev, generator=g) * 0.02).half()[:, :O].contiguous() if O else \
        torch.zeros((N, 0), dtype=torch.float16, device=dev)
    ind = torch.randperm(K, device=dev, generator=g)[:O].to(torch.int32)
    outs = []
    for v in (0, 13):
        lib.mixq_debug_set_gemm_variant(v)
        if O:
            outs.append(mixlib.mixq_linear(A.clone(), b, sW, fpw, ind))
        else:
            s = torch.empty(M, dtype=torch.float16, device=dev)
            q = mixlib.FindRowScale(A.clone(), s, M, K, 8)
            outs.append(mixlib.int8FusedDequantize(q, b, s, sW, None, M, N, K))
    ok_f = torch.equal(outs[0], outs[1])
    if not (ok_int and ok_f):
        bad += 1
        print(f"MISMATCH M={M} N={N} K={K} O={O}: int32 {'ok' if ok_int else 'BAD'} fused {'ok' if ok_f else 'BAD'}", flush=True)
lib.mixq_debug_set_gemm_variant(0)
print(f"{count} shapes, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

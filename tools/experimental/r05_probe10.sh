#!/usr/bin/env bash
# round 5: W8A16 decode kernel (fpA_intB skinny form, one token tile), weights in 256-byte runs (knob 8480 on / 8490 off): bits, cold timings
mkdir -p gpurun_out/r05p10
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r05p10/bits.txt
import ctypes, os
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
import torch, bench
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(1)
st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
bad = n = 0
for (N, K) in [(4096, 4096), (12288, 4096), (4096, 11008), (11008, 4096), (1040, 2304), (3584, 3584), (528, 8192), (18944, 3584), (272, 192)]:
    t = bench.synth_layer(N, K, dev, gen)
    for M in (1, 2, 3, 4, 5, 8, 13, 16):
        A = (torch.randn((M, K), device=dev, generator=gen)).to(torch.float16)
        outs = []
        for knob in (8490, 8480):
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
            o = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            assert lib.mixq_w8a16_gemm_forward(p(A), p(t["qweight"]), p(t["weights_scaling_factor"]), p(o), M, N, K, st0) == 0
            torch.cuda.synchronize(); outs.append(o)
        n += 1; same = torch.equal(outs[0], outs[1]) and not torch.isnan(outs[1]).any(); bad += not same
        if not same: print(f"DIFFERENT M={M} N={N} K={K}")
lib.mixq_debug_reset(); print(f"{n} cells, mismatches: {bad}")
PY
python tools/decode_cold_bench.py --shapes "12288 4096;11008 4096;4096 11008;4096 4096;18944 3584;3584 18944" --Ms 1,4 --knobs "8490;8480" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p10/w8a16_runs_cold.txt

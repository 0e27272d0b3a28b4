#!/usr/bin/env bash
# round 5: (a) skinny kernel, row-major weights in 256-byte runs through LDS (knob 884 on / 885 off): bits + cold timings;
#          (b) deep2 = register-prefetch build of the mid-M deep form (r1 / r2 / r4 / r8) against auto / e*: bits + cold timings
mkdir -p gpurun_out/r05p3
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p3/skinny_wrows_bits.txt
import ctypes, os
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
import torch, bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(1)
st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bad = 0
for (N, K) in [(4096, 4096), (12288, 4096), (4096, 11008), (1040, 2304), (528, 8192), (3584, 3584)]:
    t = bench.synth_layer(N, K, dev, gen)
    for M in (5, 8, 16, 17, 31, 32, 40, 48, 57, 64):
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        outs = []
        for knob in (885, 884):
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
            o = torch.zeros((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]); out_desc = TensorDesc.make(o.shape)
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]),
                                    (ctypes.c_void_p * 1)(o.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st0) == 0
            torch.cuda.synchronize(); outs.append((o, lib.mixq_debug_last_gemm_kernel().decode().split(" ")[0]))
            lib.mixq_destroy(h)
        same = torch.equal(outs[0][0], outs[1][0]); bad += not same
        print(f"M={M:3d} N={N} K={K} {outs[0][1]} -> {'same bits' if same else 'DIFFERENT'}")
lib.mixq_debug_reset(); print("mismatches:", bad)
PY
python tools/decode_cold_bench.py --shapes "12288 4096;11008 4096;4096 11008;4096 4096;3584 3584" --Ms 8,16,32,48,64 --knobs "885;884" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p3/skinny_wrows_cold.txt
timeout 1200 python tools/midm_cfg_sweep.py --cold --secs 0.1 --Ms 128,192,256,384,512,768,1024 \
  --shapes "12288 4096;4096 11008;4096 4096;3584 8192;1024 28672;18944 3584" \
  --only auto,e1,e2,e4,e8,r1,r2,r4,r8 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee gpurun_out/r05p3/deep2_cold.txt

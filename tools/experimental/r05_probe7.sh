#!/usr/bin/env bash
# round 5: 256-byte-run route with TWO feature tiles per workgroup (knob 896) at 17..64 rows: bits, then cold timings
mkdir -p gpurun_out/r05p7
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r05p7/bits.txt
import ctypes, os
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
import torch, bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(1)
st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bad = n = 0
for (N, K) in [(4096, 4096), (12288, 4096), (4096, 11008), (1040, 2304), (528, 8192), (3584, 3584), (5136, 1280)]:
    t = bench.synth_layer(N, K, dev, gen)
    for M in (5, 16, 17, 32, 40, 48, 57, 64):
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        outs = []
        for knobs in ((897, 885), (897, 896)):
            lib.mixq_debug_reset()
            for k in knobs: lib.mixq_debug_set_gemm_variant(k)
            o = torch.zeros((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]); out_desc = TensorDesc.make(o.shape)
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]),
                                    (ctypes.c_void_p * 1)(o.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st0) == 0
            torch.cuda.synchronize(); outs.append(o); lib.mixq_destroy(h)
        n += 1; same = torch.equal(outs[0], outs[1]); bad += not same
        if not same: print(f"DIFFERENT M={M} N={N} K={K}")
lib.mixq_debug_reset(); print(f"{n} cells, mismatches: {bad}")
PY
python tools/decode_cold_bench.py --shapes "12288 4096;11008 4096;4096 11008;8192 4096;6144 4096;18944 3584;8192 8192;5120 5120;4096 4096" --Ms 32,48,64 --knobs "0;897;897,896" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p7/nt2_cold.txt

#!/usr/bin/env bash
# round 5: the one-launch decode-batch operator (R3.14, parked on 4096 x 4096) on BIG weights: bits, then cold timings (knob 1281 on / default off)
mkdir -p gpurun_out/r05p9
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r05p9/bits.txt
import ctypes, os
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
import torch, bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(1)
st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
bad = n = 0
for (N, K) in [(4096, 4096), (12288, 4096), (4096, 11008), (1040, 2304), (3584, 3584)]:
    t = bench.synth_layer(N, K, dev, gen)
    for M in (5, 16, 17, 31, 32):
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        outs = []
        for knob in (1280, 1281, 1282):
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
            o = torch.zeros((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]); out_desc = TensorDesc.make(o.shape)
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            for rep in range(3):
                assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]),
                                        (ctypes.c_void_p * 1)(o.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st0) == 0
            torch.cuda.synchronize(); outs.append((o, lib.mixq_debug_last_gemm_kernel().decode().split(" ")[0])); lib.mixq_destroy(h)
        n += 1; same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0]); bad += not same
        if not same or M == 32: print(f"M={M} N={N} K={K} {outs[0][1]} / {outs[1][1]}: {'same' if same else 'DIFFERENT'}")
lib.mixq_debug_reset(); print(f"{n} cells, mismatches: {bad}")
PY
python tools/decode_cold_bench.py --shapes "12288 4096;11008 4096;4096 11008;4096 4096;8192 8192" --Ms 8,16,32 --knobs "0;1281" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p9/fusedq_cold.txt

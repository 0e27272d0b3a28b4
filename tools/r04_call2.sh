#!/bin/bash
# round 4, GPU call 2: steady-state refit data for the "solo rounds + split tail" rule, and the quantiser weight-prefetch experiment
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/splitk_select_sweep.py --hybrid --secs 0.35 > gpurun_out/r04_hybrid_sweep.txt 2>&1
python tools/decode_cold_bench.py --shapes "4096 4096;12288 4096;11008 4096;8192 8192" --Ms 8,32,48 --knobs "0;887;888" > gpurun_out/r04_quant_prefetch.txt 2>&1
for k in 886 887; do
  echo "### knob $k" >> gpurun_out/r04_quant_prefetch_timeline.txt
  python tools/small_m_timeline.py --M 32 --N 4096 --K 4096 --knobs $k >> gpurun_out/r04_quant_prefetch_timeline.txt 2>&1
done
python - >> gpurun_out/r04_quant_prefetch.txt 2>&1 <<'PY'
# bit-identity of the operator with the prefetch blocks in the launch
import ctypes, torch, sys
sys.path.insert(0, ".")
import bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(0)
for N, K in ((4096, 4096), (12288, 4096), (5120, 5120)):
    t = bench.synth_layer(N, K, dev, gen)
    for M in (5, 17, 32, 48, 64):
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
        in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]); 
        ptrs = (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins])
        h = ctypes.c_void_p(lib.mixq_create(M, N, K))
        ws = torch.empty(max(lib.mixq_workspace_size(h, 64, N, K), 16), dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        outs = []
        for knob in (886, 887, 888):
            lib.mixq_debug_set_gemm_variant(knob)
            o = torch.zeros((M, N), dtype=torch.float16, device=dev)
            od = TensorDesc.make(o.shape); op = (ctypes.c_void_p * 1)(o.data_ptr())
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(od), ptrs, op, ctypes.c_void_p(ws.data_ptr()), st) == 0
            torch.cuda.synchronize(); outs.append(o)
        print(f"identity M={M} N={N} K={K}:", torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
        lib.mixq_destroy(h)
lib.mixq_debug_reset()
PY
tail -3 gpurun_out/r04_hybrid_sweep.txt; cat gpurun_out/r04_quant_prefetch.txt | tail -30

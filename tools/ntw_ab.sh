#!/bin/bash
# A/B on one box: plain vs non-temporal weight loads in the decode-size kernels (ab/plain.so, ab/ntw.so built with -DMIXQ_NT_WEIGHT_LOADS):
# decode_step (cold by construction) and the small_m warm / cold points of bench.py
cd "$(dirname "$0")/.."
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do
  for which in plain ntw; do
    cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
    python - "$which" "$round" <<'PY' 2>&1 | grep -v amdgpu
import sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from mixq_tensorrt_llm_amd import _lib, parallel
from mixq_tensorrt_llm_amd._lib import TensorDesc
which, rnd = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0"); lib = _lib.load(); gen = torch.Generator(device=dev).manual_seed(0)
model = bench.Model(lib, TensorDesc, parallel, dev, gen, 64, 1, 0)
r = bench.decode_step_points(lib, TensorDesc, model, dev, gen)
print(which, "r" + rnd, "decode_step us:", {k: round(v["us_per_step"], 1) for k, v in r.items() if isinstance(v, dict)})
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
s = bench.small_m_points(lib, TensorDesc, dev, gen, st)
print(which, "r" + rnd, "small_m warm/cold us:", {k: (round(v["us_per_call"], 2), round(v.get("cold", {}).get("us_per_call", 0), 2)) for k, v in s.items() if isinstance(v, dict) and "us_per_call" in v})
PY
  done
done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

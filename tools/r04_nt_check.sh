python -m pytest tests/test_gpu_w8a16_gemm.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, ctypes, torch
sys.path.insert(0, ".")
import bench
from mixq_tensorrt_llm_amd import _lib, parallel
from mixq_tensorrt_llm_amd._lib import TensorDesc
dev = torch.device("cuda:0"); lib = _lib.load(); gen = torch.Generator(device=dev).manual_seed(0)
model = bench.Model(lib, TensorDesc, parallel, dev, gen, 64, 1, 0)
for knob in (848, 849, 848):
    lib.mixq_debug_set_gemm_variant(knob)
    r = bench.decode_step_points(lib, TensorDesc, model, dev, gen)
    print(knob, "decode_step us:", {k: round(v["us_per_step"], 1) for k, v in r.items() if isinstance(v, dict)})
PY

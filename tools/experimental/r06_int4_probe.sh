cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_int4.py -q -x 2>&1 | tail -5
timeout 600 python - <<'PY'
import os, sys, ctypes
os.environ["MIXQ_DEBUG_KNOBS"]="1"
sys.path.insert(0, "/root/repo")
import torch, bench
from mixq_tensorrt_llm_amd import _lib, parallel
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib=_lib.load(); dev=torch.device("cuda:0"); gen=torch.Generator(device=dev).manual_seed(1)
model=bench.Model(lib, TensorDesc, parallel, dev, gen, 64, 1, 0)
ds=bench.decode_step_points(lib, TensorDesc, model, dev, gen, batches=(1,2,4))
r=bench.int4_points(lib, model, dev, gen, ds, batches=(1,2,4))
import json
for k,v in r.items():
    if isinstance(v,dict): print(k, {a:(round(b["us_per_step"],1) if isinstance(b,dict) and "us_per_step" in b else b) for a,b in v.items()})
PY

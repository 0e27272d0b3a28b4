"""round 6: `decode_step` (bench.py: the 96 linears of Llama-2-7B, own weights, one HIP graph) at bs 128 / 256 / 512 on ONE box: the library as it is | with the deep plan launching
round 5's form everywhere (knob 1421) -- what the mid kernel is worth end to end, free of box-to-box spread.  Two rounds, interleaved."""
import os, sys, json
os.environ["MIXQ_DEBUG_KNOBS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, bench
from mixq_tensorrt_llm_amd import _lib, parallel
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(1)
model = bench.Model(lib, TensorDesc, parallel, dev, gen, 64, 1, 0)
res = {"round6": [], "round5_form": []}
for rnd in range(2):
    for name, knobs in (("round6", []), ("round5_form", [1421])):
        lib.mixq_debug_reset()
        for k in knobs: lib.mixq_debug_set_gemm_variant(k)
        d = bench.decode_step_points(lib, TensorDesc, model, dev, gen, batches=(128, 256, 512))
        res[name].append({k: round(v["us_per_step"], 1) for k, v in d.items() if isinstance(v, dict)})
        print(name, rnd, res[name][-1], {k: v["kernels"] for k, v in d.items() if isinstance(v, dict)} if rnd == 0 else "", flush=True)
lib.mixq_debug_reset()
for bs in ("bs128", "bs256", "bs512"):
    a = sum(r[bs] for r in res["round6"]) / 2; b = sum(r[bs] for r in res["round5_form"]) / 2
    print(f"{bs}: round-5 form {b:.0f} us -> library {a:.0f} us ({100 * (a / b - 1):+.1f} %)")

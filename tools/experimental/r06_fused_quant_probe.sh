#!/usr/bin/env bash
# round 6: quantiser inside the mid-M GEMM launch: tests, then decode_step (initialised handles vs not) from a short bench run
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_quant.py -x -q 2>&1 | tail -15
timeout 900 python bench.py --tokens 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-small-m --no-sweeps 2>gpurun_out/r06_fq_bench.err | tail -1 > gpurun_out/r06_fq_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_fq_bench.json"))
for k, v in d["decode_step"].items():
    if isinstance(v, dict):
        print(k, round(v["us_per_step"]), "us", "| not initialised:", round(v.get("handles_not_initialised", {}).get("us_per_step", 0)), "|", v.get("kernels"))
PY

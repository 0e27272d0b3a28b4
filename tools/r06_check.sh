#!/usr/bin/env bash
# round 6 mid-round check: the whole GPU suite + the default bench line (decode_step at the reference's batch sizes is in it)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_check; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 --durations=8 -x 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== bench.py (defaults)"; T0=$SECONDS; timeout 1500 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench_full.json | cut -c1-600; echo "bench wall $((SECONDS - T0)) s"

#!/bin/bash
# round 4, GPU call 3: GPU tests on the re-fitted hybrid rule, the bench line with chunk_sweep + configs, the partial-round table
# after the re-fit, and steady-state data for the one-partial-wave split rules
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_c3; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== bench (short)"; timeout 900 python bench.py --steps 5 --warmup 2 2>$OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_c3/bench.json"))
print(json.dumps(d.get("chunk_sweep"), indent=1)[:3000])
print(json.dumps(d.get("configs"), indent=1)[:4000])
PY
echo "== round table"; timeout 900 python tools/round_table.py --secs 0.4 --out $OUT/round_table.txt > $OUT/round_table.log 2>&1; tail -40 $OUT/round_table.txt
echo "== single-wave split rules, steady state"; timeout 1200 python tools/splitk_select_sweep.py --secs 0.3 > $OUT/splitk_steady.txt 2>&1; tail -5 $OUT/splitk_steady.txt

#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_c8; mkdir -p $OUT
export TMPDIR=/tmp
echo "== selection check"; timeout 1200 python tools/selection_check.py > $OUT/selection_check.txt 2>&1; tail -12 $OUT/selection_check.txt
echo "== splitk soak (hybrid rule re-fitted)"; timeout 900 python tools/splitk_soak.py --shapes 300 --seed 4 > $OUT/splitk_soak.txt 2>&1; tail -4 $OUT/splitk_soak.txt
echo "== cpu baseline leg"; python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.cpu_baseline())[:600])
PY

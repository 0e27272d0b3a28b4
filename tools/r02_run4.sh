#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run4; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest fusedq"; timeout 900 python -m pytest tests/test_gpu_fusedq.py -q -x --timeout 300 2>&1 | tail -25 | tee $OUT/pytest.log
echo "== small-M timing: one launch vs two"
for v in 81 80; do for shape in "32 4096 4096" "16 4096 4096" "8 4096 4096" "16 12288 4096" "32 4096 1024"; do set -- $shape
 echo -n "variant $v M=$1 N=$2 K=$3: "; timeout 120 python tools/enqueue_bench.py --M $1 --N $2 --K $3 --variant $v --iters 2000 2>&1 | tail -1; done; done | tee $OUT/small_m.txt

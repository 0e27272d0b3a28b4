#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run11; mkdir -p $OUT
for v in 89 88 89 88; do for shape in "65536 4096" "65536 11008" "16384 4096"; do set -- $shape
  echo -n "nt=$v M=$1 K=$2: "; timeout 100 python tools/gemm_bench.py --M $1 --N 256 --K $2 --variant $v --iters 300 --what quant 2>&1 | tail -1; done; done | tee $OUT/quant_nt.txt
echo "== bench A/B (3 steps)"
for v in 89 88; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --variant $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant', $v, d['value'], d['roofline']['avg_launch_ms'], d['quantizer']['achieved'])"; done | tee -a $OUT/quant_nt.txt

#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run12; mkdir -p $OUT
for rep in 1 2; do for v in 0 132; do for shape in "8192 12288 4096" "8192 4096 11008"; do set -- $shape
  echo -n "rep $rep variant $v: "; timeout 200 python tools/gemm_bench.py --M $1 --N $2 --K $3 --variant $v --iters 3000 --what gemm 2>&1 | tail -1; done; done; done | tee $OUT/swap.txt

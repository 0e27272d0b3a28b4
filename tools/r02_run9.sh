#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run9; mkdir -p $OUT
echo "== skinny kernel ablations, GEMM only (gemm_bench --what gemm), 4096x4096"
for M in 8 16 32; do for v in 0 92 93 94 95 98; do
  echo -n "M=$M ablation $v: "; timeout 100 python tools/gemm_bench.py --M $M --N 4096 --K 4096 --variant $v --iters 3000 --what gemm 2>&1 | tail -1; done; done | tee $OUT/skinny_ablate.txt
echo "== quant only"; for M in 8 32; do timeout 100 python tools/gemm_bench.py --M $M --N 4096 --K 4096 --iters 3000 --what quant 2>&1 | tail -1; done | tee -a $OUT/skinny_ablate.txt

#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run7; mkdir -p $OUT
for v2 in 91 90; do echo "== variant 79 + $v2"; timeout 120 python tools/gemm_bench.py --M 1024 --N 4096 --K 11008 --variant 79 --variant2 $v2 --stamps --iters 300 --what gemm 2>&1 | tail -12; done | tee $OUT/stamps.txt
echo "== S=2 shape"; timeout 120 python tools/gemm_bench.py --M 512 --N 12288 --K 4096 --variant 79 --stamps --iters 300 --what gemm 2>&1 | tail -12 | tee -a $OUT/stamps.txt

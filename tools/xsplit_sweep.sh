#!/bin/bash
# small-tile kernels, K split over workgroups as well (gemm_kernels.hip, XS): v60 = off, v62 / v64 = forced, per shape
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-"4096,11008" "4096,4096" "4096,16384" "1024,28672" "2048,8192" "5120,13824"}; do set -- ${shape//,/ }
for M in ${MS:-32 48 64 96 128 192 256}; do
  line="N=$1 K=$2 M=$M:"
  for v in 60 62 64; do
    t=$(timeout 100 python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 500 --what gemm ${CHECK:-} 2>&1 | grep -E "gemm |bit-id" | sed -E 's/.*: ([0-9.]+) us.*/\1/; s/bit-identical to the plain launch \(20 rounds\): (True|False).*/[\1]/' | tr '\n' ' ')
    line="$line v$v=$t"
  done
  echo "$line"
done; done

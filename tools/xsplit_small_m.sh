#!/bin/bash
# M <= 32: default selection with the K split over workgroups off (v60: skinny kernel) vs automatic (v69) vs forced (v64)
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-"4096,11008" "1024,28672" "2048,8192" "4096,16384" "1024,8192" "4096,4096"}; do set -- ${shape//,/ }
for M in ${MS:-8 16 24 32}; do
  line="N=$1 K=$2 M=$M:"
  for v in ${VS:-60 69 62 64 68 66}; do
    t=$(timeout 100 python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 500 --what gemm --check 2>&1 | grep -E "gemm |bit-id" | sed -E 's/.*: ([0-9.]+) us.*/\1/; s/bit-identical to the plain launch \(20 rounds\): (True|False).*/[\1]/' | tr '\n' ' ')
    line="$line v$v=$t"
  done
  echo "$line"
done; done

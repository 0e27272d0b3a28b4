#!/bin/bash
# K split over workgroups (gemm_pp_kernels.hip, SPLITK): time + bit-identity against the plain launch, per shape / factor
cd "$(dirname "$0")/.."
for shape in ${SHAPES:-"1024 4096 11008" "2048 4096 11008" "1024 11008 4096" "512 12288 4096" "1024 12288 4096" "1024 4096 4096" "2048 4096 4096" "512 4096 11008" "4096 4096 11008"}; do
  set -- $shape
  for v in 70 72 74; do
    echo "== M=$1 N=$2 K=$3 variant=$v"
    timeout 120 python tools/gemm_bench.py --M $1 --N $2 --K $3 --variant $v --iters 200 --what gemm $( [ $v != 70 ] && echo --check ) 2>&1 | grep -E "bit-identical|gemm|rror|fault" | head -4
  done
done

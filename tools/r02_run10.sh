#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run10; mkdir -p $OUT
echo "== parity (skinny paths)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_mixlinear.py -q --timeout 600 2>&1 | tail -4
echo "== GEMM only"
for shape in "8 4096 4096" "16 4096 4096" "32 4096 4096" "16 12288 4096" "16 11008 4096" "32 4096 2048" "16 3584 3584"; do set -- $shape
  echo -n "M=$1 N=$2 K=$3: "; timeout 100 python tools/gemm_bench.py --M $1 --N $2 --K $3 --iters 3000 --what gemm 2>&1 | tail -1; done | tee $OUT/gemm_only.txt
echo "== whole operator, graph"
for shape in "32 4096 4096" "16 4096 4096" "8 4096 4096" "16 12288 4096"; do set -- $shape
 echo -n "M=$1 N=$2 K=$3: "; timeout 120 python tools/enqueue_bench.py --M $1 --N $2 --K $3 --iters 3000 --graph 100 2>&1 | tail -1; done | tee $OUT/small_m.txt

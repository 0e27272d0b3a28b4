#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run6; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest splitk + xsplit + parity"; timeout 2400 python -m pytest tests/test_gpu_splitk.py tests/test_gpu_xsplit.py tests/test_gpu_parity.py -q --maxfail=10 --timeout 900 2>&1 | tail -15 | tee $OUT/pytest.log
echo "== split shapes: us per call (default patience) vs plain"
for shape in "1024 4096 11008" "1536 4096 11008" "512 12288 4096" "1024 3584 18944" "4096 1280 8192" "1536 11008 4096" "512 4096 11008" "1024 1024 28672"; do set -- $shape
 for v in 79 70; do echo -n "M=$1 N=$2 K=$3 variant $v: "; timeout 120 python tools/gemm_bench.py --M $1 --N $2 --K $3 --variant $v --iters 1500 --what gemm 2>&1 | tail -1; done; done | tee $OUT/split_shapes.txt
echo "== small M operator (quantiser block-per-row)"
for shape in "32 4096 4096" "16 4096 4096" "8 4096 4096" "32 12288 4096" "64 4096 11008"; do set -- $shape
 echo -n "M=$1 N=$2 K=$3: "; timeout 120 python tools/enqueue_bench.py --M $1 --N $2 --K $3 --iters 3000 --graph 100 2>&1 | tail -1; done | tee $OUT/small_m.txt

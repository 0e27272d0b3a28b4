#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run3; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest w8a16 gemm"; timeout 1500 python -m pytest tests/test_gpu_w8a16_gemm.py -q --maxfail=10 --timeout 900 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== timing 12288x4096"; timeout 600 python tools/w8a16_bench.py --N 12288 --K 4096 --Ms 5,32,64,128,256,512,1024 2>&1 | grep w8a16 | tee $OUT/w8a16_12288x4096.txt
echo "== timing 3584x18944"; timeout 600 python tools/w8a16_bench.py --N 3584 --K 18944 --Ms 5,32,64,128,256,512,1024 2>&1 | grep w8a16 | tee $OUT/w8a16_3584x18944.txt

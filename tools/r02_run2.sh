#!/usr/bin/env bash
# round 2, GPU run 2: fpA_intB GEMM parity + timing, power trace (longer)
set -u
OUT=gpurun_out/r02_run2; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest w8a16 gemm"; timeout 1500 python -m pytest tests/test_gpu_w8a16_gemm.py tests/test_gpu_mixlinear.py -q --maxfail=10 --timeout 900 2>&1 | tail -25 | tee $OUT/pytest.log
echo "== timing 12288x4096"; timeout 600 python tools/w8a16_bench.py --N 12288 --K 4096 --vendor 2>&1 | grep w8a16 | tee $OUT/w8a16_12288x4096.txt
echo "== timing 12288x4096 no scratch"; timeout 600 python tools/w8a16_bench.py --N 12288 --K 4096 --no-scratch --Ms 5,32,128,512 2>&1 | grep w8a16 | tee $OUT/w8a16_12288x4096_noscratch.txt
echo "== timing 3584x18944"; timeout 600 python tools/w8a16_bench.py --N 3584 --K 18944 --vendor 2>&1 | grep w8a16 | tee $OUT/w8a16_3584x18944.txt
echo "== timing 4096x11008"; timeout 600 python tools/w8a16_bench.py --N 4096 --K 11008 2>&1 | grep w8a16 | tee $OUT/w8a16_4096x11008.txt
echo "== power trace"; bash tools/power_trace.sh $OUT/power_trace_qkv.txt --M 8192 --N 12288 --K 4096 | tail -20

#!/usr/bin/env bash
# round 6: the mid kernel builds (h barrier pairs | i flag rings 5 + 4 | j 6 + 3 | k, l the same with the prefetch wave) against the deep form (e) and
# the automatic selection, cold and warm; first their parity tests
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deep.py -x -q -k "3 or epilogues" > gpurun_out/r06_mid_tests.txt 2>&1; tail -3 gpurun_out/r06_mid_tests.txt
for mode in --cold ""; do
  timeout 900 python tools/midm_cfg_sweep.py $mode --secs 0.1 --Ms 96,128,192,256,384,512 --shapes "12288 4096;11008 4096;4096 11008" \
     --only auto,e1,e2,e4,h1,h2,h4,i1,i2,i4,j1,j2,j4,k1,k2,k4,l1,l2,l4,pp128 > gpurun_out/r06_mid_sweep${mode:+_cold}.txt 2>&1
  cat gpurun_out/r06_mid_sweep${mode:+_cold}.txt | grep -v amdgpu.ids
done

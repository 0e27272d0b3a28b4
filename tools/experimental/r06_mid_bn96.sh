#!/usr/bin/env bash
# round 6: 96-wide tiles of the mid kernel (h1, by rule) against 128-wide ones (w1) and the round-5 form (e1): parity tests first, then cold / warm
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deep.py -x -q > gpurun_out/r06_mid_tests.txt 2>&1; tail -3 gpurun_out/r06_mid_tests.txt
for mode in --cold ""; do
  timeout 900 python tools/midm_cfg_sweep.py $mode --secs 0.1 --Ms 160,192,256 --shapes "12288 4096;11008 4096" --only auto,r5deep,e1,h1,w1,j1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_mid_bn96${mode:+_cold}.txt
  timeout 900 python tools/midm_cfg_sweep.py $mode --secs 0.1 --Ms 512,768 --shapes "4608 3584;4096 4096" --only auto,r5deep,e1,h1,w1,j1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_mid_bn96${mode:+_cold}.txt
  cat gpurun_out/r06_mid_bn96${mode:+_cold}.txt
done
timeout 300 python tools/experimental/r06_mid_timeline.py --knobs 1272 2>&1 | grep -v amdgpu > gpurun_out/r06_mid_timeline_bn96.txt; cat gpurun_out/r06_mid_timeline_bn96.txt

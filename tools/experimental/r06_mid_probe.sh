#!/usr/bin/env bash
# round 6: the mid kernel builds (h one barrier per pair of slices | i ping-pong phases) against the deep form (e) and
# the automatic selection, cold and warm; first their parity tests
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_deep.py -x -q -k "30 or 31 or epilogues" > gpurun_out/r06_mid_tests.txt 2>&1; tail -3 gpurun_out/r06_mid_tests.txt
for mode in --cold ""; do
  timeout 900 python tools/midm_cfg_sweep.py $mode --secs 0.1 --Ms 96,128,192,256,384,512 --shapes "12288 4096;11008 4096;4096 11008" \
     --only auto,e1,e2,e4,h1,h2,h4,j1,j2,j4,pp128,s2,s4,s8 > gpurun_out/r06_mid_sweep${mode:+_cold}.txt 2>&1
  cat gpurun_out/r06_mid_sweep${mode:+_cold}.txt | grep -v amdgpu.ids
done
timeout 300 python tools/experimental/r06_mid_timeline.py --knobs 1272 > gpurun_out/r06_mid_timeline_final.txt 2>&1; grep -v amdgpu gpurun_out/r06_mid_timeline_final.txt

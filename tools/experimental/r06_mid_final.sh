#!/usr/bin/env bash
# round 6: the shipped mid kernel (h = K walk rotated per tile row, j = not) against the round-5 deep form (e) and the selection, cold, on every
# BASELINE (N, K) in the row ranges where deep_plan_auto applies; then warm on the Llama-2-7B shapes
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
V=auto,nodeep,e1,e2,e4,e8,h1,h2,h4,h8,j1,j2,j4,j8
O=gpurun_out/r06_mid_final_sweep_cold.txt; : > $O
run() { timeout 600 python tools/midm_cfg_sweep.py --cold --secs 0.1 --Ms "$1" --shapes "$2" --only $V 2>&1 | grep -v amdgpu.ids >> $O; }
run 160,192,256 "12288 4096;11008 4096"
run 100,128 "12288 4096;11008 4096;18944 3584"
run 160,256,384,512 "4096 11008;3584 18944"
run 320,384,512 "3584 8192"
run 512,768,1024 "4608 3584;4096 4096;1280 8192"
run 320,384,512 "1024 28672"
cat $O
timeout 600 python tools/midm_cfg_sweep.py --secs 0.1 --Ms 192,256,384,512 --shapes "12288 4096;11008 4096;4096 11008" --only $V 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_mid_final_sweep_warm.txt
cat gpurun_out/r06_mid_final_sweep_warm.txt

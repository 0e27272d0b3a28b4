#!/usr/bin/env bash
# round 6: the rotated K walk of the mid kernel by the number of tile rows (unsplit cells): h1 = rotated, j1 = not, auto = the selection
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for mode in --cold ""; do
  O=gpurun_out/r06_mid_rot_rows${mode:+_cold}.txt; : > $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 384,512,640,768 --shapes '4608 3584' --only auto,r5deep,h1,j1 2>&1 | grep -v amdgpu.ids >> $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 512,640,768,896,1024 --shapes '4096 4096' --only auto,r5deep,h1,j1 2>&1 | grep -v amdgpu.ids >> $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 192,256 --shapes '12288 4096;11008 4096' --only auto,r5deep,h1,j1 2>&1 | grep -v amdgpu.ids >> $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 512,768,1024 --shapes '1280 8192;3584 8192' --only auto,r5deep,h2,j2,h4,j4 2>&1 | grep -v amdgpu.ids >> $O
  echo "== $mode"; cat $O
done

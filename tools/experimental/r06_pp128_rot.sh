#!/usr/bin/env bash
# round 6: rotated K walk in the 128 x 256 ping-pong tiles (knob 1481): auto = not rotated, r128 = rotated wherever those tiles are selected
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for mode in --cold ""; do
  O=gpurun_out/r06_pp128_rot${mode:+_cold}.txt; : > $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 320,384,448,512 --shapes '12288 4096;11008 4096' --only auto,r128 2>&1 | grep -v amdgpu.ids >> $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 256,384,512,768,1024 --shapes '18944 3584' --only auto,r128 2>&1 | grep -v amdgpu.ids >> $O
  timeout 600 python tools/midm_cfg_sweep.py $mode --secs 0.15 --Ms 1024 --shapes '4608 3584;4096 4096' --only auto,r128 2>&1 | grep -v amdgpu.ids >> $O
  echo "== $mode"; cat $O
done

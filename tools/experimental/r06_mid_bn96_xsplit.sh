#!/usr/bin/env bash
# round 6, R6.23: 96-wide tiles of the mid kernel WITH K split over workgroups.  FIRST run (one box, library without them): x2 / x4 = forced 2 / 4 parts on 96-wide tiles where workgroups <= CUs (knob 1433 then
# meant "also with K split"), j2 / j4 = 128-wide.  From the SECOND run on the library takes them by rule: auto / j = 96-wide where it fits, x = only without K split (knob 1433 now = the rule before), r5deep = round 5's form.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_deep.py tests/test_gpu_selection.py::test_described_plan_is_what_enqueue_launches -x -q 2>&1 | tail -3
O=gpurun_out/r06_mid_bn96_xsplit.txt; : > $O
timeout 600 python tools/midm_cfg_sweep.py --cold --secs 0.15 --Ms 320,384,512 --shapes '3584 8192' --only auto,r5deep,j2,x2 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python tools/midm_cfg_sweep.py --cold --secs 0.15 --Ms 512,768,1024 --shapes '1280 8192' --only auto,r5deep,j2,x2,j4,x4 2>&1 | grep -v amdgpu.ids >> $O
timeout 600 python tools/midm_cfg_sweep.py --secs 0.15 --Ms 384,1024 --shapes '3584 8192;1280 8192' --only auto,r5deep,j2,x2 2>&1 | grep -v amdgpu.ids >> $O
cat $O

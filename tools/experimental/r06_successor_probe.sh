#!/usr/bin/env bash
# round 6: trailing workgroups of the mid kernel that pull the NEXT call's weight towards the Infinity Cache (mixq_weight_successor_register):
# cold rotation, every copy hints its successor; p{x} = exactly 8 x trailing workgroups (p0: none), auto = by rule (the free CUs, at most 64)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r06_successor_probe.txt; : > $O
timeout 900 python tools/midm_cfg_sweep.py --cold --hint --secs 0.1 --Ms 192,256 --shapes '12288 4096;11008 4096' --only auto,r5deep,p0,p2,p4,p6,p8 2>&1 | grep -v amdgpu.ids >> $O
timeout 900 python tools/midm_cfg_sweep.py --cold --hint --secs 0.1 --Ms 256,512 --shapes '4096 11008;4096 4096' --only auto,r5deep,p0,p2,p4,p6,p8 2>&1 | grep -v amdgpu.ids >> $O
timeout 900 python tools/midm_cfg_sweep.py --cold --hint --secs 0.1 --Ms 512,1024 --shapes '4608 3584;1280 8192;3584 8192' --only auto,r5deep,p0,p2,p4,p6,p8 2>&1 | grep -v amdgpu.ids >> $O
echo "# warm (no rotation: nothing to hint) for reference" >> $O
timeout 900 python tools/midm_cfg_sweep.py --secs 0.1 --Ms 192,256 --shapes '12288 4096;11008 4096' --only auto,r5deep 2>&1 | grep -v amdgpu.ids >> $O
cat $O

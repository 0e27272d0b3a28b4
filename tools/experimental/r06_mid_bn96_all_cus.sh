#!/usr/bin/env bash
# round 6: 96-wide tiles of the mid kernel on EVERY CU, K walk not rotated (R6.7 (2) had measured them rotated).  FIRST run (profiles/r06_mid_bn96_all_cus.txt): auto / n1 = 128-wide, m1 = 96-wide up to one workgroup per CU;
# from the second run on the rule is the library's: auto = 96-wide up to one workgroup per CU, m1 = only with a sixteenth of the CUs free (the earlier rule), n1 = 128-wide
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
O=gpurun_out/r06_mid_bn96_all_cus.txt; : > $O
for rep in 1 2; do
timeout 600 python tools/midm_cfg_sweep.py --cold --secs 0.2 --Ms 160,192,224,256 --shapes '12288 4096;11008 4096' --only auto,n1,m1,r5deep 2>&1 | grep -v amdgpu.ids >> $O
done
timeout 600 python tools/midm_cfg_sweep.py --secs 0.2 --Ms 192,256 --shapes '12288 4096' --only auto,n1,m1,r5deep 2>&1 | grep -v amdgpu.ids >> $O
cat $O
timeout 900 python tools/experimental/r06_decode_step_ab.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r06_decode_step_ab2.txt

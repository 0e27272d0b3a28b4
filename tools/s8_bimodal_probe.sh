for i in 1 2 3; do timeout 200 python tools/midm_cfg_sweep.py --cold --secs 0.3 --only auto,s4,s8 --Ms 384,512 --shapes "4096 11008;4096 16384" 2>&1 | grep -v amdgpu; done
timeout 300 python tools/midm_cfg_sweep.py --cold --secs 0.3 --only auto,c17,c19,c21,pp128,s4,s8 --Ms 384,512 --shapes "4096 11008" 2>&1 | grep -v amdgpu
timeout 300 python tools/midm_cfg_sweep.py --secs 0.3 --only auto,s4,s8 --Ms 384,512 --shapes "4096 11008" 2>&1 | grep -v amdgpu
rocm-smi --showclocks 2>/dev/null | head -20

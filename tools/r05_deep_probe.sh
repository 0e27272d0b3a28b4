#!/usr/bin/env bash
# round 5: validation of the automatic table of the mid-M deep form: new automatic selection vs the selection with the form off vs the forced forms, cold
mkdir -p gpurun_out/r05f
(timeout 900 python -m pytest tests/test_gpu_deep.py -x -q 2>&1 | tail -8) > gpurun_out/r05f/deep_tests.log 2>&1
tail -4 gpurun_out/r05f/deep_tests.log
timeout 1100 python tools/midm_cfg_sweep.py --cold --secs 0.12 --Ms 72,112,136,144,176,224,256,288,352,448,512,576,640,896,1024 \
  --shapes "12288 4096;11008 4096;4096 11008;4096 4096;4608 3584;18944 3584;3584 18944;1280 8192;3584 8192;1024 28672;5120 5120;8192 8192;6144 4096;2048 16384" \
  --only auto,nodeep,e1,e2,e4,e8 > gpurun_out/r05f/deep_validate_cold.txt 2>&1
grep -c MISMATCH gpurun_out/r05f/deep_validate_cold.txt
awk '{print}' gpurun_out/r05f/deep_validate_cold.txt | cut -c1-170 | tail -215

#!/usr/bin/env bash
# round 5: deep form with late copy issue on half of the waves (e) vs all waves issuing first (g); cold
mkdir -p gpurun_out/r05i
timeout 900 python tools/midm_cfg_sweep.py --cold --secs 0.12 --Ms 128,192,256,384,512,768 \
  --shapes "12288 4096;4096 11008;4096 4096;3584 18944;3584 8192;1024 28672" \
  --only auto,nodeep,e1,e2,e4,e8,g1,g2,g4,g8 > gpurun_out/r05i/deep_late_ab.txt 2>&1
cut -c1-190 gpurun_out/r05i/deep_late_ab.txt
(timeout 600 python -m pytest tests/test_gpu_deep.py tests/test_gpu_splitk.py tests/test_gpu_weight_image.py -x -q 2>&1 | tail -4)

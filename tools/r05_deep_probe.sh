#!/usr/bin/env bash
# round 5: deep form builds: e = 8 waves 128x128, h = 16 waves 128x128, w = 128x256 tiles 8 waves 3 stages; cold
mkdir -p gpurun_out/r05k
timeout 900 python tools/midm_cfg_sweep.py --cold --secs 0.12 --Ms 128,192,256,384,512,768,1024 \
  --shapes "12288 4096;4096 11008;4096 4096;3584 8192;1024 28672;18944 3584" \
  --only auto,pp128,e1,e2,e4,e8,h1,h2,h4,h8,w1,w2,w4 > gpurun_out/r05k/deep_builds.txt 2>&1
cut -c1-200 gpurun_out/r05k/deep_builds.txt

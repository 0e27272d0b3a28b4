#!/bin/bash
# fpA_intB wide form: weights through LDS (846) against weights through registers (847), per forced configuration, one box
for shape in "12288 4096" "4096 4096" "28672 8192" "4096 11008" "3584 18944"; do
  set -- $shape
  for cfg in 832 835 836; do
    python tools/w8a16_bench.py --N $1 --K $2 --Ms ${MS:-64,128,256,512} --iters 200 --sweep "$cfg,846;$cfg,847" 2>&1 | grep sweep | sed "s/^/cfg $cfg /"
  done
done

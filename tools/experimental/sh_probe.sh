#!/bin/bash
# fpA_intB wide form: K halves with register weights (835/836 + 847) against stage halves (837/838), same K split, one box
for shape in "12288 4096" "4096 4096" "28672 8192" "4096 11008" "3584 18944"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms ${MS:-64,128,256,512} --iters 200 --sweep "80;835,847;837;836,847;838" 2>&1 | grep sweep
done

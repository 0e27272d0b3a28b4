#!/bin/bash
# fpA_intB GEMM plan check: automatic plan (80) next to narrow passes (81) and the wide form with 128- / 256-row tiles (82 / 84)
# x K split automatic (85) / 1 / 2 / 4 / 8 (86..89); $1 = "full" for the whole grid
SW="81;82,86;84,86;82,85;84,85;80"
[ "$1" = full ] && SW="81;82,86;82,87;82,88;82,89;84,86;84,87;84,88;84,89;80"
for shape in "12288 4096" "4096 11008" "3584 18944" "4096 4096" "11008 4096" "18944 3584" "1280 8192" "28672 8192"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 257,384,512,768,1024,1536,2048,3072,4096 --iters 60 --sweep "$SW" 2>&1 | grep sweep
done

#!/bin/bash
# fpA_intB GEMM plan sweep: narrow passes (81) and the configurations of the wide form (831 / 832 / 833 / 834: 32 / 64 / 128 / 256-row
# tiles, 835 / 836: the 64- / 128-row tiles as K halves; K split automatic), automatic plan (80) last
SW="${SW:-81;831;832;835;833;836;834;80}"
for shape in ${SHAPES:-"12288 4096" "4096 11008" "3584 18944" "4096 4096" "1280 8192" "28672 8192"}; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms ${MS:-5,16,32,64,128,256,384,512,768,1024,2048,4096} --iters 60 --sweep "$SW" 2>&1 | grep sweep
done

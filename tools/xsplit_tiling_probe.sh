#!/bin/bash
# small-tile K split: tallest tile allowed 32 (knob 63) vs 64 (knob 65, default); for M >= 256 also against the 256x256 split
cd "$(dirname "$0")/.."
run() { timeout 100 python tools/gemm_bench.py --M $1 --N $2 --K $3 --variant $4 --variant2 $5 ${6:+--variant3 $6} --iters 500 --what gemm --check 2>&1 | grep -E "gemm |bit-id" | sed -E 's/.*: ([0-9.]+) us.*/\1/; s/bit-identical to the plain launch \(20 rounds\): (True|False).*/[\1]/' | tr '\n' ' '; }
for s in ${SHAPES:-"96,4096,11008" "128,4096,11008" "192,4096,11008" "256,4096,11008" "384,4096,11008" "128,4096,16384" "256,4096,16384" "192,1024,28672" "256,1024,28672" "512,1024,28672" "128,2048,16384" "256,2048,16384" "128,3584,18944" "256,3584,18944" "128,8192,8192"}; do
  IFS=, read -r m n k <<< "$s"
  echo "M=$m N=$n K=$k: bm<=32 $(run $m $n $k 63 70 69) | bm<=64 $(run $m $n $k 65 70 69) | default (256x256 split allowed) $(run $m $n $k 65 79)"
done

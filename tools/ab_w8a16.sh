#!/bin/bash
# A/B of two builds (ab/old.so, ab/new.so) of the fpA_intB GEMM on ONE box, interleaved; usage: [MS=..] [SW=..] bash tools/ab_w8a16.sh "N K" ...
cd "$(dirname "$0")/.."
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do
  for which in old new; do
    cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
    for s in "$@"; do
      read -r n k <<< "$s"
      echo -n "$which r$round: N=$n K=$k "; timeout 200 python tools/w8a16_bench.py --N $n --K $k --Ms ${MS:-64,96,128,192,256} --iters 200 --sweep "${SW:-80}" 2>&1 | grep sweep | sed -E 's/sweep N=[0-9]+ K=[0-9]+ //' | tr '\n' '|'; echo
    done
  done
done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

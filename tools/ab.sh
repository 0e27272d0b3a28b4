#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is +-4 %): ab/old.so vs ab/new.so, interleaved.
# usage: bash tools/ab.sh "8192 12288 4096" "8192 4096 11008" ...
cd "$(dirname "$0")/.."
SHAPES=("$@")
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do
  for which in old new; do
    cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
    for s in "${SHAPES[@]}"; do
      read -r m n k <<< "$s"
      echo -n "$which r$round: "; timeout 200 python tools/gemm_bench.py --M $m --N $n --K $k --iters ${ITERS:-2000} --what gemm ${EXTRA:-} 2>&1 | tail -1
    done
  done
done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

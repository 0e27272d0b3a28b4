#!/bin/bash
# three builds on one box, interleaved: ab/old.so ab/new.so ab/new2.so; usage: bash tools/experimental/ab3.sh "M N K" ...
cd "$(dirname "$0")/../.."
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do for which in old new new2; do cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
  for s in "$@"; do read -r m n k <<< "$s"; echo -n "$which r$round: "; timeout 200 python tools/gemm_bench.py --M $m --N $n --K $k --iters ${ITERS:-40} --what gemm 2>&1 | tail -1; done; done; done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

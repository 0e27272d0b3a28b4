#!/bin/bash
# two builds (ab/old.so, ab/new.so) of the decode-batch operator through mixq_enqueue, HIP graph of 100 calls, one box
cd "$(dirname "$0")/../.."
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do for which in old new; do cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
  for s in "$@"; do read -r m n k <<< "$s"; echo -n "$which r$round: M=$m N=$n K=$k "; timeout 200 python tools/enqueue_bench.py --M $m --N $n --K $k --graph 100 2>&1 | tail -1; done; done; done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

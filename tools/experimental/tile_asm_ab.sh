#!/bin/bash
# A/B of the two-barrier tile kernel with builtin vs asm-form LDS-DMA copies (ab/old.so, ab/new.so), decode batches to mid M
SH=()
for nk in "12288 4096" "11008 4096" "4096 11008" "4096 4096" "28672 8192" "3584 18944"; do
  for m in 48 64 128 256 512; do SH+=("$m $nk"); done
done
ITERS=${ITERS:-500} bash tools/ab.sh "${SH[@]}"

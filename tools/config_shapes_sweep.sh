#!/bin/bash
# Large-M rate of the fused GEMM on every (N, K) of the BASELINE configurations (Llama-2-7B, Qwen2-7B, Llama-2-70B unsharded
# and its TP = 8 column shards), one chunk of M tokens: us per call, TOPS, fraction of the nominal int8 peak.
cd "$(dirname "$0")/.."
M=${M:-16384}
for s in "12288 4096" "11008 4096" "4096 11008" "4096 4096" \
         "4608 3584" "18944 3584" "3584 18944" \
         "10240 8192" "28672 8192" "8192 28672" \
         "1280 8192" "3584 8192" "1024 28672"; do
  set -- $s
  timeout 300 python tools/gemm_bench.py --M $M --N $1 --K $2 --iters ${ITERS:-30} --what both 2>&1 | tail -2 | tr '\n' ' '; echo
done

#!/usr/bin/env bash
# 128x256 ping-pong kernel (variant 5) vs the default selection (variant 0) over mid-size shapes: us per call + tile counts
cd "$(dirname "$0")/.."
for nk in "4096 4096" "12288 4096" "11008 4096" "4096 11008" "3584 3584" "4608 3584" "5120 5120" "8192 8192" "1280 8192" "18944 3584"; do set -- $nk
  for M in 256 384 512 768 1024 1536 2048 3072 4096; do
    t0=$(timeout 100 python tools/gemm_bench.py --M $M --N $1 --K $2 --variant 0 --iters 500 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    t5=$(timeout 100 python tools/gemm_bench.py --M $M --N $1 --K $2 --variant 5 --iters 500 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    t128=$(( ((M+127)/128) * (($1+255)/256) )); t256=$(( ((M+255)/256) * (($1+255)/256) ))
    echo "M=$M N=$1 K=$2 tiles128=$t128 tiles256=$t256 default=$t0 pp128=$t5"
  done
done

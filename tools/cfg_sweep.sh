# us per call of gemm_w8a8o16_kernel tile configurations (variant 10 + i) vs the default selection (v0)
CFGS="${CFGS:-0 26 28 30}"
for shape in "12288 4096" "4096 11008" "4096 4096"; do set -- $shape
for M in ${MS:-8 16 24 32 48 64 128 256}; do
  line="N=$1 K=$2 M=$M:"
  for v in $CFGS; do
    t=$(python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 300 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    line="$line v$v=$t"
  done
  echo "$line"
done; done

for shape in "12288 4096" "4096 11008" "4096 4096" "11008 4096"; do set -- $shape
for M in 8 16 32 48 64; do
  line="N=$1 K=$2 M=$M:"
  for v in 0 44 48 56 17; do
    t=$(python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 300 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    line="$line v$v=$t"
  done
  echo "$line"
done; done

for shape in "12288 4096" "4096 11008" "4096 4096"; do set -- $shape
for M in 32 64 128 256 512 1024; do
  line="N=$1 K=$2 M=$M:"
  for v in 0 10 13 15 17 19 20 21 22 23 24 25; do
    t=$(python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 300 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    line="$line v$v=$t"
  done
  echo "$line"
done; done

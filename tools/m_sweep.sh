# default kernel selection over M for the three Llama-2-7B shapes + 4096x4096 (us per call, steady state)
for shape in "12288 4096" "11008 4096" "4096 11008" "4096 4096"; do set -- $shape
  line="N=$1 K=$2:"
  for M in 8 32 64 128 256 512 1024 2048; do
    t=$(python tools/gemm_bench.py --M $M --N $1 --K $2 --variant 0 --iters 300 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
    line="$line M$M=$t"
  done
  echo "$line"
done

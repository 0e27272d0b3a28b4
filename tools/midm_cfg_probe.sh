cd /root/repo
for s in "512 4096 4096" "768 4096 4096" "1024 4096 4096" "1536 4096 4096" "2048 4096 4096" "1024 5120 5120" "1024 8192 8192"; do
  read -r m n k <<< "$s"; line="M=$m N=$n K=$k:"
  for v in ${VS:-70 14 24 16}; do
    t=$(timeout 100 python tools/gemm_bench.py --M $m --N $n --K $k --variant $v --iters 500 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/'); line="$line v$v=$t"
  done; echo "$line"
done

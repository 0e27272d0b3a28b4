cd /root/repo
for s in "64 4096 11008" "48 4096 11008" "64 4096 16384" "192 1024 28672" "128 1024 28672" "64 2048 8192" "96 2048 16384" "128 2048 16384" "64 3584 18944" "40 8192 8192" "64 8192 8192"; do
  read -r m n k <<< "$s"; line="M=$m N=$n K=$k:"
  for pref in 63 61; do
    t=$(timeout 100 python tools/gemm_bench.py --M $m --N $n --K $k --variant $pref --variant2 69 --iters 500 --what gemm --check 2>&1 | grep -E "gemm |bit-id" | sed -E 's/.*: ([0-9.]+) us.*/\1/; s/bit-identical to the plain launch \(20 rounds\): (True|False).*/[\1]/' | tr '\n' ' ')
    line="$line pref$pref=$t"
  done; echo "$line"
done

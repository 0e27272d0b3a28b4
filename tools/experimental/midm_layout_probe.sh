# timing probe (wrong results): mid-M kernels reading A as K-slice-major (knob 898) vs row-major -- GEMM only, K splits over workgroups off (70)
cd "$(dirname "$0")/../.."
for s in "64 12288 4096" "128 12288 4096" "256 12288 4096" "128 4096 4096" "128 4096 11008" "512 12288 4096" "96 11008 4096"; do read -r m n k <<< "$s"; for v in 899 898; do echo -n "M=$m N=$n K=$k knob $v: "; python tools/gemm_bench.py --M $m --N $n --K $k --variant $v --variant2 70 --iters 3000 --what gemm 2>&1 | tail -1; done; done

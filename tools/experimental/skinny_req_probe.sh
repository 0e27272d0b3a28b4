cd "$(dirname "$0")/../.."
for m in 32 16; do for v in 0 96; do echo -n "M=$m v=$v: "; python tools/gemm_bench.py --M $m --N 4096 --K 4096 --variant $v --iters 3000 --what gemm 2>&1 | tail -1; done; done

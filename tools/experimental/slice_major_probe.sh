# timing probe: the ping-pong GEMM reading A (qA) as K-slice-major [K/128][M][128 B] (ablation 512, wrong results) vs row-major
cd "$(dirname "$0")/../.."
for r in 1 2; do for s in "8192 12288 4096" "8192 4096 11008" "8192 11008 4096"; do read -r m n k <<< "$s"; for v in 0 612; do echo -n "run $r v=$v: "; python tools/gemm_bench.py --M $m --N $n --K $k --variant $v --iters 2000 --what gemm 2>&1 | tail -1; done; done; done

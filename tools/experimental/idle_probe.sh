# Is the large-M GEMM time-bound or energy-bound?  Add ~4k / ~16k idle cycles per 256 x 256 tile (s_sleep at the start of the
# epilogue; results unchanged) and see by how much the launch grows.  Tile time ~87k cycles at K = 4096.
cd "$(dirname "$0")/../.."
for r in 1 2; do for v in 0 228 356; do echo -n "run $r variant $v: "; python tools/gemm_bench.py --M 8192 --N 12288 --K 4096 --variant $v --iters 3000 --what gemm 2>&1 | tail -1; done; done
for v in 0 228 356; do echo -n "K=11008 variant $v: "; python tools/gemm_bench.py --M 8192 --N 4096 --K 11008 --variant $v --iters 3000 --what gemm 2>&1 | tail -1; done

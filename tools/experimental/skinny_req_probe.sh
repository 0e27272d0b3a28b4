cd "$(dirname "$0")/../.."
python -m pytest tests/test_gpu_parity.py -q -x -k "int32 or schedule or fused_dequant or enqueue" 2>&1 | tail -3
for s in "32 4096 4096" "16 4096 4096" "8 4096 4096" "32 4096 11008" "16 12288 4096" "32 3584 3584" "24 4096 5120" "48 4096 4096" "64 4096 4096"; do read -r m n k <<< "$s"; echo -n "M=$m N=$n K=$k: "; python tools/gemm_bench.py --M $m --N $n --K $k --variant 40 --iters 3000 --what gemm 2>&1 | tail -1; done
python tools/small_m_timeline.py --M 32 2>&1 | grep -v amdgpu

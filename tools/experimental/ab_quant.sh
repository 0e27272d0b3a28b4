cd /root/repo
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do for which in old new; do cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
 for s in "65536 4096" "65536 11008" "16384 4096" "8192 8192"; do read -r m k <<< "$s"; echo -n "$which r$round: "; timeout 200 python tools/gemm_bench.py --M $m --N 4096 --K $k --iters 60 --what quant 2>&1 | tail -1; done; done; done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

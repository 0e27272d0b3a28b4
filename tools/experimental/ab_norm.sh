cd "$(dirname "$0")/../.."
cp mixq_tensorrt_llm_amd/libmixq_mi355x.so /tmp/keep.so
for round in 1 2; do for which in old new; do cp ab/$which.so mixq_tensorrt_llm_amd/libmixq_mi355x.so
  echo "$which r$round:"; python tools/norm_bench.py --Ms 2048,16384,65536 2>&1 | grep "M="; done; done
cp /tmp/keep.so mixq_tensorrt_llm_amd/libmixq_mi355x.so

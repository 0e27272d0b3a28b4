#!/bin/bash
# quick check of the fpA_intB GEMM: tests, then the automatic plan (80) and the two wide tile heights without K split
timeout 600 python -m pytest tests/test_gpu_w8a16_gemm.py -q -x --timeout 600 2>&1 | tail -3
for shape in "12288 4096" "4096 4096" "3584 18944"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms ${MS:-512,1024,2048,4096} --iters 60 --sweep "${SW:-82,86;84,86;80}" 2>&1 | grep sweep
done

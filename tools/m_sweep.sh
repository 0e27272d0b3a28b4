#!/bin/bash
# default kernel selection over M for the three Llama-2-7B shapes + 4096x4096 (us per call, steady state):
# "now" = default build with scratch (all K splits over workgroups on), "v70" = the same with both splits off,
# "vendor" = torch._int_mm (plain s8 GEMM, no epilogue)
cd "$(dirname "$0")/.."
for shape in "12288 4096" "11008 4096" "4096 11008" "4096 4096"; do set -- $shape
  for v in 0 70; do
    line="N=$1 K=$2 variant=$v:"
    for M in 8 32 64 128 256 512 1024 2048; do
      t=$(timeout 100 python tools/gemm_bench.py --M $M --N $1 --K $2 --variant $v --iters 300 --what gemm 2>&1 | tail -1 | sed -E 's/.*: ([0-9.]+) us.*/\1/')
      line="$line M$M=$t"
    done
    echo "$line"
  done
done
python - <<'PY'
import torch
for N, K in ((12288, 4096), (11008, 4096), (4096, 11008), (4096, 4096)):
    W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda").t()
    line = f"N={N} K={K} vendor:"
    for M in (32, 64, 128, 256, 512, 1024, 2048):
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
        for _ in range(5):
            torch._int_mm(a, W)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            torch._int_mm(a, W)
        e1.record(); torch.cuda.synchronize()
        line += f" M{M}={e0.elapsed_time(e1) / 300 * 1e3:.1f}"
    print(line)
PY

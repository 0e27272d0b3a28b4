#!/usr/bin/env bash
# round 2, GPU run 1: full GPU test-suite on the batch-A changes, new bench line, 2-rank control-flow check, power trace
set -u
OUT=gpurun_out/r02_run1; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 --timeout 900 2>&1 | tail -30 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
echo "== bench (3 steps)"; timeout 900 python bench.py --steps 3 --warmup 1 2>$OUT/bench.err | tail -1 | tee $OUT/bench.json
tail -3 $OUT/bench.err
echo "== 2 ranks on one GPU (control flow of the tp object)"
MIXQ_BENCH_SINGLE_GPU_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 1 --warmup 0 --tokens 16384 --chunk 8192 --tp-steps 1 --no-cpu-baseline 2>$OUT/bench2.err | tail -1 | tee $OUT/bench2.json
tail -3 $OUT/bench2.err
echo "== power trace"; ITERS=20000 bash tools/power_trace.sh $OUT/power_trace_qkv.txt --M 8192 --N 12288 --K 4096 | tail -16

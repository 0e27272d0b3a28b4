#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run8; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest tp"; timeout 900 python -m pytest tests/test_gpu_tp.py -q --timeout 600 2>&1 | tail -12 | tee $OUT/pytest_tp.log
echo "== 2 ranks on one GPU: bench control flow with the peer transport"
MIXQ_BENCH_SINGLE_GPU_RANKS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 1 --warmup 0 --tokens 16384 --chunk 8192 --tp-steps 2 --no-cpu-baseline 2>$OUT/bench2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['tp']))" | tee $OUT/bench2_tp.json
tail -3 $OUT/bench2.err
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --timeout 900 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== pmc w8a16"; bash tools/pmc_w8a16.sh $OUT/pmc_w8a16 2>&1 | tail -12

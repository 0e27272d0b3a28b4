#!/bin/bash
# round 4, GPU call 4: full GPU suite (8-rank TP tests, element-wise bounds everywhere), next-layer weight hint experiment
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_c4; mkdir -p $OUT
export TMPDIR=/tmp
echo "== decode hint"; timeout 900 python tools/decode_hint_bench.py > $OUT/decode_hint.txt 2>&1; grep -v amdgpu $OUT/decode_hint.txt | tail -25
echo "== pytest -m gpu (tp first)"; timeout 1500 python -m pytest tests/test_gpu_tp.py -m gpu -q -x --timeout 1200 2>&1 | tail -15 | tee $OUT/pytest_tp.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_tp.py 2>&1 | tail -8 | tee $OUT/pytest_gpu.log

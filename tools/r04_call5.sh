#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_c5; mkdir -p $OUT
export TMPDIR=/tmp
echo "== hint tests"; timeout 900 python -m pytest tests/test_gpu_hint.py -m gpu -q -x --timeout 600 2>&1 | tail -6 | tee $OUT/pytest_hint.log
echo "== decode hint, whole-line mode"; MIXQ_HINT_MODE=2 timeout 900 python tools/decode_hint_bench.py --fracs 0.125,0.5,1.0 > $OUT/decode_hint_mode2.txt 2>&1; grep -v amdgpu $OUT/decode_hint_mode2.txt | tail -12
echo "== eager decode step"; timeout 900 python tools/eager_decode_step.py --bs 32,8 > $OUT/eager_decode_step.txt 2>&1; grep -v amdgpu $OUT/eager_decode_step.txt | tail -4

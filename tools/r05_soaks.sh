#!/usr/bin/env bash
# round 5, final library: every soak tool once (random shapes, bit-exact comparisons), one summary line each
mkdir -p gpurun_out/r05_soaks
( timeout 900 python tools/stress_shapes.py 300 5 2>&1 | tail -2
  timeout 900 python tools/splitk_soak.py --shapes 300 2>&1 | tail -2
  timeout 900 python tools/wimg_soak.py --iters 500 2>&1 | tail -2
  timeout 600 python tools/skinny_rows_soak.py --n 400 2>&1 | tail -1
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "capturable" 2>&1 | tail -1 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_soaks/summary.txt

#!/usr/bin/env bash
mkdir -p gpurun_out/r05p5
timeout 1500 python -m pytest tests/test_gpu_skinny_rows.py tests/test_gpu_int4.py -q -x -s 2>&1 | grep -v "^  File\|Extension modules\|amdgpu.ids" | tail -8 | tee gpurun_out/r05p5/pytest.txt
python tools/int4_stream_bench.py --Ms 1,8,32,64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p5/int4_stream_bench.txt

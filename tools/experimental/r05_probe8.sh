#!/usr/bin/env bash
mkdir -p gpurun_out/r05p8
timeout 1500 python -m pytest tests/test_gpu_skinny_rows.py tests/test_gpu_parity.py tests/test_gpu_weight_image.py tests/test_gpu_mixlinear.py tests/test_gpu_selection.py tests/test_gpu_norm.py -q -x 2>&1 | tail -4 | tee gpurun_out/r05p8/pytest.txt
python tools/skinny_rows_soak.py --n 300 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r05p8/soak.txt
python tools/decode_cold_bench.py --shapes "5120 5120;6144 4096;7168 4096;8192 4096;8192 8192;5120 13824;4608 3584;8192 1024" --Ms 8,16,32,48,64 --knobs "0;895;896" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p8/nt2_rule_cold.txt
timeout 900 python tools/selection_check.py > gpurun_out/r05p8/selection_check.txt 2>&1; tail -4 gpurun_out/r05p8/selection_check.txt

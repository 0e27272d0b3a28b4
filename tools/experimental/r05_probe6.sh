#!/usr/bin/env bash
mkdir -p gpurun_out/r05p6
timeout 1500 python -m pytest tests/test_gpu_deep.py tests/test_gpu_selection.py tests/test_gpu_parity.py tests/test_gpu_weight_image.py tests/test_gpu_skinny_rows.py tests/test_gpu_mixlinear.py -q -x 2>&1 | tail -4 | tee gpurun_out/r05p6/pytest.txt
timeout 900 python tools/selection_check.py > gpurun_out/r05p6/selection_check.txt 2>&1; tail -6 gpurun_out/r05p6/selection_check.txt
bash tools/skinny_rows_pmc.sh 2>&1 | tee gpurun_out/r05p6/skinny_rows_pmc.txt

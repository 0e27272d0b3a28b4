#!/usr/bin/env bash
# round 5: after the 256-byte-run route of the skinny kernel: (a) parity tests that reach it, (b) the 33..64-row skinny / tiles crossover again
mkdir -p gpurun_out/r05p4
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_weight_image.py tests/test_gpu_mixlinear.py tests/test_gpu_selection.py -q -x --timeout 1200 2>&1 | tail -6 | tee gpurun_out/r05p4/pytest.txt
python tools/decode_cold_bench.py --shapes "12288 4096;11008 4096;8192 4096;6144 4096;5120 5120;4608 3584;18944 3584;3584 8192;1280 8192;8192 8192;4096 11008" --Ms 40,48,64 --knobs "0;897;885" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p4/rows_33_64.txt
python tools/decode_cold_bench.py --shapes "12288 4096;18944 3584;28672 8192" --Ms 24,32 --knobs "0;892;885" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05p4/rows_33_64.txt

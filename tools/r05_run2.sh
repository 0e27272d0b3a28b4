#!/usr/bin/env bash
# round 5: whole GPU suite (every failure listed), the selection gate's output
mkdir -p gpurun_out/r05h
(timeout 3000 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -60) > gpurun_out/r05h/gpu_tests.log 2>&1
tail -8 gpurun_out/r05h/gpu_tests.log
timeout 900 python tools/selection_check.py > gpurun_out/r05h/selection_check.txt 2>&1
tail -12 gpurun_out/r05h/selection_check.txt

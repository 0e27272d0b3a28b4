#!/usr/bin/env bash
# Samples power / clocks (rocm-smi) while the GEMM runs back-to-back.  usage: bash tools/power_probe.sh [variant] [extra args]
V="${1:-2}"; shift || true
python tools/gemm_bench.py --variant "$V" --what gemm --iters 30000 "$@" > gpurun_out/power_gemm.log 2>&1 &
PID=$!
sleep 8
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';' ; echo
  sleep 0.4
done
wait $PID
cat gpurun_out/power_gemm.log | grep gemm

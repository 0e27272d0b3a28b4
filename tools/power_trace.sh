#!/usr/bin/env bash
# Power / clock trace of the GPU while the fused GEMM runs back-to-back (evidence for the "power-capped, not issue-bound"
# reading of DESIGN.md 2.3).  One line per sample: t, package power, power cap, sclk, mclk, fclk, junction temperature.
# usage: bash tools/power_trace.sh OUTFILE [gemm_bench args...]        (from the repo root, on the GPU box)
OUT="${1:-gpurun_out/power_trace.txt}"; shift || true
mkdir -p "$(dirname "$OUT")"
{
  echo "# $(date -u +%FT%TZ)  $(rocm-smi --showproductname 2>/dev/null | grep -m1 -E 'Card [Ss]eries|Card model' | sed 's/  */ /g')"
  echo "# power cap: $(rocm-smi --showmaxpower 2>/dev/null | grep -m1 -i 'max' | sed 's/  */ /g')"
  echo "# idle sample:"
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | sed 's/  */ /g' | tr '\n' ';'; echo
  echo "# load: python tools/gemm_bench.py --what gemm --iters ${ITERS:-60000} $*"
} > "$OUT"
python tools/gemm_bench.py --what gemm --iters "${ITERS:-60000}" "$@" > "$OUT.gemm.log" 2>&1 &
PID=$!
sleep 6   # import + clock ramp
T0=$(date +%s%N)
for i in $(seq 1 "${SAMPLES:-16}"); do
  NOW=$(date +%s%N)
  printf "t=%d.%ds " $(( (NOW - T0) / 1000000000 )) $(( (NOW - T0) / 100000000 % 10 )) >> "$OUT"
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|junction" | sed 's/  */ /g' | tr '\n' ';' >> "$OUT"
  echo >> "$OUT"
  kill -0 $PID 2>/dev/null || break
  sleep 0.5
done
wait $PID
echo "# result: $(grep gemm "$OUT.gemm.log" | tail -1)" >> "$OUT"
rm -f "$OUT.gemm.log"
cat "$OUT"

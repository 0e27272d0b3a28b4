#!/usr/bin/env bash
# One gpurun call: GPU parity tests, smoke, a short bench, and a rocprofv3 kernel-trace summary.
# Usage (from the repo root on the GPU box): bash tools/gpu_check.sh [quick|full]
set -u
MODE="${1:-quick}"
OUT=gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== nproc $(nproc)"
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 2>&1 | tail -40 | tee "$OUT/pytest_gpu.log"
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== bench"
if [ "$MODE" = "full" ]; then
  timeout 1200 python bench.py 2>&1 | tail -3 | tee "$OUT/bench.log"
else
  timeout 600 python bench.py --tokens 131072 --steps 1 --warmup 1 2>&1 | tail -3 | tee "$OUT/bench.log"
fi
echo "== mid-M leg (short prefill, K split over workgroups)"
timeout 600 python bench.py --tokens 65536 --steps 1 --warmup 1 --no-cpu-baseline --mid-m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['mid_m']))" | tee "$OUT/mid_m.json"
echo "== bench A/B (v1 two-barrier kernel)"
timeout 600 python bench.py --tokens 131072 --steps 1 --warmup 1 --variant 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v1', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
echo "== rocprofv3 kernel trace"
rm -rf "$OUT/prof" ; mkdir -p "$OUT/prof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o mixq -- \
    python "$OLDPWD/bench.py" --tokens 65536 --steps 1 --warmup 1 --no-cpu-baseline ) > "$OUT/rocprof.log" 2>&1
tail -3 "$OUT/rocprof.log"
find "$OUT/prof" -name "*kernel_stats*" | head -3
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do head -6 "$f" | cut -c1-220; done

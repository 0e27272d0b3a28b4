#!/usr/bin/env bash
# round 5 quick evidence: smoke, default bench line, rocprofv3 kernel stats of the same command family (no PMC, no test suite).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_bench_prof; mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench.py (defaults)"; T0=$SECONDS; timeout 1500 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench_full.json | cut -c1-600; echo "bench wall $((SECONDS - T0)) s"
echo "== rocprofv3 kernel trace + stats"
rm -rf "$OUT/prof"; mkdir -p "$OUT/prof"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o mixq -- \
    python "$OLDPWD/bench.py" --tokens 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-decode-step --no-sweeps ) > "$OUT/rocprof.log" 2>&1
tail -2 "$OUT/rocprof.log"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -6 "$f" | cut -c1-200; done
rm -rf "$OUT/prof"

#!/usr/bin/env bash
# round 3 "final numbers" run: default bench, rocprofv3 kernel stats of the same command family, PMC traffic, mid-M leg
set -u
OUT=gpurun_out/r03_final; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== bench.py (defaults: 30 steps, 10 warm-up)"; timeout 1500 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench_full.json | cut -c1-600
echo "== mid-M leg"; timeout 600 python bench.py --tokens 65536 --steps 1 --warmup 1 --no-cpu-baseline --mid-m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['mid_m']))" | tee $OUT/mid_m.json | cut -c1-400
echo "== rocprofv3 kernel trace + stats"
rm -rf "$OUT/prof"; mkdir -p "$OUT/prof"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o mixq -- \
    python "$OLDPWD/bench.py" --tokens 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-decode-step ) > "$OUT/rocprof.log" 2>&1
tail -2 "$OUT/rocprof.log"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -6 "$f" | cut -c1-200; done
rm -rf "$OUT/prof"   # (raw traces: tens of MiB; gpurun copies back at most 64 MiB)
echo "== PMC traffic"; bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_bench/summary.txt $OUT/pmc_traffic_summary.txt; cat $OUT/pmc_traffic_summary.txt
rm -rf gpurun_out/pmc_bench/FETCH_SIZE gpurun_out/pmc_bench/WRITE_SIZE

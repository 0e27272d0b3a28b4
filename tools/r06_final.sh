#!/usr/bin/env bash
# round 6 "final numbers" run: GPU suite, smoke, default bench (the line the driver will reproduce), rocprofv3 kernel stats of the
# same command family, PMC traffic.  Outputs under gpurun_out/r06_final (small: gpurun copies back 64 MiB at most).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_final; mkdir -p $OUT
export TMPDIR=/tmp
echo "== PMC traffic (first: bench.py below reports roofline.traffic from the file this re-stamps)"
bash tools/pmc_bench.sh > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_bench/summary.txt $OUT/pmc_traffic_summary.txt; grep pp_kernel $OUT/pmc_traffic_summary.txt | head -2
python tools/pmc_traffic_update.py --note "${PMC_NOTE:-tools/r06_final.sh}" && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf gpurun_out/pmc_bench/FETCH_SIZE gpurun_out/pmc_bench/WRITE_SIZE
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q --timeout 1500 --durations=8 2>&1 | tail -18 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== bench.py (defaults)"; T0=$SECONDS; timeout 1500 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench_full.json | cut -c1-500; echo "bench wall $((SECONDS - T0)) s"
echo "== rocprofv3 kernel trace + stats"
rm -rf "$OUT/prof"; mkdir -p "$OUT/prof"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o mixq -- \
    python "$OLDPWD/bench.py" --tokens 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-decode-step --no-sweeps ) > "$OUT/rocprof.log" 2>&1
tail -2 "$OUT/rocprof.log"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats.csv; head -6 "$f" | cut -c1-200; done
rm -rf "$OUT/prof"

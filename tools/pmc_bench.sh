#!/usr/bin/env bash
# HBM traffic of the bench's kernels from PMC counters (separate rocprofv3 passes, kernel-trace only).
# Writes gpurun_out/pmc_bench/summary.txt: per-kernel mean FETCH_SIZE / WRITE_SIZE (KiB, raw) per dispatch.
set -u
OUT="$PWD/gpurun_out/pmc_bench"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2; do   # (a counter pass now and then dies at start-up with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: once more)
    rm -rf "$OUT/$c"
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o p -- \
        python "$OLDPWD/bench.py" --tokens 65536 --steps 1 --warmup 0 --no-cpu-baseline --no-decode-step --no-sweeps --no-small-m ) > "$OUT/$c.log" 2>&1
    find "$OUT/$c" -name "*counter_collection.csv" 2>/dev/null | grep -q . && break
  done
done
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
d = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        by_grid = collections.defaultdict(list)   # the three GEMM shapes differ in grid size: per-shape means
        for r in csv.DictReader(open(f)):
            if "mixq::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0]
                agg[k].append(float(r["Counter_Value"]))
                if "gemm_w8a8o16_pp_kernel" in k:
                    by_grid[(k, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{c} {k} mean_per_dispatch_KiB {sum(v)/len(v):.1f} n {len(v)}")
        for (k, g), v in sorted(by_grid.items()):
            print(f"{c} {k} grid {g} mean_per_dispatch_KiB {sum(v)/len(v):.1f} n {len(v)}")
PY

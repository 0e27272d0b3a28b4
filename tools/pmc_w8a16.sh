#!/usr/bin/env bash
# "Weight bytes read once": FETCH_SIZE of the fpA_intB GEMM per launch vs the N*K weight bytes.
# usage: bash tools/pmc_w8a16.sh OUTDIR     (separate rocprofv3 pass, kernel-trace only; gfx950: FETCH_SIZE counts 64 B per
# 128-B request of a wide coalesced stream -> doubled before comparing, MI355X_MICROARCH.md "HBM")
set -u
OUT="$PWD/${1:-gpurun_out/pmc_w8a16}"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for shape in "12288 4096" "3584 18944"; do set -- $shape
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/f_$1x$2" -o p -- \
      python "$OLDPWD/tools/w8a16_bench.py" --N $1 --K $2 --Ms 5,32,128,256,512 --iters 3 ) > "$OUT/f_$1x$2.log" 2>&1
  python - "$OUT/f_$1x$2" $1 $2 <<'PY'
import csv, glob, sys, collections
d, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "w8a16_gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            agg[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    for (k, g), v in sorted(agg.items()):
        kib = sum(v) / len(v)
        print(f"N={N} K={K} {k} grid {g}: FETCH_SIZE {kib:.0f} KiB raw -> x2 = {2*kib/1024:.1f} MiB per launch; "
              f"weights {N*K/2**20:.1f} MiB  (n={len(v)})")
PY
done | tee "$OUT/summary.txt"

#!/usr/bin/env bash
# PMC passes for the GEMM kernel (separate rocprofv3 runs per counter group; no trace domains besides kernel-trace).
# usage: bash tools/pmc.sh <variant> [extra gemm_bench args]
set -u
V="${1:-0}"; shift || true
OUT="$PWD/gpurun_out/pmc_v$V"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # name counters...
  local name="$1"; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- \
      python "$OLDPWD/tools/gemm_bench.py" --variant "$V" --iters 3 --what gemm "${EXTRA[@]}" ) > "$OUT/$name.log" 2>&1
  python - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_w8a8o16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"  {k:32s} mean/dispatch {sum(v)/len(v):.4g}  (n={len(v)})")
PY
}
EXTRA=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT

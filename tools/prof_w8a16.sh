#!/usr/bin/env bash
# rocprofv3 kernel stats of the fpA_intB / decode forms: which kernel serves which token count, average durations
# usage: bash tools/prof_w8a16.sh OUTDIR
set -u
OUT="$PWD/${1:-gpurun_out/prof_w8a16}"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o w8a16 -- \
    python "$OLDPWD/tools/w8a16_bench.py" --N 12288 --K 4096 --Ms 1,4,8,32,64,256,1024,4096 --iters 50 ) > "$OUT/run.log" 2>&1
grep w8a16 "$OUT/run.log"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); do cp "$f" "$OUT/kernel_stats.csv"; grep -E "Name|w8a16|gemm_w8a8o16" "$f" | cut -c1-220; done

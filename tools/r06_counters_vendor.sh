#!/usr/bin/env bash
# round 6: (1) SQ counters of the mid-M families at their own sizes, the shipped mid kernel next to the round-5 deep form (VERDICT r5 #1:
# "SQ_VALU_MFMA_BUSY share at 256 rows"); (2) the vendor yardstick on all nine BASELINE (N, K), 64..1024 rows, warm and cold
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r06_tile_families_sq_pmc.txt; : > $O
for cfg in "mid 256 0" "r5deep 256 1421" "pp128 512 0" ; do
  set -- $cfg
  echo "=== $1: M=$2 N=12288 K=4096 (knob $3)" >> $O
  bash tools/pmc.sh 0 --variant2 $3 --M $2 --N 12288 --K 4096 2>&1 | grep -E "SQ_|GRBM" >> $O
done
cat $O
timeout 1500 python tools/vendor_int8_gemm.py --mid --all --secs 0.1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_vendor_mid_m_all_shapes.txt
cat gpurun_out/r06_vendor_mid_m_all_shapes.txt

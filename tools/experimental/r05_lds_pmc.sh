#!/usr/bin/env bash
# round 5: SQ counters of the three tile families at their own sizes -- how busy is the LDS pipe against the MFMA pipe (the LDS-bandwidth account)
for cfg in "deep 256" "pp128 512" "pp256 8192"; do
  set -- $cfg
  echo "=== $1: M=$2 N=12288 K=4096"
  bash tools/pmc.sh 0 --M $2 --N 12288 --K 4096 2>&1 | grep -E "SQ_|GRBM" 
done

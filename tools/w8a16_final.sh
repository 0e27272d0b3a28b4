#!/bin/bash
# fpA_intB GEMM, automatic plan, with the vendor fp16 GEMM on pre-dequantised weights beside it + the ablations of the 256-row form
for shape in "12288 4096" "3584 18944" "4096 4096" "4096 11008"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 5,32,64,128,256,512,1024,2048,4096 --iters 100 --vendor 2>&1 | grep w8a16
done
echo "# ablations of the 256-row wide form (834, K unsplit; measurement only, wrong results): full | 801 no copies in the loop | 802 no dequantisation | 804 token fragments read once | 808 no MFMAs | 807 = 1+2+4 (MFMAs + weight reads + barriers only) | 814 = 2+4+8 (copies + barriers only)"
python tools/w8a16_bench.py --N 4096 --K 4096 --Ms 512,4096 --iters 60 --sweep "834,86;834,86,801;834,86,802;834,86,804;834,86,808;834,86,807;834,86,814" 2>&1 | grep sweep

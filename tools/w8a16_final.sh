#!/bin/bash
# fpA_intB GEMM, automatic plan, with the vendor fp16 GEMM on pre-dequantised weights beside it + the ablations of the 256-row form
for shape in "12288 4096" "3584 18944" "4096 4096" "4096 11008"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 5,32,64,128,256,512,1024,2048,4096 --iters 100 --vendor 2>&1 | grep w8a16
done
echo "# ablations of the 256-row wide form (834, K unsplit; measurement only, wrong results): full | 801 no copies in the loop | 802 no dequantisation | 804 token fragments read once | 808 no MFMAs | 807 = 1+2+4 (MFMAs + weight reads + barriers only) | 814 = 2+4+8 (copies + barriers only)"
python tools/w8a16_bench.py --N 4096 --K 4096 --Ms 512,4096 --iters 60 --sweep "841,834,86;841,834,86,801;841,834,86,802;841,834,86,804;841,834,86,808;841,834,86,807;841,834,86,814" 2>&1 | grep sweep
echo "# two-pass form off (841) / forced (842) / automatic (80): us per call"
for shape in "12288 4096" "4096 4096" "3584 18944" "4096 11008" "28672 8192"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 1024,1536,2048,4096,8192 --iters 40 --sweep "841;842;80" 2>&1 | grep sweep
done

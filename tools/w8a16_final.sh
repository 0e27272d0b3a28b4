#!/bin/bash
# fpA_intB GEMM / decode, automatic plan, with the vendor fp16 GEMM on pre-dequantised weights beside it; then the form-by-form blocks
for shape in "12288 4096" "3584 18944" "4096 4096" "4096 11008"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 1,2,4,5,16,32,64,128,256,512,1024,2048,4096 --iters 100 --vendor 2>&1 | grep w8a16
done
echo "# decode batches: GEMV (857) / MFMA skinny form (856) / automatic (858); 5..32 tokens: narrow form (851) / skinny shapes 852..855 (32 x 8, 32 x 16, 64 x 8, 64 x 16 columns x waves) / automatic (80)"
for shape in "12288 4096" "4096 4096" "4096 11008" "3584 18944" "28672 8192" "1280 8192"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 1,2,3,4 --iters 300 --sweep "857;856;858" 2>&1 | grep sweep
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 5,8,16,17,24,32 --iters 200 --sweep "851;852;853;854;855;80" 2>&1 | grep sweep
done
echo "# ablations of the 256-row wide form (834, K unsplit, two-pass off; measurement only, wrong results): full | 801 no copies in the loop | 802 no dequantisation | 804 token fragments read once | 808 no MFMAs | 807 = 1+2+4 (MFMAs + weight reads + barriers only) | 814 = 2+4+8 (copies + barriers only)"
python tools/w8a16_bench.py --N 4096 --K 4096 --Ms 512,4096 --iters 60 --sweep "841,834,86;841,834,86,801;841,834,86,802;841,834,86,804;841,834,86,808;841,834,86,807;841,834,86,814" 2>&1 | grep sweep
echo "# two-pass form off (841) / forced (842) / automatic (80): us per call"
for shape in "12288 4096" "4096 4096" "3584 18944" "4096 11008" "28672 8192"; do
  set -- $shape
  python tools/w8a16_bench.py --N $1 --K $2 --Ms 1024,1536,2048,4096,8192 --iters 40 --sweep "841;842;80" 2>&1 | grep sweep
done

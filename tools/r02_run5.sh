#!/usr/bin/env bash
set -u
OUT=gpurun_out/r02_run5; mkdir -p $OUT
export TMPDIR=/tmp
echo "== small-M timing, eager vs graph, one launch (81) vs two (80)"
for v in 81 80; do for shape in "32 4096 4096" "16 4096 4096" "32 4096 1024"; do set -- $shape
 echo -n "variant $v M=$1 N=$2 K=$3 eager: "; timeout 120 python tools/enqueue_bench.py --M $1 --N $2 --K $3 --variant $v --iters 2000 2>&1 | tail -1
 echo -n "variant $v M=$1 N=$2 K=$3 graph: "; timeout 120 python tools/enqueue_bench.py --M $1 --N $2 --K $3 --variant $v --iters 4000 --graph 100 2>&1 | tail -1; done; done | tee $OUT/small_m.txt
echo "== rocprof kernel trace of both"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof81 -o t -- python $OLDPWD/tools/enqueue_bench.py --variant 81 --iters 500 ) > $OUT/prof81.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof80 -o t -- python $OLDPWD/tools/enqueue_bench.py --variant 80 --iters 500 ) > $OUT/prof80.log 2>&1
for d in prof81 prof80; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo $d; head -5 $f | cut -c1-200; done

#!/bin/bash
OUT=$PWD/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/s2 -o r --output-format csv -- python $ROOT/bench.py --streams 1 --other off --cpu-sample 0 --steps 10 > $OUT/s2.json 2> $OUT/s2.err
find $OUT/s2 -name "*kernel_stats.csv" -exec cp {} $OUT/s2_kernel_stats.csv \;
rm -rf $OUT/s2
cd $ROOT
rm -f $OUT/t7_bench.log
for hook in "" "--debug no_settle=1" "" "--debug no_settle=1"; do
  echo "== 1080p $hook" >> $OUT/t7_bench.log
  python bench.py --other off --cpu-sample 0 --steps 30 --warmup 3 $hook >> $OUT/t7_bench.log 2>&1
done
for hook in "" "--debug no_settle=1"; do
  echo "== share $hook" >> $OUT/t7_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --levels 0-8 --poses 128 $hook >> $OUT/t7_bench.log 2>&1
  echo "== config4 $hook" >> $OUT/t7_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --levels 0-8 $hook >> $OUT/t7_bench.log 2>&1
done

#!/bin/bash
OUT=$PWD/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for v in settle nosettle; do
  H=""; [ $v = nosettle ] && H="--debug no_settle=1"
  rocprofv3 --kernel-trace --stats -d $OUT/st_$v -o r --output-format csv -- python $ROOT/bench.py --streams 1 --other off --cpu-sample 0 --steps 10 $H > $OUT/st_$v.json 2> $OUT/st_$v.err
  find $OUT/st_$v -name "*kernel_stats.csv" -exec cp {} $OUT/st_${v}_kernel_stats.csv \;
  rm -rf $OUT/st_$v
done

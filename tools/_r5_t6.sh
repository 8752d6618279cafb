#!/bin/bash
OUT=$PWD/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
i=0
for ARGS in "--big" "--width 320 --height 200 --poses 8192" "--big --width 3840 --height 2160 --poses 256 --time-varying" "--levels 0-8 --poses 128"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats -d $OUT/sb_$i -o r --output-format csv -- python $ROOT/bench.py --streams 1 --other off --cpu-sample 0 --steps 6 $ARGS > $OUT/sb_$i.json 2> $OUT/sb_$i.err
  find $OUT/sb_$i -name "*kernel_stats.csv" -exec cp {} $OUT/sb_${i}_kernel_stats.csv \;
  rm -rf $OUT/sb_$i
done

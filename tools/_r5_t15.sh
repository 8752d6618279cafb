#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
{
for ARGS in "--other off" "--other off --levels 0-8 --poses 128" "--other off --streams 1 --poses 128" "--other off --streams 1" "--other off --big" "--other off --width 320 --height 200 --poses 8192" "--other off --width 3840 --height 2160 --poses 256"; do
  echo "== $ARGS"
  bash tools/ab_so.sh _variants/rowsinner.so _variants/rowsouter.so -- $ARGS
done
} > $OUT/t15_ab.log 2>&1

#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t14_bench.log
for p in 64 128 256 512 1024 2048; do
  echo "== poses $p" >> $OUT/t14_bench.log
  python bench.py --other off --cpu-sample 0 --steps 20 --warmup 3 --streams 1 --poses $p >> $OUT/t14_bench.log 2>&1
done

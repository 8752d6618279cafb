#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t8_bench.log
for hook in "" "--debug bin_threads=512"; do
  echo "== config5 $hook" >> $OUT/t8_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --big --width 3840 --height 2160 --poses 256 --time-varying --streams 1 $hook >> $OUT/t8_bench.log 2>&1
  echo "== share $hook" >> $OUT/t8_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --levels 0-8 --poses 128 $hook >> $OUT/t8_bench.log 2>&1
  echo "== big512 $hook" >> $OUT/t8_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --big --poses 512 --streams 1 $hook >> $OUT/t8_bench.log 2>&1
  echo "== 1080p-512 $hook" >> $OUT/t8_bench.log
  python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 --poses 512 --streams 1 $hook >> $OUT/t8_bench.log 2>&1
done

#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t13_bench.log
run() { echo "== $1" >> $OUT/t13_bench.log; shift; python bench.py --other off --cpu-sample 0 --warmup 3 "$@" >> $OUT/t13_bench.log 2>&1; }
run "share 20" --levels 0-8 --poses 128 --steps 20
run "share 40" --levels 0-8 --poses 128 --steps 40
run "share 40" --levels 0-8 --poses 128 --steps 40
run "config4 20" --levels 0-8 --steps 20

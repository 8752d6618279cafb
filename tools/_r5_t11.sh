#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t11_bench.log
run() { echo "== $1" >> $OUT/t11_bench.log; shift; python bench.py --other off --cpu-sample 0 --steps 20 --warmup 3 "$@" >> $OUT/t11_bench.log 2>&1; }
run "1080p"
run "1080p"
run "share" --levels 0-8 --poses 128
run "config4" --levels 0-8
run "320" --width 320 --height 200 --poses 8192
run "4k" --width 3840 --height 2160 --poses 256
timeout 900 python -m pytest tests/test_gpu_debug_paths.py tests/test_gpu_raster_parity.py tests/test_golden.py -x -q > $OUT/t11_pytest.log 2>&1

#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t9_bench.log
run() { echo "== $1" >> $OUT/t9_bench.log; shift; python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 "$@" >> $OUT/t9_bench.log 2>&1; }
run "1080p s1" --streams 1
run "1080p s2"
run "1080p s3" --streams 3
run "config5" --big --width 3840 --height 2160 --poses 256 --time-varying
run "share" --levels 0-8 --poses 128
run "big" --big
run "320" --width 320 --height 200 --poses 8192 --streams 3
run "320 nosettle" --width 320 --height 200 --poses 8192 --streams 3 --debug no_settle=1
timeout 900 python -m pytest tests/test_gpu_debug_paths.py tests/test_big_level.py tests/test_gpu_raster_parity.py -x -q > $OUT/t9_pytest.log 2>&1

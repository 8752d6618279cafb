#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t10_bench.log
run() { echo "== $1" >> $OUT/t10_bench.log; shift; python bench.py --other off --cpu-sample 0 --steps 10 --warmup 3 "$@" >> $OUT/t10_bench.log 2>&1; }
run "1080p s2"
run "1080p s3" --streams 3
run "1080p s4" --streams 4
run "config5" --big --width 3840 --height 2160 --poses 256 --time-varying
run "config5 s3" --big --width 3840 --height 2160 --poses 256 --time-varying --streams 3
run "share" --levels 0-8 --poses 128
run "config4" --levels 0-8
run "big" --big
run "big s3" --big --streams 3
run "320" --width 320 --height 200 --poses 8192 --streams 3
run "4k" --width 3840 --height 2160 --poses 256
run "4k s3" --width 3840 --height 2160 --poses 256 --streams 3
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/t10_pytest.log 2>&1
echo "rc=$?" >> $OUT/t10_pytest.log

#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t20.log
run() { echo "== $*" >> $OUT/t20.log; python bench.py --other off --cpu-sample 0 --long 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steps'], d['config']['kernels_ms'], d.get('single_stream',{}).get('ms_per_step'))" >> $OUT/t20.log; }
run --big
run --big --debug bin_threads=512
run --big --poses 2048 --width 1280 --height 720
run --big --poses 2048 --width 1280 --height 720 --debug bin_threads=512

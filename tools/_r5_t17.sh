#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t17.log
run() { echo "== $*" >> $OUT/t17.log; python bench.py --other off --cpu-sample 0 --long 1.5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steps'], d['config']['kernels_ms'])" >> $OUT/t17.log; }
run
run --streams 2
run --debug frag_chunk=8
run --debug frag_chunk=32
run --debug settle_max=16
run --debug settle_max=64
run --debug frag_bw=2
run

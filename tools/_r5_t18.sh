#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t18.log
run() { echo "== $*" >> $OUT/t18.log; python bench.py --other off --cpu-sample 0 --long 1.5 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steps'], d['config']['kernels_ms'])" >> $OUT/t18.log; }
for BW in 3 2; do
run --debug frag_bw=$BW
run --debug frag_bw=$BW --width 3840 --height 2160 --poses 256
run --debug frag_bw=$BW --big
run --debug frag_bw=$BW --width 320 --height 200 --poses 8192
run --debug frag_bw=$BW --levels 0-8
run --debug frag_bw=$BW --width 1280 --height 720 --poses 2048
done
run --debug frag_bw=2 --debug frag_chunk=32
run --debug frag_bw=2 --debug frag_chunk=8
run --debug frag_bw=1
run --debug frag_bw=4

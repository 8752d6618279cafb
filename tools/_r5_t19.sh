#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
rm -f $OUT/t19.log
run() { echo "== $*" >> $OUT/t19.log; python bench.py --other off --cpu-sample 0 --long 1 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['steps'], d['config']['kernels_ms'], d.get('single_stream',{}).get('ms_per_step'))" >> $OUT/t19.log; }
run
run --width 320 --height 200 --poses 8192
run --big
run --big --width 3840 --height 2160 --poses 256 --time-varying
run --levels 0-8 --poses 128
run --levels 0-8
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/t19_pytest.log 2>&1
echo "rc=$?" >> $OUT/t19_pytest.log

#!/bin/bash
# per-kernel rocprofv3 stats of the small-frame and large-level workloads (run from the repo root through gpurun)
set -u
OUT=$PWD/gpurun_out/${1:-prof_sizes}; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD; cd /tmp
for tag in "small:--width 320 --height 200 --poses 8192" "big:--big" "base:"; do
  name=${tag%%:*}; args=${tag#*:}
  rocprofv3 --kernel-trace --stats -d $OUT/$name -o r --output-format csv -- python $ROOT/bench.py $args --streams 1 --steps 5 --warmup 2 --cpu-sample 0 > $OUT/$name.json 2> $OUT/$name.err
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1); echo "== $name"; head -12 "$f" | cut -d, -f1-6
done

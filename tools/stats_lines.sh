#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats, one stream) of the workloads set-up / binning are judged on.
#   tools/stats_lines.sh <tag>   -> gpurun_out/<tag>_stats.txt
TAG=${1:-st}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${TAG}_stats.txt
: > $OUT
i=0
for ARGS in "" "--width 320 --height 200 --poses 8192" "--big" "--big --width 3840 --height 2160 --poses 256 --time-varying"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/st_$TAG_$i -o r --output-format csv -- python $ROOT/bench.py $ARGS --streams 1 --steps 10 --warmup 2 --cpu-sample 0 --other off > /dev/null 2>&1)
  echo "== ${ARGS:-default}" >> $OUT
  python - /tmp/st_$TAG_$i/r_kernel_stats.csv >> $OUT <<'P'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = row['Name'].replace('rdoom_dev::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if float(row['Percentage']) >= 0.3:
        print('   %-60s calls %4s  avg %9.1f us  %5.1f %%' % (name[:60], row['Calls'], float(row['AverageNs']) / 1e3, float(row['Percentage'])))
P
done
cat $OUT

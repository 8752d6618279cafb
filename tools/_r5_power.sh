#!/bin/bash
OUT=$PWD/gpurun_out/r5; mkdir -p $OUT
{
echo "# rocm-smi while bench.py runs (default workload, --long): power and clocks under the overlapped default, then --streams 1"
rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "GPU\[0\]|Power|sclk|mclk|use" | head -12
for S in 3 1; do
  python bench.py --other off --cpu-sample 0 --long 6 --streams $S > $OUT/power_bench_$S.json 2>/dev/null &
  BP=$!
  sleep 9
  for i in 1 2 3 4; do
    echo "-- streams $S sample $i"
    rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Power|sclk|use \(%\)|busy" | head -6
    sleep 0.7
  done
  wait $BP
  tail -1 $OUT/power_bench_$S.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench streams', d['config']['streams'], d['value'], d['ms_per_step'], d['steps'], d.get('long'))"
done
} > $OUT/power.log 2>&1

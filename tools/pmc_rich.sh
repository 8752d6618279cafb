#!/bin/bash
# The texture-rich stand-in (17.9 MB texel store: larger than an XCD's 4 MiB L2) next to the default level: HBM read bytes and
# L2 hit rate of fragment_kernel (PMC passes of their own), the kernel's duration from the same traces.   -> gpurun_out/pmc_rich.txt
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_rich.txt
: > $OUT
for W in "" "--rich"; do
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pr_${W#--}_$i -o p --output-format csv -- python $ROOT/bench.py $W --streams 1 --steps 1 --warmup 1 --cpu-sample 0 --other off > /dev/null 2>&1)
    python - /tmp/pr_${W#--}_$i/p_counter_collection.csv "${W:-default}" >> $OUT <<'P'
import csv, sys, collections
g = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'fragment_kernel' in r['Kernel_Name']:
        g[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 if 'End_Timestamp' in r else 0.0
d = max(g, key=lambda k: sum(g[k].values()))
print('%-8s fragment_kernel %s  (%.3f ms under the counters)' % (sys.argv[2], '  '.join('%s %.1f M' % (k, v / 1e6) if not k.endswith('SIZE') else '%s %.1f MiB' % (k, v / 1024.0) for k, v in sorted(g[d].items())), dur.get(d, 0.0)))
P
  done
done
cat $OUT

#!/bin/bash
# Write amplification of the fragment kernel's framebuffer stores: the default 32 x 16 block stores
# 32-byte row segments (four lanes x 8 bytes), the 64 x 8 block (hook frag_bw=3) 64-byte ones.  Per setting: WRITE_SIZE / FETCH_SIZE
# of fragment_kernel (own --pmc passes) and the kernel's duration from the same passes' traces, then the plain bench line.
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/write_amp.txt
: > $OUT
for BW in 2 3; do
  for C in WRITE_SIZE FETCH_SIZE; do
    (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/wa_${BW}_$C -o p --output-format csv -- python $ROOT/bench.py --streams 1 --steps 1 --warmup 1 --cpu-sample 0 --other off --debug frag_bw=$BW > /dev/null 2>&1)
    python - /tmp/wa_${BW}_$C/p_counter_collection.csv $BW $C >> $OUT <<'P'
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'fragment_kernel' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[3]]
g = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'fragment_kernel' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[3]:
        g.setdefault(r['Dispatch_Id'], 0.0); g[r['Dispatch_Id']] += float(r['Counter_Value'])
big = max(g.values())
print('frag_bw=%s %s per launch: %.1f MiB (x 1024 = %.3f GB)' % (sys.argv[2], sys.argv[3], big / 1024.0, big * 1024 / 1e9))
P
  done
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --other off --debug frag_bw=$BW 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('frag_bw=$BW bench: %.1f Gpixel/s, %.3f ms per step, fragment stage %.3f ms' % (d['value']/1e3, d['ms_per_step'], d['config']['kernels_ms']['fragment']))" >> $OUT
done
cat $OUT

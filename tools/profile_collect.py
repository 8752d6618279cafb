#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked):
   profiles/r<NN>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python bench.py`
   profiles/r<NN>_pmc_summary.txt       per-kernel PMC means per dispatch
   profiles/r<NN>_bench_1gpu.json       the un-profiled bench line of the same box
   profiles/pmc_fragment_latest.json    HBM bytes per fragment_kernel launch (read by bench.py -> roofline.traffic)
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of wide (16 B/lane) reads
(MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], int(sys.argv[2])
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'stats', 'r_kernel_stats.csv'), os.path.join(dst, 'r%02d_kernel_stats.csv' % rnd))
line = open(os.path.join(src, 'bench_plain.json')).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, 'r%02d_bench_1gpu.json' % rnd), 'w').write(line + '\n')
files = sorted(glob.glob(os.path.join(src, 'pmc*', 'p_counter_collection.csv')))
txt = subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py')] + files, text=True)
open(os.path.join(dst, 'r%02d_pmc_summary.txt' % rnd), 'w').write(txt)
vals = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if 'fragment_kernel' in row['Kernel_Name'] and row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals[row['Counter_Name']].append(float(row['Counter_Value']))
fetch = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE'])
write = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE'])
bench = json.loads(line)
out = {'kernel': 'fragment_kernel', 'poses': bench['config']['poses_per_gpu'], 'width': bench['config']['width'],
       'height': bench['config']['height'], 'workload': bench['config']['workload_key'],
       'kernel_sources': bench['config']['kernel_sources'],  # bench.py quotes the traffic only for this workload AND these device sources
       'FETCH_SIZE_KiB': fetch, 'WRITE_SIZE_KiB': write,
       'hbm_read_bytes_per_launch': 2.0 * fetch * 1024.0, 'hbm_write_bytes_per_launch': write * 1024.0,
       'hbm_bytes_per_launch': 2.0 * fetch * 1024.0 + write * 1024.0,
       'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950, 16 B/lane reads)',
       'round': rnd}
json.dump(out, open(os.path.join(dst, 'pmc_fragment_latest.json'), 'w'), indent=1)
print(json.dumps(out))

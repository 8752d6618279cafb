#!/usr/bin/env python3
"""Copies the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked):
   profiles/r<NN>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python bench.py`
   profiles/r<NN>_pmc_summary.txt       per-kernel PMC means per dispatch
   profiles/r<NN>_bench_1gpu.json       the un-profiled bench line of the same box
   profiles/r<NN>_bench_other.jsonl     bench lines of the other workloads (frame sizes, big level, level sweep, 2 ranks)
   profiles/r<NN>_summary.md            the tables DESIGN.md quotes, generated from the files above
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
if os.path.exists(os.path.join(src, 'stats_default', 'r_kernel_stats.csv')):
    shutil.copy(os.path.join(src, 'stats_default', 'r_kernel_stats.csv'), os.path.join(dst, 'r%02d_kernel_stats_default.csv' % rnd))
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
# the VALU-issue roofline of the two hot kernels ("the yardstick no longer measures what binds the step"):
# per launch, means over the dispatches of the whole batch
def kernel_counters(key):
    acc = collections.defaultdict(list)
    for f in files:
        per = collections.defaultdict(dict)
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            if key in name and ('quadrant' in key) == ('fragment_quadrant' in name):
                per[row['Dispatch_Id']][row['Counter_Name']] = per[row['Dispatch_Id']].get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
                per[row['Dispatch_Id']]['_grid'] = int(row['Grid_Size'])
        if per:
            big = max(d['_grid'] for d in per.values())
            for d in per.values():
                if d['_grid'] == big:
                    for c, v in d.items():
                        acc[c].append(v)
    return {c: sum(v) / len(v) for c, v in acc.items()}


def valu_roofline(pixels):
    out, step_issue, step_insts = {}, 0.0, 0.0
    for key in ('fragment_kernel', 'fragment_quadrant_kernel', 'raster_wave_kernel'):
        c = kernel_counters(key)
        if not all(k in c for k in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE')):
            continue
        cycles = c['GRBM_GUI_ACTIVE'] / 8.0                  # the counter sums the 8 XCDs
        issue = c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0      # quad-cycles over all SIMDs -> cycles per SIMD
        out[key] = {'insts_per_pixel': round(c['SQ_INSTS_VALU'] * 64.0 / pixels, 2), 'wave_insts_per_launch': c['SQ_INSTS_VALU'],
                    'issue_cycles': round(issue), 'kernel_cycles': round(cycles), 'frac': round(issue / cycles, 4)}
        step_issue += issue
        step_insts += c['SQ_INSTS_VALU']
    if out:
        out['note'] = ('lane-instructions per output pixel = SQ_INSTS_VALU x 64 / pixels; issue cycles = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs; '
                       'kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; separate --pmc passes of bench.py --streams 1 (tools/profile_round.sh).  The counter '
                       'charges every VALU instruction one quad-cycle: tools/ubench_valu.hip measures 2.4 cycles for v_fma/mul/add_f32, v_add_u32, v_and/or_b32 '
                       'and 4.2 for the rest, so the fraction is an upper bound of the true issue share')
        out['step'] = {'insts_per_pixel': round(step_insts * 64.0 / pixels, 2), 'issue_cycles': round(step_issue)}
    return out


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
       'round': rnd, 'taken': __import__('datetime').datetime.utcnow().strftime('%Y-%m-%d') + ' (round %d, tools/profile_round.sh)' % rnd}
out['valu'] = valu_roofline(bench['config']['poses_per_gpu'] * bench['config']['width'] * bench['config']['height']) or None
json.dump(out, open(os.path.join(dst, 'pmc_fragment_latest.json'), 'w'), indent=1)
print(json.dumps(out))

# ---- the tables DESIGN.md quotes -----------------------------------------------------------------------------------
other = os.path.join(src, 'bench_other.jsonl')
lines = [json.loads(x) for x in open(other)] if os.path.exists(other) else []
if lines:
    shutil.copy(other, os.path.join(dst, 'r%02d_bench_other.jsonl' % rnd))
md = ['# Round %d measurements (one MI355X; generated by tools/profile_collect.py from gpurun_out/%s)' % (rnd, tag), '',
      '## rocprofv3 --kernel-trace --stats of `python bench.py --streams 1` (E1M1, 1024 poses, 1920x1080: one launch per kernel and step,',
      'nothing overlapping -- the per-kernel accounting the bench line\'s `kernels_ms` / `roofline` must agree with)', '',
      '| kernel | calls | average (us) | share |', '|---|---|---|---|']
for row in csv.DictReader(open(os.path.join(dst, 'r%02d_kernel_stats.csv' % rnd))):
    if float(row['Percentage']) < 0.05:
        continue
    name = row['Name'].replace('rdoom_dev::(anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
    md.append('| `%s` | %s | %.1f | %.1f %% |' % (name.replace('void ', ''), row['Calls'], float(row['AverageNs']) / 1e3, float(row['Percentage'])))
trace = os.path.join(src, 'stats_default', 'r_kernel_trace.csv')
if os.path.exists(trace):
    rows = [r for r in csv.DictReader(open(trace)) if 'fragment_kernel' in r['Kernel_Name'] or 'raster_wave_kernel' in r['Kernel_Name'] or 'settle_kernel' in r['Kernel_Name']]
    md += ['', '## the default command (sub-batches on the stream pool, then its single-stream pass over the whole batch): per-launch durations from the kernel trace', '',
           '| kernel | launches | mean of the sub-batch launches on the pool (warm-up + timed region; they overlap) us | mean of the last K launches (the single-stream pass: whole batch, one launch per step) us |', '|---|---|---|---|']
    for key in ('fragment_kernel', 'raster_wave_kernel', 'settle_kernel'):
        d = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if key in r['Kernel_Name']]
        d.sort()
        dur = [x[1] for x in d]
        k2 = bench.get('steps_requested', bench['steps'])  # the single-stream pass runs the requested steps (the timed region is stretched to 0.5 s)
        if len(dur) > k2:
            clean, over = dur[-k2:], dur[:-k2]
            md.append('| `%s` | %d | %.1f | %.1f |' % (key, len(dur), sum(over) / len(over) / 1e3, sum(clean) / len(clean) / 1e3))
md += ['', '## bench lines', '', '| workload | GPUs | Mpixel/s | frames/s | setup+bin / raster / fragment (ms) | fragment vs 6 B/px HBM-read roofline |',
       '|---|---|---|---|---|---|']
for d in [bench] + lines:
    k = d['config'].get('kernels_ms')   # (the threads launcher times whole steps only)
    md.append('| %s%s%s | %d | %.0f | %.0f | %s | %s |' % (
        d['config']['workload'], (' [%s scaling; %s]' % (d['scaling'], d.get('note', ''))) if d['n_gpus'] > 1 else '',
        (' [launcher: %s]' % d['config']['launcher']) if 'launcher' in d['config'] else '', d['n_gpus'], d['value'],
        d['frames_per_s'], ('%.2f / %.2f / %.2f' % (k['setup'], k['raster'], k['fragment'])) if k else '--',
        ('%.1f %%' % (100.0 * d['roofline']['frac'])) if d.get('roofline') else '--'))
    if d.get('scaling_proxy'):
        sp = d['scaling_proxy']
        md.append('| &nbsp;&nbsp;its 1/%d share (%d poses per level), 1 / 2 / 3 streams: %s ms per step; full batch %.3f ms -> predicted speed-up at %d GPUs %.2f (ideal share %.3f ms) | 1 | | | | |'
                  % (sp['gpus'], sp['poses_per_level_share'], ' / '.join('%.3f' % sp['share_ms_by_streams'][s] for s in ('1', '2', '3')), sp['full_ms'], sp['gpus'],
                     sp['predicted_speedup_at_%d' % sp['gpus']], sp['ideal_share_ms']))
hooks = os.path.join(src, 'bench_hooks.jsonl')
if os.path.exists(hooks):
    shutil.copy(hooks, os.path.join(dst, 'r%02d_bench_hooks.jsonl' % rnd))
    md += ['', '## the same box with this round\'s hooks (equivalent paths: same images)', '', '| hook | workload | Mpixel/s | ms per step | setup+bin / raster / fragment (ms) |', '|---|---|---|---|---|']
    for d in [json.loads(x) for x in open(hooks)]:
        k = d['config']['kernels_ms']
        md.append('| %s | %s | %.0f | %.3f | %.2f / %.2f / %.2f |' % (' '.join(d['config'].get('debug', [])), d['config']['workload'][:90], d['value'], d['ms_per_step'], k['setup'], k['raster'], k['fragment']))
frag_ms = bench['config']['kernels_ms']['fragment']
md += ['', '## fragment kernel, HBM traffic (PMC, separate passes)', '',
       'FETCH_SIZE %.0f KiB x 2 (gfx950: wide reads counted half) = %.2f GB read, WRITE_SIZE %.0f KiB = %.2f GB written per launch; '
       'algorithmic bytes (6 B/px) %.2f GB.  At %.3f ms per launch: algorithmic %.0f GB/s = %.1f %% of 8 TB/s; actual traffic %.0f GB/s = %.1f %% of 8 TB/s.'
       % (fetch, 2 * fetch * 1024 / 1e9, write, write * 1024 / 1e9, bench['config']['poses_per_gpu'] * bench['config']['width'] * bench['config']['height'] * 6 / 1e9,
          frag_ms, bench['roofline']['achieved'], 100 * bench['roofline']['frac'], out['hbm_bytes_per_launch'] / frag_ms / 1e6,
          100 * out['hbm_bytes_per_launch'] / frag_ms / 1e6 / 8000.0), '']
open(os.path.join(dst, 'r%02d_summary.md' % rnd), 'w').write('\n'.join(md) + '\n')
print('wrote profiles/r%02d_summary.md' % rnd)

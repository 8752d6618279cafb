#!/usr/bin/env python3
"""Summary of tools/ab_cycles.sh: per variant and hot kernel, from ONE set of launches -- duration (us), GPU cycles
(GRBM_GUI_ACTIVE / 8 XCDs), effective clock (GHz), VALU instructions per launch (millions), VALU issue share
(SQ_ACTIVE_INST_VALU x 4 cycles / 1024 SIMDs / kernel cycles), issue-stall share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).

    python tools/ab_cycles_summary.py gpurun_out/<dir>
"""
import collections
import csv
import glob
import os
import re
import sys

HOT = ('fragment_kernel', 'fragment_quadrant_kernel', 'raster_wave_kernel', 'bin_kernel', 'setup_kernel', 'cull_kernel')


def main(root):
    rows = collections.OrderedDict()
    for path in sorted(glob.glob(os.path.join(root, '*.r*', '**', '*counter_collection.csv'), recursive=True)):
        variant = re.sub(r'\.r\d+$', '', os.path.relpath(path, root).split(os.sep)[0])
        per = collections.defaultdict(dict)   # dispatch -> counter -> value (+ kernel, duration)
        with open(path) as f:
            for r in csv.DictReader(f):
                d = per[r['Dispatch_Id']]
                d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
                d['_k'] = r['Kernel_Name']
                d['_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                d['_grid'] = int(r['Grid_Size'])
        for d in per.values():
            k = next((h for h in HOT if h in d['_k'] and (h != 'fragment_kernel' or 'quadrant' not in d['_k'])), None)
            if k is None:
                continue
            rows.setdefault((variant, k), []).append(d)
    print('%-22s %-26s %4s %9s %11s %6s %9s %8s %8s %8s' % ('variant', 'kernel', 'n', 'us', 'cycles', 'GHz', 'VALU (M)', 'SALU (M)', 'VALU %', 'stall %'))
    for (variant, k), ds in rows.items():
        big = max(d['_grid'] for d in ds)
        ds = [d for d in ds if d['_grid'] == big]   # the full-batch launches only
        n = len(ds)
        mean = lambda key: sum(d.get(key, 0.0) for d in ds) / n
        us, cyc = mean('_us'), mean('GRBM_GUI_ACTIVE') / 8.0
        valu_share = mean('SQ_ACTIVE_INST_VALU') * 4.0 / 1024.0 / cyc if cyc else 0.0
        stall = mean('SQ_WAIT_INST_ANY') / mean('SQ_WAVE_CYCLES') if mean('SQ_WAVE_CYCLES') else 0.0
        print('%-22s %-26s %4d %9.1f %11.0f %6.3f %9.2f %8.2f %8.1f %8.1f' % (variant, k, n, us, cyc, cyc / us / 1e3 if us else 0.0, mean('SQ_INSTS_VALU') / 1e6,
                                                                       mean('SQ_INSTS_SALU') / 1e6, 100.0 * valu_share, 100.0 * stall))


if __name__ == '__main__':
    main(sys.argv[1])

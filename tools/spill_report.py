#!/usr/bin/env python3
"""Where do a kernel's scalar-register spills sit?  Reads `hipcc -S -gline-tables-only` output: a VGPR that is ever the destination
of v_writelane_b32 is a spill register (the kernels never use v_writelane themselves); every v_writelane_b32 to one is a spill
store, every v_readlane_b32 from one a reload.  Reported per basic block with the source lines the block's instructions come
from (.loc directives), so that "inside the walk loop" can be read off against raster.hip's line numbers.

    hipcc --offload-arch=gfx950 -x hip -O3 ... -gline-tables-only --cuda-device-only -S raster.hip -o raster.s
    python tools/spill_report.py raster.s 'raster_wave_kernelILb0ELb1ELb0ELb1ELb1E' [source-file-substring]
"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    only = sys.argv[3] if len(sys.argv) > 3 else 'raster.hip'
    lines = open(path).read().split('\n')
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and want in l and ':' in l)
    body = []
    for l in lines[start + 1:]:
        if l.strip().startswith('.Lfunc_end'):
            break
        body.append(l)
    spill_regs = set()
    for l in body:
        m = re.match(r'\s*v_writelane_b32\s+(v\d+),', l)
        if m:
            spill_regs.add(m.group(1))
    blocks, name, cur_line = [], 'entry', None
    cur = {'name': name, 'st': 0, 'ld': 0, 'valu': 0, 'lines': set()}
    for l in body:
        s = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append(cur)
            cur = {'name': m.group(1), 'st': 0, 'ld': 0, 'valu': 0, 'lines': set()}
            continue
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
        if m:
            cur_line = int(m.group(2)) if only in files.get(int(m.group(1)), '') else None
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        op = s.split()[0]
        if op.startswith('v_'):
            cur['valu'] += 1
        if cur_line is not None:
            cur['lines'].add(cur_line)
        m = re.match(r'v_writelane_b32\s+(v\d+),', s)
        if m and m.group(1) in spill_regs:
            cur['st'] += 1
        m = re.match(r'v_readlane_b32\s+\S+,\s*(v\d+),', s)
        if m and m.group(1) in spill_regs:
            cur['ld'] += 1
    blocks.append(cur)
    print('spill VGPRs: %s' % ' '.join(sorted(spill_regs, key=lambda r: int(r[1:]))))
    print('%-12s %5s %6s %6s  source lines' % ('block', 'VALU', 'stores', 'reloads'))
    tst = tld = 0
    for b in blocks:
        tst += b['st']
        tld += b['ld']
        if b['st'] or b['ld']:
            ls = sorted(b['lines'])
            print('%-12s %5d %6d %6d  %s' % (b['name'], b['valu'], b['st'], b['ld'], ('%d-%d' % (ls[0], ls[-1])) if ls else '-'))
    print('total: %d spill stores, %d reloads in %d blocks' % (tst, tld, len(blocks)))


if __name__ == '__main__':
    main()

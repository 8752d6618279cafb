#!/usr/bin/env python3
"""Basic-block census of a gfx950 kernel from hipcc -S output: instruction counts per class and an issue-cycle estimate with the
per-class costs measured by tools/ubench_valu.hip on MI355X (profiles/r04_ubench_valu.txt): fast VALU (v_fma/mul/add_f32,
v_add_u32, v_and/or_b32, v_mov_b32) 2.4 cycles per wave64 instruction, every other VALU 4.2, v_rcp_f32 8.2.

    python tools/isa_blocks.py file.s kernel_name_substring [min_cycles]
"""
import re
import sys

FAST = ('v_fma_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_mov_b32',
        'v_fmac_f32', 'v_mac_f32', 'v_add_co_u32', 'v_addc_co_u32', 'v_mad_f32')
TRANS = ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_exp_f32', 'v_log_f32', 'v_sin_f32', 'v_cos_f32', 'v_rcp_iflag_f32')


def cost(op):
    base = re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
    if base in TRANS:
        return 8.2
    if base in FAST:
        return 2.4
    return 4.2


def main():
    path, want = sys.argv[1], sys.argv[2]
    min_cycles = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and want in l and l.rstrip().endswith(('@' + l.split(':')[0], ':')) or (l.startswith('_Z') and want in l and ':' in l))
    blocks, cur, name = [], [], 'entry'
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith('.Lfunc_end'):
            break
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        if not s or s.startswith((';', '.', '//')):
            continue
        cur.append(s.split()[0])
        if s.startswith(('s_cbranch', 's_branch', 's_endpgm')):  # a block ends at its branch: the fall-through part gets its own row
            blocks.append((name, cur))
            name, cur = name.rstrip("'") + "'", []
    blocks.append((name, cur))
    tot = {'valu': 0, 'salu': 0, 'vmem': 0, 'lds': 0, 'smem': 0, 'cyc': 0.0}
    print('%-12s %6s %6s %5s %5s %5s %5s %9s  top VALU' % ('block', 'insts', 'VALU', 'SALU', 'VMEM', 'LDS', 'SMEM', 'VALU cyc'))
    for name, ops in blocks:
        valu = [o for o in ops if o.startswith('v_')]
        salu = [o for o in ops if o.startswith('s_') and not o.startswith(('s_load', 's_buffer_load', 's_waitcnt', 's_nop', 's_memtime'))]
        vmem = [o for o in ops if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))]
        lds = [o for o in ops if o.startswith('ds_')]
        smem = [o for o in ops if o.startswith(('s_load', 's_buffer_load'))]
        cyc = sum(cost(o) for o in valu)
        for k, v in (('valu', len(valu)), ('salu', len(salu)), ('vmem', len(vmem)), ('lds', len(lds)), ('smem', len(smem)), ('cyc', cyc)):
            tot[k] += v
        if cyc >= min_cycles:
            hist = {}
            for o in valu:
                hist[o] = hist.get(o, 0) + 1
            top = ', '.join('%s x%d' % kv for kv in sorted(hist.items(), key=lambda kv: -kv[1])[:8])
            print('%-12s %6d %6d %5d %5d %5d %5d %9.0f  %s' % (name, len(ops), len(valu), len(salu), len(vmem), len(lds), len(smem), cyc, top))
    print('static total: VALU %d SALU %d VMEM %d LDS %d SMEM %d, VALU cycles %.0f' % (tot['valu'], tot['salu'], tot['vmem'], tot['lds'], tot['smem'], tot['cyc']))


if __name__ == '__main__':
    main()

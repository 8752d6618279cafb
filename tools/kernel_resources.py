#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every gfx950 kernel in the SHIPPED library (rust-doom_amd/librdoom_hip.so), read
from the code objects embedded in it: the .so is copied to a scratch directory, `llvm-objdump --offloading` extracts
the device code objects, `llvm-readelf --notes` prints their AMDGPU metadata.  DESIGN section 5 quotes this table;
tests/test_kernel_resources.py asserts that the hot kernels use no scratch memory.

    python tools/kernel_resources.py [--json]
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
FIELDS = ('vgpr_count', 'vgpr_spill_count', 'sgpr_count', 'sgpr_spill_count', 'private_segment_fixed_size',
          'group_segment_fixed_size', 'max_flat_workgroup_size')


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.split('\n')[:len(names)]


def kernel_resources(lib=os.path.join(ROOT, 'rust-doom_amd', 'librdoom_hip.so')):
    tmp = tempfile.mkdtemp(prefix='rdoom_co_')
    try:
        so = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', so], capture_output=True, check=True, cwd=tmp)
        kernels = {}
        for co in sorted(glob.glob(so + '.*gfx950*')):
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
            for block in notes.split('- .agpr_count:')[1:]:
                m = re.search(r'\.name:\s+(\S+)', block)
                if not m:
                    continue
                rec = {}
                for f in FIELDS:
                    v = re.search(r'\.%s:\s+(\d+)' % f, block)
                    rec[f] = int(v.group(1)) if v else None
                kernels[m.group(1)] = rec
        names = list(kernels)
        return {d: kernels[n] for n, d in zip(names, demangle(names))}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name).replace('rdoom_dev::', '')


def main():
    res = kernel_resources()
    if '--json' in sys.argv:
        print(json.dumps({short(k): v for k, v in res.items()}, indent=1, sort_keys=True))
        return
    print('%-44s %5s %6s %5s %6s %8s %6s' % ('kernel', 'VGPR', 'vspill', 'SGPR', 'sspill', 'scratch', 'LDS'))
    for k in sorted(res, key=short):
        r = res[k]
        print('%-44s %5d %6d %5d %6d %8d %6d' % (short(k)[:44], r['vgpr_count'], r['vgpr_spill_count'], r['sgpr_count'],
                                                 r['sgpr_spill_count'], r['private_segment_fixed_size'], r['group_segment_fixed_size']))


if __name__ == '__main__':
    main()

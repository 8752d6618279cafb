"""Builds librdoom_hip.so in-tree: HIP kernels for gfx950 + the C++ host library behind include/rdoom.h.

hipcc cross-compiles gfx950 without a GPU.  The built .so is git-ignored but travels to the GPU box
with the repo snapshot.  `python rust-doom_amd/build.py` or `__graft_entry__.build()`.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, 'librdoom_hip.so')
OBJ = os.path.join(HERE, 'csrc', '_obj')
COMMON = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'),
          '-Wall', '-Wno-unused-function', '-Wno-bitwise-instead-of-logical']  # bitwise and on bools is deliberate (branch-free)


# per-unit extra flags.  raster.hip: LLVM's SLP vectoriser packs the boolean results of independent compares into 16-bit lanes
# (v_cndmask 0/1, shifts, ors) where four compares and scalar ANDs would do, and the packed fmas it also forms do not make up
# for it: without it the rasteriser is 1-1.5 % faster on every workload (A/B on one box, profiles/r03_ab.txt)
UNIT_FLAGS = {'raster.hip': ['-fno-slp-vectorize']}


def sources():
    hip = sorted(glob.glob(os.path.join(HERE, 'csrc', 'hip', '*.hip')))
    cpp = sorted(glob.glob(os.path.join(HERE, 'csrc', 'host', '*.cpp')))
    return hip, cpp


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hip, cpp = sources()
    headers = glob.glob(os.path.join(HERE, 'csrc', '**', '*.hpp'), recursive=True) + \
        glob.glob(os.path.join(ROOT, 'include', '*.h')) + [os.path.abspath(__file__)]
    os.makedirs(OBJ, exist_ok=True)
    objs, jobs = [], []
    for src in hip + cpp:
        obj = os.path.join(OBJ, os.path.basename(src) + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            if src.endswith('.hip'):
                cmd = [hipcc, '--offload-arch=gfx950', '-x', 'hip'] + COMMON + UNIT_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
            else:
                cmd = [hipcc, '-x', 'c++'] + COMMON + ['-c', src, '-o', obj]
            jobs.append(cmd)
    if jobs:  # one translation unit per kernel: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if force or _stale(OUT, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

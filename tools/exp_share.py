#!/usr/bin/env python3
"""Experiment (GPU box, ctypes only): the 1/8 strong-scaling share of BASELINE config 4 -- E1M1..E1M9 at 1080p, 128 poses per
level -- under different ways of queueing the nine levels' renders: how many HIP streams in all, sub-batches per level,
profiled (four events per render) or plain renders.  Prints host enqueue time and GPU step time per variant.
usage: python tools/exp_share.py [--poses 128] [--steps 10]"""
import argparse
import ctypes
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_doom_amd as rd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--poses', type=int, default=128)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--levels', type=int, default=9)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
a = ap.parse_args()
sharding = importlib.import_module('rust-doom_amd.sharding')
syn = importlib.import_module('rust-doom_amd.synthetic')
hip = ctypes.CDLL('libamdhip64.so')
rd.set_device(0)
wad = rd.Wad(syn.ensure_wad(), syn.META_PATH)
levels = []
for i in range(a.levels):
    built = wad.build_level(i, gpu_tessellation=True)
    levels.append((built, rd.DeviceLevel(built), sharding.pose_sweep(rd, built, a.poses, a.width, a.height), built.lights_at(0.0)))


def streams(n, flags=0):
    out = []
    for _ in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), flags) == 0
        out.append(s)
    return out


def variant(name, n_streams, parts, profiled, assign='part'):
    ss = streams(n_streams) if n_streams else [ctypes.c_void_p(0)]
    work = []
    for li, (built, level, poses, lights) in enumerate(levels):
        for p in range(parts):
            lo, hi = sharding.shard_range(a.poses, p, parts)
            b = rd.Batch(level, a.width, a.height, hi - lo)
            k = (li * parts + p) if assign == 'all' else (p if assign == 'part' else li)
            work.append((b, poses[lo:hi], lights, ss[k % len(ss)].value))

    def step():
        for b, p, l, st in work:
            (b.render_profiled if profiled else b.render)(p, l, stream=st)

    for _ in range(3):
        step()
    for b, *_ in work:
        if profiled:
            b.collect_timings()
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    worst = 0.0
    for _ in range(a.steps):
        ts = time.perf_counter()
        step()
        worst = max(worst, time.perf_counter() - ts)
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    for b, *_ in work:
        if profiled:
            b.collect_timings()
        b.close()
    for s in ss:
        if s.value:
            hip.hipStreamDestroy(s)
    print('%-58s host enqueue %7.3f ms/step (slowest step %.2f ms)   step %7.3f ms' % (name, (t1 - t0) / a.steps * 1e3, worst * 1e3, (t2 - t0) / a.steps * 1e3), flush=True)


variant('null stream, 1 part, plain', 0, 1, False)
variant('1 stream, 1 part, plain', 1, 1, False)
variant('1 stream, 1 part, profiled', 1, 1, True)
variant('2 streams (by part), 2 parts, profiled', 2, 2, True)
variant('2 streams (by part), 2 parts, plain', 2, 2, False)
variant('2 streams (levels alternate), 1 part, plain', 2, 1, False, 'level')
variant('3 streams (levels alternate), 1 part, plain', 3, 1, False, 'level')
variant('4 streams (levels alternate), 1 part, plain', 4, 1, False, 'level')
variant('9 streams (one per level), 1 part, plain', 9, 1, False, 'level')
variant('18 streams (one per level and part), 2 parts, plain', 18, 2, False, 'all')
variant('18 streams (one per level and part), 2 parts, profiled', 18, 2, True, 'all')
variant('3 streams (by part), 3 parts, plain', 3, 3, False)
variant('4 streams (levels alternate), 2 parts, plain', 4, 2, False, 'all')

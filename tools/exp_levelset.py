#!/usr/bin/env python3
"""Experiment (GPU box, ctypes only): the 1/8 strong-scaling share of BASELINE config 4 -- E1M1..E1M9 at 1080p, 128 poses per
level -- as nine per-level batches alternating over three streams (round 5's way) against ONE level set rendered as S sub-batches
of mixed poses on S streams (rdoom_levelset_create + rdoom_batch_render_levels), and the full batch (1024 poses per level) both ways.
Prints GPU step times; the frames of both ways are compared once (must be identical).
usage: python tools/exp_levelset.py [--poses 128] [--steps 20] [--full]"""
import argparse
import ctypes
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_doom_amd as rd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--poses', type=int, default=128)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--levels', type=int, default=9)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
ap.add_argument('--full', action='store_true', help='also the full batch (8 x poses per level)')
ap.add_argument('--no-check', action='store_true')
a = ap.parse_args()
sharding = importlib.import_module('rust-doom_amd.sharding')
syn = importlib.import_module('rust-doom_amd.synthetic')
hip = ctypes.CDLL('libamdhip64.so')
rd.set_device(0)
wad = rd.Wad(syn.ensure_wad(), syn.META_PATH)


def streams(n):
    out = []
    for _ in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 0) == 0
        out.append(s)
    return out


def time_steps(step, label, steps):
    for _ in range(3):
        step()
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    hip.hipDeviceSynchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('%-86s %8.3f ms/step' % (label, ms), flush=True)
    return ms


def run(n_per_level, tag):
    built = [wad.build_level(i, gpu_tessellation=True) for i in range(a.levels)]
    poses = [sharding.pose_sweep(rd, b, n_per_level, a.width, a.height) for b in built]
    lights = [b.lights_at(0.0) for b in built]
    # (a) one batch per level, the levels alternate over three streams
    ss = streams(3)
    levels = [rd.DeviceLevel(b) for b in built]
    batches = [rd.Batch(lv, a.width, a.height, n_per_level) for lv in levels]

    def per_level():
        for li, b in enumerate(batches):
            b.render(poses[li], lights[li], stream=ss[li % 3].value)
    base = time_steps(per_level, '%s: nine per-level batches alternating over 3 streams' % tag, a.steps)
    ref = None
    if not a.no_check:
        ref = [b.read_framebuffer(0, min(4, n_per_level)) for b in batches]
    for b in batches:
        b.close()
    for lv in levels:
        lv.close()
    # (b) one level set
    lset = rd.DeviceLevelSet(built)
    all_poses = np.concatenate(poses)
    all_lights = np.concatenate([np.tile(l, (n_per_level, 1)) for l in lights])
    lop = np.repeat(np.arange(a.levels, dtype=np.uint32), n_per_level)
    n = len(all_poses)
    for order_name, order in (('grouped by level', np.arange(n)), ('interleaved', np.arange(n).reshape(a.levels, n_per_level).T.reshape(-1))):
        for parts in (1, 2, 3, 4):
            work = []
            for p in range(parts):
                lo, hi = sharding.shard_range(n, p, parts)
                sel = order[lo:hi]
                work.append((rd.Batch(lset, a.width, a.height, hi - lo), all_poses[sel], all_lights[sel], lop[sel], ss[p % 3].value if parts > 1 else None, sel))

            def merged():
                for b, p, l, lp, st, _sel in work:
                    b.render(p, l, stream=st, level_of_pose=lp)
            ms = time_steps(merged, '%s: ONE level set, %d sub-batch(es) of mixed poses (%s)' % (tag, parts, order_name), a.steps)
            print('    -> %.3f x the per-level way' % (ms / base))
            if ref is not None and parts == 3:
                for b, _p, _l, lp, _st, sel in work:
                    fb = b.read_framebuffer()
                    for j, gi in enumerate(sel):
                        li, k = divmod(int(gi), n_per_level)
                        if k < len(ref[li]):
                            assert np.array_equal(fb[j], ref[li][k]), (order_name, li, k)
                print('    frames identical to the per-level batches (first %d poses of every level)' % len(ref[0]))
            for b, *_ in work:
                b.close()
    lset.close()


run(a.poses, 'share (%d poses per level)' % a.poses)
if a.full:
    run(a.poses * 8, 'full (%d poses per level)' % (a.poses * 8))

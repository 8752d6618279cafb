#!/usr/bin/env python3
"""Census of the rasteriser's paths (rdoom_debug_set raster_stats=1: instrumented instantiation, stderr) for one batch.
usage (GPU box): python tools/raster_stats.py [--big] [--poses N] [--width W --height H]"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_doom_amd as rd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--big', action='store_true')
ap.add_argument('--poses', type=int, default=256)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
ap.add_argument('--level', type=int, default=0)
a = ap.parse_args()
sharding = importlib.import_module('rust-doom_amd.sharding')
syn = importlib.import_module('rust-doom_amd.synthetic')
rd.set_device(0)
for hook in os.environ.get('RDOOM_STATS_HOOK', '').split():  # e.g. RDOOM_STATS_HOOK=no_split: the census of an equivalent path
    rd.debug_set(hook, 1)
built = rd.Wad(syn.ensure_big_wad() if a.big else syn.ensure_wad(), syn.META_PATH).build_level(a.level, gpu_tessellation=True)
level = rd.DeviceLevel(built)
batch = rd.Batch(level, a.width, a.height, a.poses)
poses = sharding.pose_sweep(rd, built, a.poses, a.width, a.height)
lights = built.lights_at(0.0)
batch.render(poses, lights, timed=True)
print('plain   ', batch.render(poses, lights, timed=True))
rd.debug_set('raster_stats', 1)
print('census  ', batch.render(poses, lights, timed=True))

#!/usr/bin/env python3
"""Experiment (GPU box): does the PHASE between the two half-batch streams matter?  bench.py's default queues [set-up, raster,
fragment] of half A on stream 0 and of half B on stream 1, steps back to back; whether one half's rasteriser meets the other's
fragment kernel is left to chance.  Here stream 1 is delayed by a spin kernel of d microseconds before the timed steps, and an
event chain variant forces strict alternation (raster of B may only start when raster of A is done, and so on).
    python tools/exp_phase.py [steps]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_doom_amd as rd  # noqa: E402

sharding = importlib.import_module('rust-doom_amd.sharding')
synthetic = importlib.import_module('rust-doom_amd.synthetic')


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    W, H, N = 1920, 1080, 1024
    built = rd.Wad(synthetic.ensure_wad(), synthetic.META_PATH).build_level(0, gpu_tessellation=True)
    level = rd.DeviceLevel(built)
    poses = sharding.pose_sweep(rd, built, N, W, H)
    lights = built.lights_at(0.0)
    halves = [(rd.Batch(level, W, H, N // 2), poses[:N // 2], torch.cuda.Stream()), (rd.Batch(level, W, H, N // 2), poses[N // 2:], torch.cuda.Stream())]
    px = N * W * H

    def run(k):
        for _ in range(k):
            for b, p, s in halves:
                b.render_profiled(p, lights, stream=s.cuda_stream)
            if _ % 25 == 24:
                for b, _p, _s in halves:
                    b.collect_timings()

    for delay_us in (0, 0, 300, 700, 1000, 1400, 1900, 2400, 0):
        run(3)
        for b, _p, _s in halves:
            b.collect_timings()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if delay_us:
            with torch.cuda.stream(halves[1][2]):
                torch.cuda._sleep(int(delay_us * 2100))   # cycles of the spin kernel's clock (~2.1 GHz)
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for b, _p, _s in halves:
            b.collect_timings()
        print('stream 1 delayed by %4d us: %.3f ms per step (delay included), %.1f Gpixel/s' % (delay_us, dt / steps * 1e3, px * steps / dt / 1e9), flush=True)


if __name__ == '__main__':
    main()

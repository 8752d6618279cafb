#!/usr/bin/env python3
"""bench.py -- pose-batch throughput of the MI355X renderer (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic camera poses:
  H2D of the pose constants -> setup kernel -> tiled rasteriser -> fragment kernel,
for `--poses` poses (default 1024) at `--width`x`--height` (default 1920x1080) of level E1M1 of the
synthetic IWAD (no DOOM1.WAD exists here; pass --iwad/--metadata to use a real one).  Level arrays are
resident in HBM before the timed region; framebuffers stay on the device (D2H is not part of the metric).

Multi-GPU: one process per GPU (torch.distributed / RCCL used ONLY for the timing barrier and the
max-over-ranks reduction).  Every rank renders its own disjoint batch of `--poses` poses (weak scaling,
no data-path collective).

Prints ONE JSON line on rank 0 with `roofline` (fragment kernel vs the HBM-read roofline, SURVEY 8(d):
6 algorithmic bytes read per output pixel) and `cpu_baseline` (the oracle's scalar rasteriser timed on
the host cores over a bounded sample of the same poses).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_READ_BYTES_PER_PIXEL = 6   # 4 B visibility record + 2 B atlas texel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--poses', type=int, default=1024)
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--level', type=int, default=0)
    ap.add_argument('--iwad', default=None)
    ap.add_argument('--metadata', default=None)
    ap.add_argument('--cpu-sample', type=int, default=512, help='poses rendered by the CPU oracle (0 = skip)')
    args = ap.parse_args()

    import torch
    import rust_doom_amd as rd
    from util import META_PATH, ensure_wad
    sharding = importlib.import_module('rust-doom_amd.sharding')

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # one process per GPU; RDOOM_DIST_BACKEND=gloo (with ranks wrapped onto the GPUs present) exists only so that the
    # N > 1 code path can be exercised on a one-GPU box
    backend = os.environ.get('RDOOM_DIST_BACKEND', 'nccl')
    device_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device_index))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
    rd.set_device(device_index)

    iwad = args.iwad or ensure_wad()
    meta = args.metadata or META_PATH
    wad = rd.Wad(iwad, meta)
    t0 = time.perf_counter()
    built = wad.build_level(args.level, gpu_tessellation=True)   # SSECTOR -> polygon, SEG -> quad kernels
    t_build = time.perf_counter() - t0
    level = rd.DeviceLevel(built)                      # level arrays now resident in HBM
    batch = rd.Batch(level, args.width, args.height, args.poses)
    poses = sharding.pose_sweep(rd, built, args.poses, args.width, args.height, first=rank * args.poses)
    lights = built.lights_at(0.0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.render(poses, lights, timed=True)
    barrier()
    t_start = time.perf_counter()
    frag_ms, raster_ms, setup_ms, vis_tris, fixups = [], [], [], 0, 0
    for _ in range(args.steps):
        t = batch.render(poses, lights, timed=True)    # hipEvents on the render stream, per kernel
        frag_ms.append(t['fragment_ms'])
        raster_ms.append(t['raster_ms'])
        setup_ms.append(t['setup_ms'])
        vis_tris = t['visible_triangles']
        fixups = t['fixup_pixels']
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = sharding.max_over_ranks(elapsed, dist, 'cuda' if backend == 'nccl' else 'cpu')

    if rank == 0:
        px_per_step = args.poses * args.width * args.height
        total_px = px_per_step * args.steps * world
        frag = float(np.mean(frag_ms))
        achieved = px_per_step * ALG_READ_BYTES_PER_PIXEL / (frag * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_fragment_latest.json')
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))  # PMC passes of an earlier profile run (tools/profile_round.sh) on this workload
                if (rec.get('poses'), rec.get('width'), rec.get('height')) == (args.poses, args.width, args.height):
                    traffic = rec.get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        cpu = None
        if args.cpu_sample > 0 and world == 1:  # the CPU baseline is a rank-0, N = 1 measurement
            from oracle import raster
            ro = raster.RasterOracle(built.arrays())
            cores = os.cpu_count() or 1
            n = min(args.cpu_sample, args.poses)
            sample = np.zeros((n, 33), np.float32)
            sample[:, :16] = poses['modelview'][:n]
            sample[:, 16:32] = poses['projection'][:n]
            sample[:, 32] = poses['time'][:n]
            tc = time.perf_counter()
            ro.render_batch(sample, np.tile(lights, (n, 1)), args.width, args.height, threads=cores)
            tc = time.perf_counter() - tc
            cpu = {'value': round(n * args.width * args.height / tc / 1e6, 3), 'unit': 'Mpixels/s',
                   'cores': min(cores, n), 'kind': 'port',
                   'sample': '%d poses of the same sweep at %dx%d, oracle/raster_oracle.c, %.1f s; level build '
                             '(C++ walk + device tessellation kernels, first use) %.1f ms' % (n, args.width, args.height, tc, t_build * 1e3)}
        out = {
            'metric': 'Mpixels/s, E1M1 1920x1080 pose batch', 'value': round(total_px / elapsed / 1e6, 1),
            'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32+u8', 'data': 'synthetic',
            'frames_per_s': round(args.poses * args.steps * world / elapsed, 1),
            'config': {'workload': '%s %d-pose sweep at %dx%d per GPU, walls+flats+decor+sky'
                                   % ('E1M1 (synthetic IWAD, tools/mkwad.py)' if args.iwad is None and args.level == 0 else
                                      'level %d of %s' % (args.level, os.path.basename(iwad)), args.poses, args.width, args.height),
                       'poses_per_gpu': args.poses, 'width': args.width, 'height': args.height,
                       'visible_triangles_per_pose': round(vis_tris / args.poses, 1),
                       'alpha_leak_fixup_pixels_per_step': fixups,
                       'kernels_ms': {'setup': round(float(np.mean(setup_ms)), 3),
                                      'raster': round(float(np.mean(raster_ms)), 3), 'fragment': round(frag, 3)},
                       'parallelism': 'pose-sharded x%d, no collective' % world},
            'roofline': {'bound': 'hbm', 'kernel': 'fragment_kernel', 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic},
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

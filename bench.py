#!/usr/bin/env python3
"""bench.py -- pose-batch throughput of the MI355X renderer (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic camera poses per level:
  H2D of the pose constants -> setup kernel -> binning -> tiled rasteriser -> fragment kernel,
for `--poses` poses (default 1024) at `--width`x`--height` (default 1920x1080) of level E1M1 of the synthetic IWAD
(no DOOM1.WAD exists here; pass --iwad/--metadata to use a real one).  Level arrays are resident in HBM before the
timed region; framebuffers stay on the device (D2H is not part of the metric).

Workloads (BASELINE.json configs):
  default                         config 3: E1M1, 1024-pose sweep, 1920x1080, 1 GPU
  --levels 0-8 [--scaling strong] config 4: E1M1..E1M9, one batch per level, pose ranges [g n/G, (g+1) n/G) per GPU
  --big --width 3840 --height 2160 --time-varying
                                  config 5 class: the 10x-E1M1 stand-in for MAP29 at 4K, pose i at time i/35 s with its
                                  own light table (animated flats, scrolling walls, sector light effects)

Multi-GPU: one process per GPU, no data-path collective (poses are independent; torch.distributed is used ONLY for the
timing barrier and the max-over-ranks reduction).  `--gpus N` launched WITHOUT a torch.distributed environment spawns
the N ranks itself (python -m torch.distributed.run, 127.0.0.1); under torchrun it is one rank of WORLD_SIZE.
The rank's renders are spread over a small pool of HIP streams (`--streams`, default: auto): with ONE level its poses are cut
into S sub-batches, one per stream (auto: 3 -- one sub-batch's rasteriser overlaps another's fragment kernel); with several levels
(config 4) the levels ALTERNATE over the pool, each as one batch (auto: 3) -- one level's latency-bound set-up and binning
kernels run under another's rasteriser and fragment kernel, which is what keeps a small share of the batch (strong scaling:
128 poses per level and GPU at 8 GPUs) near the full batch's rate.  `kernels_ms` and `roofline` then come from a single-stream
pass of the same steps after the timed region (with `--streams 1` from the timed region itself).
`--share G` adds the strong-scaling proxy a one-GPU box can give: the same workload with poses / G per level (the share of one
of G GPUs), timed the same way; `scaling_proxy.predicted_speedup_at_G` = full step time / share step time (a PREDICTION: the
other G - 1 GPUs are assumed to do as well on their shares; nothing is exchanged between them).
`--scaling weak` (default): every rank renders its own `--poses` poses; `--scaling strong`: ONE batch of `--poses`
poses is cut into contiguous ranges.  When fewer GPUs are present than ranks (a one-GPU box), ranks wrap onto the GPUs
present over gloo -- that exercises the code path, it is not a scaling measurement, and the line says so
(`gpus_present`).

Prints ONE JSON line on rank 0 with `roofline` (fragment kernel vs the HBM-read roofline, SURVEY 8(d): 6 algorithmic
bytes read per output pixel; plus the actual-bytes fraction from the PMC counters when they were collected for exactly
this workload and these kernel sources) and `cpu_baseline` (the oracle's scalar rasteriser timed on the host cores over a
bounded sample of the same poses).
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_READ_BYTES_PER_PIXEL = 6   # 4 B visibility record + 2 B atlas texel
LAYOUT_READ_BYTES_PER_PIXEL = 4  # what this layout stores per pixel: a 16-bit visibility word + a 16-bit texel


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--poses', type=int, default=1024, help='poses per GPU (weak) or in total (strong), per level')
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--level', type=int, default=0)
    ap.add_argument('--levels', default=None, help='level sweep, e.g. 0-8 or 0,3,5 (BASELINE config 4)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--time-varying', action='store_true', help='pose i at time i/35 s with fill_buffer_at(time) lights')
    ap.add_argument('--big', action='store_true', help='the 10x-E1M1 synthetic level (stand-in for DOOM2 MAP29)')
    ap.add_argument('--iwad', default=None)
    ap.add_argument('--metadata', default=None)
    ap.add_argument('--streams', type=int, default=0,
                    help='HIP streams the rank\'s renders are spread over (0 = auto: 3 -- one level: its poses as three sub-batches, '
                         'one sub-batch\'s rasteriser overlaps another\'s fragment kernel; several levels alternate over '
                         'the pool as one batch each).  With S > 1 the kernels overlap, so their own durations -- kernels_ms, the '
                         'roofline -- are measured in a single-stream pass of the same K steps AFTER the timed region; S = 1 '
                         'measures them in the timed region itself')
    ap.add_argument('--share', type=int, default=0, metavar='G',
                    help='strong-scaling proxy on ONE GPU: after the headline the same workload is timed with poses / G per level '
                         '(what one of G GPUs would render under --scaling strong); prints scaling_proxy')
    ap.add_argument('--long', type=float, default=0.5, metavar='SECONDS',
                    help='raise the number of timed steps until the timed region lasts at least this long (from a calibration pass of --steps '
                         'steps; e.g. --long 2: an external sampler then sees the GPU busy); the line\'s `steps` is the number really timed, '
                         '`steps_requested` what --steps said.  Default 0.5 (twenty 4 ms steps are too thin a region); 0: exactly --steps')
    ap.add_argument('--per-level-batches', action='store_true',
                    help='several levels (--levels): one batch per level alternating over the streams (round 5\'s way) instead of ONE level set '
                         'whose mixed poses are cut into one sub-batch per stream')
    ap.add_argument('--rich', action='store_true', help='the texture-rich E1M1 stand-in (tools/mkwad.py --rich: every wall and flat its own texture, '
                                                        'a texel store of several MB -- beyond one XCD\'s L2)')
    ap.add_argument('--no-rich-line', dest='rich_line', action='store_false', help='other_workloads: skip the texture-rich stand-in')
    ap.add_argument('--launcher', choices=('processes', 'threads'), default='processes',
                    help='--gpus N > 1: one PROCESS per GPU (torch.distributed.run; RCCL for the timing barrier only), or one process '
                         'with one host THREAD per GPU driving the C ABI directly (what INTEGRATION.md tells a Rust host to do; '
                         'threads wrap onto the GPUs present)')
    ap.add_argument('--cpu-sample', type=int, default=512, help='poses rendered by the CPU oracle (0 = skip)')
    ap.add_argument('--debug', action='append', default=[], metavar='NAME=VALUE',
                    help='experiment: rdoom_debug_set hook (an equivalent path: same image), e.g. frag_bw=2; repeatable')
    ap.add_argument('--other', choices=('auto', 'on', 'off'), default='auto',
                    help='short passes over BASELINE configs 2 / 4 / 5 after the headline (`other_workloads` on the JSON line); '
                         'auto: with the default workload on one GPU')
    ap.add_argument('--dry-run', action='store_true',
                    help='everything but the device work (rank launch, pose partition, barrier): for hosts without a GPU; prints no value')
    return ap.parse_args(argv)


def level_list(args):
    if args.levels is None:
        return [args.level]
    out = []
    for part in args.levels.split(','):
        if '-' in part:
            lo, hi = part.split('-')
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def kernel_source_digest():
    """sha256 over the device sources: PMC traffic figures are only quoted for the kernels they were measured on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'rust-doom_amd', 'csrc', 'hip')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.hpp')):
            with open(os.path.join(d, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def workload_key(args, levels):
    """names the WORKLOAD (level, frame, poses): PMC records are keyed on it.  How it was timed is `measurement_key`."""
    return '%s|levels=%s|%dx%d|poses=%d|tv=%d' % ('rich' if args.rich else 'big' if args.big else (os.path.basename(args.iwad) if args.iwad else 'synth'),
                                                   ','.join(map(str, levels)), args.width, args.height, args.poses,
                                                   int(args.time_varying))


def measurement_key(args, levels):
    """workload + how `value` was timed: round-over-round comparisons must not mix single-stream values (rounds 1-2, and
    `single_stream.value` since) with the overlapped default (`value` since round 3)"""
    return '%s|streams=%d' % (workload_key(args, levels), args.streams)


def resolve_streams(requested, n_levels):
    """--streams 0 = auto: three -- three sub-batches of the one level's poses, or three streams that several levels alternate over
    (measured on one box, round 5: 1080p 474 / 487 / 422 Gpixel/s on 2 / 3 / 4 streams; 4K 572 / 588; the 10x level 314 / 321)"""
    del n_levels
    return requested if requested > 0 else 3


def parts_per_level(n_levels, streams):
    """sub-batches a level's poses are cut into: S for one level; with several levels each is ONE batch (they alternate over the
    stream pool) unless there are fewer levels than streams"""
    if n_levels == 1:
        return streams
    return 1 if n_levels >= streams else -(-streams // n_levels)


def merge_path_stats(batches):
    """rdoom_batch_path_stats of the batches' last renders, summed: which paths the workload took"""
    tot = {}
    for b in batches:
        for k, v in b.path_stats().items():
            tot[k] = tot.get(k, 0) + v
    if not tot or not tot.get('tiles'):
        return None
    return {'bins_overflowed_poses': tot['bins_overflowed_poses'], 'poses': tot['poses'],
            'split_tiles_pct': round(100.0 * tot['split_tiles'] / tot['tiles'], 2),
            'entries_per_tile': round(tot['tile_entries'] / tot['tiles'], 1),
            'described_quadrants_pct': round(100.0 * tot['described_quadrants'] / max(1, tot['quadrants']), 1)}


OTHER_WORKLOADS = (   # BASELINE configs 2 / 4 / 5 (and the 4K frame) in short: what the driver's one command would otherwise never see
    dict(name='config 4 on one GPU: E1M1..E1M9 as ONE level set, poses of all levels in every render', levels=list(range(9)), big=False, width=1920, height=1080, poses=1024, tv=False, share=8),
    dict(name='config 5 class: 10x-E1M1 (MAP29 stand-in), pose i at time i/35 s with its own light table', levels=[0], big=True, width=3840, height=2160, poses=256, tv=True, parity=4),
    # (small frames: the set-up kernels -- a third of that step, latency-bound -- overlap better over three sub-batches than two)
    dict(name='config 2 frame size: E1M1', levels=[0], big=False, width=320, height=200, poses=8192, tv=False, streams=3),
    dict(name='E1M1', levels=[0], big=False, width=3840, height=2160, poses=256, tv=False),
)


def build_renders(rd, sharding, wad, levels, lo, hi, width, height, tv, handles, merged=True):
    """The renders of ONE step of one rank: poses [lo, hi) of the seeded sweep of every level in `levels`, spread over the HIP
    streams `handles` (raw hipStream_t values; [None] = the default stream).
      one level        its poses as S = len(handles) sub-batches, one per stream;
      several, merged  ONE level set (rdoom_levelset_create), every level's poses concatenated level after level and cut into S
                       contiguous sub-batches of mixed poses, one per stream -- S launch sets per step whatever the number of levels;
      several, not     one batch per level, the levels alternating over the streams (round 5's way; --per-level-batches).
    Returns (items, full, closers, builts): items = [(batch, poses, lights, level_of_pose or None, stream)];
    full = the same poses for the per-kernel pass on the default stream (one render per kernel set where that fits: one level ->
    ONE batch of all its poses; a level set -> the same sub-batches one after the other)."""
    S = max(1, len(handles))
    n = hi - lo
    builts = [wad.build_level(i, gpu_tessellation=True) for i in levels]   # SSECTOR -> polygon, SEG -> quad kernels
    closers, items, full = [], [], []

    def sweep(built):
        poses = sharding.pose_sweep(rd, built, n, width, height, first=lo)
        if tv:
            poses['time'] = (np.arange(lo, hi) / 35.0).astype(np.float32)
            lights = np.stack([built.lights_at(float(t)) for t in poses['time']]) if n else np.zeros((0, 256), np.uint8)
        else:
            lights = built.lights_at(0.0)
        return poses, lights

    if len(levels) > 1 and merged:
        lset = rd.DeviceLevelSet(builts)                   # every level's arrays resident in HBM, one handle
        per = [sweep(b) for b in builts]
        poses = np.concatenate([p for p, _l in per])
        lights = np.concatenate([l if l.ndim == 2 else np.tile(l, (n, 1)) for _p, l in per])
        lop = np.repeat(np.arange(len(levels), dtype=np.uint32), n)
        for part in range(S):
            plo, phi = sharding.shard_range(len(poses), part, S)
            if phi > plo:
                b = rd.Batch(lset, width, height, phi - plo)
                items.append((b, poses[plo:phi], lights[plo:phi], lop[plo:phi], handles[part % S]))
                closers.append(b)
        full = [it[:4] + (None,) for it in items]          # (the same batches on the default stream: a barrier separates the passes)
        closers.append(lset)
    else:
        parts = parts_per_level(len(levels), S)
        for li, built in enumerate(builts):
            level = rd.DeviceLevel(built)                  # level arrays now resident in HBM
            poses, lights = sweep(built)
            mine = []
            for part in range(parts):
                plo, phi = sharding.shard_range(n, part, parts)
                if phi > plo or n == 0:
                    b = rd.Batch(level, width, height, max(phi - plo, 1))
                    mine.append((b, poses[plo:phi], lights[plo:phi] if lights.ndim == 2 else lights, None, handles[(li * parts + part) % S]))
                    closers.append(b)
            items += mine
            if len(mine) == 1:
                full.append(mine[0][:4] + (None,))         # (the same batch on the default stream)
            else:
                b = rd.Batch(level, width, height, max(n, 1))
                full.append((b, poses, lights, None, None))
                closers.append(b)
            closers.append(level)
    return items, full, closers + builts, builts


def run_renders(items, k, profiled):
    for _ in range(k):
        for b, p, l, lop, st in items:
            if len(p):
                (b.render_profiled if profiled else b.render)(p, l, stream=st, level_of_pose=lop)


def collect_kernels(items, acc=None):
    """rdoom_batch_collect_timings of every batch: sums of the pending profiled renders' per-kernel times"""
    for b in {id(it[0]): it[0] for it in items}.values():
        t = b.collect_timings()
        if acc is not None and t['renders']:
            for k in ('setup_ms', 'raster_ms', 'fragment_ms'):
                acc[k] += t[k]
            acc['visible_triangles'] = acc.get('visible_triangles', 0) + t['visible_triangles'] * t['renders']
            acc['fixup_pixels'] = acc.get('fixup_pixels', 0) + t['fixup_pixels'] * t['renders']
    return acc


def texel_store_bytes(builts):
    """bytes of the u16 texel store the fragment kernel gathers from (wall + flat + decor atlases of every level resident)"""
    total = 0
    for b in builts:
        d = b.desc
        total += 2 * (d.wall_w * d.wall_h + d.flat_w * d.flat_h + d.decor_w * d.decor_h)
    return int(total)


def parity_stamp(rd, built, batch, poses, lights, first, count, width, height, threads):
    """HIP == oracle on `count` of the poses this bench just timed: the oracle's frames (test infrastructure, imported here as the
    CHECKER and the cpu_baseline only) against rdoom_batch_read_framebuffer of the same poses of the batch's LAST render.
    Returns (stamp dict, oracle seconds)."""
    from oracle import raster
    ro = raster.RasterOracle(built.arrays())
    n = min(count, len(poses) - first)
    sample = np.zeros((n, 33), np.float32)
    sample[:, :16] = poses['modelview'][first:first + n]
    sample[:, 16:32] = poses['projection'][first:first + n]
    sample[:, 32] = poses['time'][first:first + n]
    li = lights[first:first + n] if lights.ndim == 2 else np.tile(lights, (n, 1))
    got = batch.read_framebuffer(first, n)
    t0 = time.perf_counter()
    want = ro.render_batch(sample, li, width, height, threads=threads)
    secs = time.perf_counter() - t0
    bad = int((want != got).sum())
    return {'poses': n, 'pixels': int(want.size), 'mismatching_pixels': bad,
            'what': 'frames of the timed poses read back with rdoom_batch_read_framebuffer == oracle/raster_oracle.c, byte for byte'}, secs, ro, sample, li


def quick_line(rd, torch, sharding, wad, spec, streams, steps=12, warmup=3, kernels=True, rank=0, world=1, dist=None, backend_device='cpu', merged=True):
    """One short measurement of another workload, timed like the headline: the renders of a step spread over `streams` HIP
    streams (value), then the per-kernel pass on one stream (kernels=False: skipped).  world > 1: --scaling strong -- every rank
    renders poses [g n/G, (g+1) n/G) of every level, barriers on both sides, the slowest rank's time counts."""
    w, h, n = spec['width'], spec['height'], spec['poses']
    n_levels = len(spec['levels'])
    lo, hi = sharding.shard_range(n, rank, world)
    pool = [torch.cuda.Stream() for _ in range(streams)] if streams > 1 else []
    handles = [st.cuda_stream for st in pool] if pool else [None]
    items, full, closers, builts = build_renders(rd, sharding, wad, spec['levels'], lo, hi, w, h, spec['tv'], handles, merged=merged)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # (with several streams the per-kernel events would only add barriers between the kernels of a stream: plain renders)
    run_renders(items, warmup, streams == 1)
    collect_kernels(items)
    barrier()
    t0 = time.perf_counter()
    run_renders(items, steps, streams == 1)
    barrier()
    mine = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(mine, dist, backend_device)
    acc = collect_kernels(items, {'setup_ms': 0.0, 'raster_ms': 0.0, 'fragment_ms': 0.0})
    if streams > 1 and kernels:
        run_renders(full, warmup, True)
        collect_kernels(full)
        barrier()
        run_renders(full, steps, True)
        barrier()
        acc = collect_kernels(full, {'setup_ms': 0.0, 'raster_ms': 0.0, 'fragment_ms': 0.0})
    px = n * w * h * n_levels
    my_px = (hi - lo) * w * h * n_levels
    if n_levels == 1:
        plan = '%d sub-batches of the level\'s poses, one per stream' % len(items)
    elif merged:
        plan = 'ONE level set: the %d levels\' poses, level after level, cut into %d sub-batches of mixed poses, one per stream' % (n_levels, len(items))
    else:
        plan = 'the %d levels alternate over the %d streams, one batch each' % (n_levels, streams)
    out = {'workload': '%s, %d poses%s at %dx%d' % (spec['name'], n, ' per level' if n_levels > 1 else '', w, h),
           'value': round(px * steps / elapsed / 1e6, 1), 'unit': 'Mpixels/s', 'ms_per_step': round(elapsed / steps * 1e3, 3), 'steps': steps, 'streams': streams,
           'stream_plan': plan, 'texel_store_bytes': texel_store_bytes(builts)}
    if world > 1:
        out['poses_per_level_and_rank'] = hi - lo
        out['rank_ms_per_step'] = round(mine / steps * 1e3, 3)
    if streams == 1 or kernels:
        frag = acc['fragment_ms'] / steps
        out['kernels_ms'] = {k[:-3]: round(acc[k] / steps, 3) for k in ('setup_ms', 'raster_ms', 'fragment_ms')}
        out['roofline_frac'] = round(my_px * ALG_READ_BYTES_PER_PIXEL / (frag * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if frag > 0 else None
    out['paths'] = merge_path_stats([it[0] for it in items])
    if spec.get('parity') and world == 1:   # a few of the timed poses against the oracle, on the line itself
        b, p, l, lop, _st = items[0]
        stamp, _secs, _ro, _s, _l = parity_stamp(rd, builts[int(lop[0]) if lop is not None else 0], b, p, l, 0, spec['parity'], w, h, min(spec['parity'], usable_cores()))
        out['parity'] = stamp
    for c in closers:
        c.close()
    return out


def scaling_proxy(rd, torch, sharding, wad, spec, G, steps=20, warmup=3):
    """The strong-scaling share a one-GPU box can measure: the workload of `spec` with poses / G per level -- what ONE of G GPUs renders
    under --scaling strong (its contiguous range of every level's batch) -- next to the full batch, BOTH timed on 1, 2 and 3 streams
    (the same code, the same number of steps) and the best of each taken: predicted_speedup_at_G = best full step time / best share
    step time -- the speed-up G GPUs would give IF each did as well on its share as this one (nothing is exchanged between them:
    DESIGN section 6).  A prediction, not a scaling measurement."""
    share = dict(spec, poses=max(1, spec['poses'] // G), name=spec['name'] + ' -- the 1/%d share' % G)
    fulls, shares = {}, {}
    for s in (1, 2, 3):
        shares[s] = quick_line(rd, torch, sharding, wad, share, s, steps=steps, warmup=warmup, kernels=(s == 1))
    for s in (2, 3):   # (one stream is never the full batch's best: DESIGN section 5; its 37 ms steps are not spent a third time)
        fulls[s] = quick_line(rd, torch, sharding, wad, spec, s, steps=steps, warmup=warmup, kernels=False)
    bs, bf = min(shares, key=lambda k: shares[k]['ms_per_step']), min(fulls, key=lambda k: fulls[k]['ms_per_step'])
    return {'gpus': G, 'poses_per_level_full': spec['poses'], 'poses_per_level_share': share['poses'],
            'full_ms_by_streams': {str(k): v['ms_per_step'] for k, v in fulls.items()}, 'full_ms': fulls[bf]['ms_per_step'], 'full_streams': bf,
            'share_ms_by_streams': {str(k): v['ms_per_step'] for k, v in shares.items()}, 'share_ms': shares[bs]['ms_per_step'],
            'share_streams': bs, 'share_kernels_ms_one_stream': shares[1].get('kernels_ms'),
            'predicted_speedup_at_%d' % G: round(fulls[bf]['ms_per_step'] / shares[bs]['ms_per_step'], 2),
            'ideal_share_ms': round(fulls[bf]['ms_per_step'] / G, 3),
            'what': 'one GPU, the same box and library: full = %d poses per level, share = %d (the contiguous range one of %d GPUs renders under '
                    '--scaling strong), each the best of the stream counts tried, %d steps each; a prediction from one GPU, not a scaling measurement' % (
                        spec['poses'], share['poses'], G, steps)}


def usable_cores():
    """host threads this process may really run at once: the scheduler affinity, capped by the cgroup's CPU quota (a
    container on a 256-thread box may be allowed a handful of them; os.cpu_count() would still say 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def metric_label(args, levels):
    """BASELINE.json's metric for the default workload; for the others the same quantity with the workload named
    (`config.workload` stays the authoritative description)"""
    if args.iwad:
        what = '%s level(s) %s' % (os.path.basename(args.iwad), ','.join(map(str, levels)))
    elif args.rich:
        what = 'texture-rich E1M1'
    elif args.big:
        what = '10x-E1M1'
    elif len(levels) == 1:
        what = 'E1M%d' % (levels[0] + 1)
    else:
        what = 'E1M%d-E1M%d' % (levels[0] + 1, levels[-1] + 1) if levels == list(range(levels[0], levels[-1] + 1)) else 'E1M' + ','.join(str(i + 1) for i in levels)
    return 'Mpixels/s, %s %dx%d pose batch%s' % (what, args.width, args.height, ', time-varying' if args.time_varying else '')


def spawn_ranks(args):
    """`python bench.py --gpus N` without a torch.distributed environment: launch the N ranks (one process per GPU)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world, rd, sharding, synthetic):
    """No device work: launches, partitions the sweep exactly as the real run does, meets at the barrier, and reports
    which pose range every rank would have rendered.  There is no CPU renderer in the product -- no `value`."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    levels = level_list(args)
    lo, hi = sharding.shard_range(args.poses, rank, world) if args.scaling == 'strong' else (rank * args.poses, (rank + 1) * args.poses)
    built = rd.Wad(args.iwad or (synthetic.ensure_big_wad() if args.big else synthetic.ensure_wad()),
                   args.metadata or synthetic.META_PATH).build_level(levels[0])
    poses = sharding.pose_sweep(rd, built, hi - lo, args.width, args.height, first=lo)
    ranges = [None] * world
    if world > 1:
        dist.all_gather_object(ranges, (lo, hi, hashlib.sha256(poses.tobytes()).hexdigest()[:16]))
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranges = [(lo, hi, hashlib.sha256(poses.tobytes()).hexdigest()[:16])]
    if rank == 0:
        print(json.dumps({'dry_run': True, 'value': None, 'n_gpus': world, 'scaling': args.scaling, 'levels': levels,
                          'pose_ranges': [list(r) for r in ranges]}), flush=True)


def cpu_baseline_and_parity(rd, args, built, batch, poses, lights, iwad, meta, level_index):
    """rank 0, N = 1: the oracle's scalar rasteriser (`kind: port`) on a bounded sample of the timed poses, on the host cores --
    and, because its frames are THE expected output, the parity stamp of the same poses (parity_stamp)."""
    frame_px = args.width * args.height
    cores = usable_cores()
    n = min(args.cpu_sample, len(poses))
    stamp, tc, ro, sample, li = parity_stamp(rd, built, batch, poses, lights, 0, n, args.width, args.height, cores)
    # the same scalar loop on ONE thread (a few poses: about a second per megapixel), and the CPU-only geometry
    # build of BASELINE config 1 timed per phase on one thread (tools/dump_geometry.py: t_load = rows a1-a7,
    # t_walk = rows a9-a15 of SURVEY 8(a); medians of 9 runs, no GPU involved)
    n1 = min(n, max(1, int(round(40.0e6 / frame_px))))
    t1 = time.perf_counter()
    ro.render_batch(sample[:n1], li[:n1], args.width, args.height, threads=1)
    t1 = time.perf_counter() - t1
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import dump_geometry
    host = dump_geometry.host_timings(iwad, meta, level_index)
    cpu = {'value': round(n * frame_px / tc / 1e6, 3), 'unit': 'Mpixels/s',
           'cores': min(cores, n), 'kind': 'port',
           'sample': '%d poses of the same sweep at %dx%d, oracle/raster_oracle.c, %.1f s on %d threads; single thread: '
                     '%d poses in %.1f s (os.cpu_count() = %d); geometry build: CPU-only C++ path (use_gpu_tessellation = 0), one thread, '
                     'median of %d runs' % (n, args.width, args.height, tc, min(cores, n), n1, t1, os.cpu_count() or 1, host['repeat']),
           't_raster_1': {'value': round(n1 * frame_px / t1 / 1e6, 3), 'unit': 'Mpixels/s', 'cores': 1, 'poses': n1},
           't_load_ms': host['t_load_ms'], 't_walk_ms': host['t_walk_ms'], 'geometry_phases_ms': host['phases_ms'],
           'geometry_first_run_ms': host['first_run_ms']}
    return cpu, stamp


def run_threads(args):
    """--launcher threads: ONE process, one host thread per GPU, the C ABI only (no torch, no torch.distributed) -- the shape
    INTEGRATION.md gives a Rust host: a `std::thread` per device, each with its own rdoom_wad handle (the reference's Archive is
    !Sync), its own level copy, batches and streams; a barrier on both sides of the timed region, the slowest thread's time
    counts.  With fewer GPUs than threads the threads wrap onto the GPUs present (exercises the path; the line says so).
    Thread 0 then measures what the process launcher's rank 0 measures: the per-kernel pass on one stream (roofline) and, at
    N = 1, the CPU baseline with the parity stamp.  --dry-run: everything but the device work."""
    import ctypes
    import threading
    import rust_doom_amd as rd
    sharding = importlib.import_module('rust-doom_amd.sharding')
    synthetic = importlib.import_module('rust-doom_amd.synthetic')
    world, levels = args.gpus, level_list(args)
    iwad = args.iwad or (synthetic.ensure_big_wad() if args.big else synthetic.ensure_wad())
    meta = args.metadata or synthetic.META_PATH
    present = 0 if args.dry_run else rd.device_count()
    if not args.dry_run and present < 1:
        raise SystemExit('bench.py needs a GPU: there is no CPU fallback (--dry-run checks the launch and the partition only)')
    hip = None if args.dry_run else ctypes.CDLL('libamdhip64.so')
    for item in args.debug:
        name, _, value = item.partition('=')
        rd.debug_set(name, int(value or 1))
    barrier = threading.Barrier(world)
    elapsed, ranges, errors = [0.0] * world, [None] * world, []
    extras = {}

    def worker(rank):
        try:
            lo, hi = sharding.shard_range(args.poses, rank, world) if args.scaling == 'strong' else (rank * args.poses, (rank + 1) * args.poses)
            wad = rd.Wad(iwad, meta)                 # one handle per thread
            if args.dry_run:
                built = wad.build_level(levels[0])
                poses = sharding.pose_sweep(rd, built, hi - lo, args.width, args.height, first=lo)
                ranges[rank] = (lo, hi, hashlib.sha256(poses.tobytes()).hexdigest()[:16])
                barrier.wait()
                return
            rd.set_device(rank % present)             # hipSetDevice is per host thread
            pool = []
            for _ in range(args.streams):
                st = ctypes.c_void_p()
                if hip.hipStreamCreate(ctypes.byref(st)) != 0:
                    raise RuntimeError('hipStreamCreate failed')
                pool.append(st)
            # one copy of the level(s) per device (replicated: SURVEY 8(e)); several levels = one level set
            items, full, closers, builts = build_renders(rd, sharding, wad, levels, lo, hi, args.width, args.height, args.time_varying,
                                                         [st.value for st in pool], merged=not args.per_level_batches)
            ranges[rank] = (lo, hi)

            def steps(k, which=items, profiled=False):
                run_renders(which, k, profiled)
                for b in {id(it[0]): it[0] for it in which}.values():
                    b.finish()                        # waits for THIS batch's stream only; raises if the device flagged a problem

            steps(args.warmup)
            barrier.wait()
            t0 = time.perf_counter()
            steps(args.steps)
            elapsed[rank] = time.perf_counter() - t0
            barrier.wait()
            if rank == 0:   # what rank 0 of the process launcher measures after the timed region
                steps(args.warmup, full, True)
                collect_kernels(full)
                steps(args.steps, full, True)
                extras['acc'] = collect_kernels(full, {'setup_ms': 0.0, 'raster_ms': 0.0, 'fragment_ms': 0.0})
                extras['my_px_per_step'] = (hi - lo) * args.width * args.height * len(levels)
                extras['paths'] = merge_path_stats([it[0] for it in full])
                if world == 1 and args.cpu_sample > 0:
                    b, p, l, lop, _st = full[0]
                    n0 = len(p) if lop is None else int((lop == lop[0]).cumprod().sum())   # (the leading poses of one level)
                    extras['cpu'], extras['parity'] = cpu_baseline_and_parity(rd, args, builts[int(lop[0]) if lop is not None else 0], b, p[:n0],
                                                                              l[:n0] if l.ndim == 2 else l, iwad, meta, levels[0])
            barrier.wait()
            for c in closers:
                c.close()
            for st in pool:
                hip.hipStreamDestroy(st)
        except Exception as e:  # noqa: BLE001  (reported by the main thread)
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise SystemExit('bench.py --launcher threads: %r' % errors)
    if args.dry_run:
        print(json.dumps({'dry_run': True, 'value': None, 'n_gpus': world, 'launcher': 'threads', 'scaling': args.scaling, 'levels': levels,
                          'pose_ranges': [list(r) for r in ranges]}), flush=True)
        return
    t = max(elapsed)
    frame_px = args.width * args.height
    poses_global = (args.poses if args.scaling == 'strong' else args.poses * world) * len(levels)
    acc = extras['acc']
    frag = acc['fragment_ms'] / args.steps
    achieved = extras['my_px_per_step'] * ALG_READ_BYTES_PER_PIXEL / (frag * 1e-3) / 1e9 if frag > 0 else 0.0
    out = {'metric': metric_label(args, levels), 'value': round(poses_global * frame_px * args.steps / t / 1e6, 1), 'unit': 'Mpixels/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(t / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling,
           'vs_baseline': None, 'dtype': 'f32+u8 (triangle set-up f64)', 'data': 'synthetic', 'frames_per_s': round(poses_global * args.steps / t, 1),
           'config': {'workload': '%s, %d poses %s at %dx%d' % (metric_label(args, levels), args.poses, 'per GPU' if args.scaling == 'weak' else 'in total, cut into contiguous ranges',
                                                                 args.width, args.height),
                      'levels': levels, 'launcher': 'threads: one process, one host thread per GPU, C ABI only', 'streams': args.streams,
                      'kernels_ms': {k[:-3]: round(acc[k] / args.steps, 3) for k in ('setup_ms', 'raster_ms', 'fragment_ms')},
                      'kernels_ms_from': 'thread 0, a single-stream pass after the timed region',
                      'paths': extras['paths'],
                      'pose_ranges': [list(r) for r in ranges], 'parallelism': 'pose-sharded x%d, no collective' % world,
                      'kernel_sources': kernel_source_digest()},
           'per_thread_ms_per_step': [round(e / args.steps * 1e3, 3) for e in elapsed],
           'roofline': {'bound': 'hbm', 'kernel': 'fragment_kernel', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None, 'measured_in': 'thread 0, single-stream pass after the timed region'},
           'cpu_baseline': extras.get('cpu'), 'parity': extras.get('parity'),
           'note_launcher': 'cpu_baseline / parity are N = 1 measurements (thread 0); roofline: thread 0\'s fragment kernel'}
    if present < world:
        out['gpus_present'] = present
        out['note'] = '%d threads wrapped onto %d GPU(s): exercises the N > 1 path, not a scaling measurement' % (world, present)
    print(json.dumps(out), flush=True)
    if out['parity'] and out['parity']['mismatching_pixels']:
        raise SystemExit('bench.py: the timed frames differ from the oracle in %d pixels' % out['parity']['mismatching_pixels'])


def main():
    args = parse_args()
    cli_streams = args.streams
    args.streams = resolve_streams(args.streams, len(level_list(args)))
    if args.launcher == 'threads':
        return run_threads(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import rust_doom_amd as rd
    sharding = importlib.import_module('rust-doom_amd.sharding')
    synthetic = importlib.import_module('rust-doom_amd.synthetic')

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.dry_run:
        return dry_run(args, rank, world, rd, sharding, synthetic)
    present = torch.cuda.device_count()
    if present < 1:
        raise SystemExit('bench.py needs a GPU: there is no CPU fallback (--dry-run checks the launch and the partition only)')
    # one process per GPU over RCCL; with fewer GPUs than ranks the ranks wrap onto the GPUs present and the timing
    # barrier goes over gloo (RCCL refuses two ranks on one device)
    backend = os.environ.get('RDOOM_DIST_BACKEND', 'nccl' if present >= world else 'gloo')
    device_index = local_rank % present
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
    for item in args.debug:
        name, _, value = item.partition('=')
        rd.debug_set(name, int(value or 1))

    iwad = args.iwad or (synthetic.ensure_rich_wad() if args.rich else (synthetic.ensure_big_wad() if args.big else synthetic.ensure_wad()))
    meta = args.metadata or synthetic.META_PATH
    wad = rd.Wad(iwad, meta)
    levels = level_list(args)
    dist_device = 'cuda' if backend == 'nccl' else 'cpu'
    if args.scaling == 'strong':
        lo, hi = sharding.shard_range(args.poses, rank, world)   # contiguous ranges of ONE batch
    else:
        lo, hi = rank * args.poses, (rank + 1) * args.poses      # every GPU its own batch
    n_mine = hi - lo
    # the stream pool (--streams S): one level -> its poses as S sub-batches, one per stream (one sub-batch's rasteriser overlaps
    # another's fragment kernel); several levels -> ONE level set whose mixed poses are cut into S sub-batches (build_renders)
    tstreams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else []
    merged = not args.per_level_batches
    t0 = time.perf_counter()
    work, work_full, closers, builts = build_renders(rd, sharding, wad, levels, lo, hi, args.width, args.height, args.time_varying,
                                                     [st.cuda_stream for st in tstreams] if tstreams else [None], merged=merged)
    if args.streams == 1:
        work_full = []

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Steps are queued without a host synchronisation in between (the staging of step i + 1 overlaps the kernels of
    # step i); the hipEvents around every kernel of every timed step stay pending on the render stream and are read
    # after the closing barrier (at most 64 renders per batch may be pending: collected in between if K is larger).
    # With several streams the timed region uses PLAIN renders: the four events a profiled render records are barriers between
    # the kernels of its stream, and the per-kernel times of overlapped kernels mean nothing anyway (they come from the
    # single-stream pass below).
    timed_profiled = args.streams == 1

    def step(i, items, profiled=True):
        run_renders(items, 1, profiled)
        return profiled and i % 30 == 29   # (at most 64 profiled renders may be pending per batch)

    for i in range(args.warmup):
        if step(i, work, timed_profiled):
            collect_kernels(work)
    collect_kernels(work)
    barrier()
    steps_requested = args.steps
    if args.long > 0:   # as many timed steps as fill that many seconds, from a calibration pass of K steps (every rank agrees)
        t_w = time.perf_counter()
        for i in range(args.steps):
            if step(i, work, timed_profiled):
                collect_kernels(work)
        collect_kernels(work)
        barrier()
        est = sharding.max_over_ranks((time.perf_counter() - t_w) / args.steps, dist, dist_device)
        args.steps = max(args.steps, int(np.ceil(1.05 * args.long / max(est, 1e-6))))
    # per-step spread: an event at the end of every step on every render stream (GPU time stamps; nothing waits for them)
    mark_streams = tstreams if tstreams else [torch.cuda.current_stream()]
    ev_start = torch.cuda.Event(enable_timing=True)
    # (the per-step marks exist before the timed region and the collector is off inside it: an allocation or a collection in the
    # loop stalls the host for tens of milliseconds once in a few dozen steps -- seen as ONE 40 ms step in 20 on the nine-level
    # workloads, whose renders are too short for the queue to ride such a pause out)
    step_marks = [[torch.cuda.Event(enable_timing=True) for _ in mark_streams] for _ in range(args.steps)]
    import gc
    gc.collect()
    gc.disable()
    ev_start.record(mark_streams[0])
    t_start = time.perf_counter()
    acc = {'setup_ms': 0.0, 'raster_ms': 0.0, 'fragment_ms': 0.0}
    for i in range(args.steps):
        flush = step(i, work, timed_profiled)
        for e, st in zip(step_marks[i], mark_streams):
            e.record(st)
        if flush:
            collect_kernels(work, acc)   # (a host synchronisation every 30 steps)
    barrier()
    elapsed_mine = time.perf_counter() - t_start
    gc.enable()
    collect_kernels(work, acc)
    ends = [max(ev_start.elapsed_time(e) for e in marks) for marks in step_marks]   # ms since the start mark, per step
    step_ms = [b - a for a, b in zip([0.0] + ends[:-1], ends)]
    elapsed = sharding.max_over_ranks(elapsed_mine, dist, dist_device)
    # With several streams the kernels of different sub-batches run side by side: the step time above is what the metric
    # asks for, but a kernel's own duration cannot be read off overlapped events.  The same poses are therefore rendered
    # again on ONE stream (one level: as ONE batch, one launch per kernel and step; nothing overlapping), outside the timed
    # region, and kernels_ms / roofline come from that pass -- exactly what `--streams 1` measures in its timed region.
    single_elapsed = None
    k_steps = min(args.steps, max(steps_requested, 20))   # (the per-kernel pass need not be as long as a stretched timed region)
    if args.streams > 1:
        for i in range(args.warmup):
            step(i, work_full)
        collect_kernels(work_full)
        barrier()
        t1 = time.perf_counter()
        acc = {'setup_ms': 0.0, 'raster_ms': 0.0, 'fragment_ms': 0.0}
        for i in range(k_steps):
            if step(i, work_full):
                collect_kernels(work_full, acc)
        barrier()
        single_elapsed = time.perf_counter() - t1
        collect_kernels(work_full, acc)
        single_elapsed = sharding.max_over_ranks(single_elapsed, dist, dist_device)
    else:
        k_steps = args.steps
    per_rank_ms = None
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, (round(elapsed_mine / args.steps * 1e3, 3), device_index))
        per_rank_ms = [g[0] for g in gathered]
        rank_devices = [g[1] for g in gathered]

    default_workload = (levels == [0] and not args.big and not args.rich and not args.iwad and not args.time_varying and not args.debug and
                        (args.width, args.height, args.poses) == (1920, 1080, 1024))
    frame_px = args.width * args.height
    out = None
    if rank == 0:
        poses_per_step_global = (args.poses if args.scaling == 'strong' else args.poses * world) * len(levels)
        total_px = poses_per_step_global * frame_px * args.steps
        my_px_per_step = n_mine * frame_px * len(levels)
        frag = acc['fragment_ms'] / k_steps             # this rank's fragment-kernel time per step (all levels)
        achieved = my_px_per_step * ALG_READ_BYTES_PER_PIXEL / (frag * 1e-3) / 1e9 if frag > 0 else 0.0
        traffic = frac_actual = valu = pmc_source = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_fragment_latest.json')
        if os.path.exists(pmc) and world == 1:
            try:
                rec = json.load(open(pmc))  # PMC passes of an earlier profile run (tools/profile_round.sh)
                if rec.get('workload') == workload_key(args, levels) and rec.get('kernel_sources') == kernel_source_digest():
                    traffic = rec.get('hbm_bytes_per_launch')
                    frac_actual = round(traffic / (frag * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    valu = rec.get('valu')   # the VALU-issue roofline of the hot kernels (same passes: see tools/profile_collect.py)
                    # REPLAYED, not measured by this run: counter passes cannot share a run with the timing (rocprofv3 serialises
                    # the kernels); they are quoted only while the device sources are byte-identical to the ones they were taken on
                    pmc_source = 'profiles/pmc_fragment_latest.json (replayed: rocprofv3 --pmc passes of %s on kernel sources %s, taken %s)' % (
                        rec.get('workload'), rec.get('kernel_sources'), rec.get('taken', 'by tools/profile_round.sh'))
            except Exception:
                traffic = None
        paths = merge_path_stats([it[0] for it in (work_full or work)])   # of the last renders: which paths the workload took
        # the CPU baseline (rank 0, N = 1) and, from the frames it computes anyway, the parity stamp: the oracle's frames of the
        # first poses of the timed sweep against rdoom_batch_read_framebuffer of the SAME poses (the batch still holds its last render)
        cpu = parity = None
        if args.cpu_sample > 0 and world == 1:
            b, p, l, lop, _st = (work_full or work)[0]
            n0 = len(p) if lop is None else int((lop == lop[0]).cumprod().sum())   # (the leading poses of one level)
            cpu, parity = cpu_baseline_and_parity(rd, args, builts[int(lop[0]) if lop is not None else 0], b, p[:n0], l[:n0] if l.ndim == 2 else l,
                                                  iwad, meta, levels[0])
        if args.iwad:
            what = 'level(s) %s of %s' % (','.join(map(str, levels)), os.path.basename(iwad))
        elif args.rich:
            what = 'texture-rich E1M1 stand-in (tools/mkwad.py --rich: %.1f MB texel store)' % (texel_store_bytes(builts) / 1e6)
        elif args.big:
            what = '10x-E1M1 synthetic level (MAP29 stand-in, tools/mkwad.py)'
        elif levels == [0]:
            what = 'E1M1 (synthetic IWAD, tools/mkwad.py)'
        else:
            what = 'E1M%s (synthetic IWAD, tools/mkwad.py), %s' % (','.join(str(i + 1) for i in levels), 'one level set' if merged else 'one batch per level')
        if len(levels) == 1:
            plan = '%d sub-batches of the poses, one per stream' % len(work)
        elif merged:
            plan = 'ONE level set: the %d levels\' poses, level after level, cut into %d sub-batches of mixed poses, one per stream' % (len(levels), len(work))
        else:
            plan = 'the %d levels alternate over the %d streams, one batch each' % (len(levels), args.streams)
        out = {
            'metric': metric_label(args, levels), 'value': round(total_px / elapsed / 1e6, 1),
            'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f32+u8 (triangle set-up f64)', 'data': 'synthetic',
            'frames_per_s': round(poses_per_step_global * args.steps / elapsed, 1),
            # --steps K asks for AT LEAST K timed steps; the timed region is stretched to --long seconds (default 0.5: twenty steps
            # of 4.4 ms are 88 ms, a fifth of which is the ramp of the first step) and `steps` is the number really timed
            'steps_requested': steps_requested, 'timed_seconds': round(elapsed, 3),
            'config': {'workload': '%s %d-pose sweep at %dx%d %s, walls+flats+decor+sky%s'
                                   % (what, args.poses, args.width, args.height,
                                      'per GPU' if args.scaling == 'weak' else 'in total, cut into contiguous ranges',
                                      ', pose i at time i/35 s with its own light table' if args.time_varying else ''),
                       'levels': levels, 'poses_per_gpu': n_mine, 'width': args.width, 'height': args.height,
                       'texel_store_bytes': texel_store_bytes(builts),
                       # (counts of the LAST render of each batch x renders: every step re-renders the same poses)
                       'visible_triangles_per_pose': round(acc.get('visible_triangles', 0) / max(1, k_steps * n_mine * len(levels)), 1),
                       'alpha_leak_fixup_pixels_per_step': acc.get('fixup_pixels', 0) // max(1, k_steps),
                       'kernels_ms': {k[:-3]: round(acc[k] / k_steps, 3) for k in ('setup_ms', 'raster_ms', 'fragment_ms')},
                       'parallelism': 'pose-sharded x%d, no collective' % world,
                       'kernels_ms_from': ('the timed region (one stream)' if args.streams == 1 else
                                           'a single-stream pass after the timed region (%s, %d steps): '
                                           'with %d streams the kernels overlap, so their sum exceeds ms_per_step' % (
                                               'the same poses as ONE batch, one launch per kernel and step' if len(levels) == 1 else 'the same sub-batches one after the other',
                                               k_steps, args.streams)),
                       'streams': args.streams, 'stream_plan': plan,
                       # which paths the renders took (rdoom_batch_path_stats of every batch's last render)
                       'paths': paths,
                       **({'debug': args.debug} if args.debug else {}),
                       'workload_key': workload_key(args, levels), 'measurement_key': measurement_key(args, levels),
                       'comparable_with_rounds_1_2': 'single_stream.value' if args.streams > 1 else 'value',
                       'kernel_sources': kernel_source_digest()},
            # spread of the timed steps (GPU time stamps at the end of every step, max over the render streams)
            'value_spread': {'step_ms_min': round(min(step_ms), 3), 'step_ms_median': round(float(np.median(step_ms)), 3),
                             'step_ms_max': round(max(step_ms), 3),
                             'value_min': round(my_px_per_step * world / max(step_ms) / 1e3, 1), 'value_max': round(my_px_per_step * world / min(step_ms) / 1e3, 1),
                             # (the first step after the opening barrier starts on an empty GPU and pays the set-up chain's latency
                             # un-overlapped; the last one drains alone: the mean over K steps carries both, the median does not)
                             'step_ms_first': round(step_ms[0], 3), 'step_ms_last': round(step_ms[-1], 3),
                             'value_at_median_step': round(my_px_per_step * world / float(np.median(step_ms)) / 1e3, 1),
                             'from': 'rank 0, one event per step and render stream'},
            'roofline': {'bound': 'hbm', 'kernel': 'fragment_kernel', 'achieved': round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         # frac: SURVEY 8(d)'s 6 B/px; frac_layout_bytes: the 2 + 2 B/px this layout keeps per pixel (the quadrant
                         # table lets uniform quadrants skip even those); frac_actual_bytes: HBM traffic by the PMC counters
                         'frac_layout_bytes': round(achieved / HBM_PEAK_GBS * LAYOUT_READ_BYTES_PER_PIXEL / ALG_READ_BYTES_PER_PIXEL, 4),
                         'traffic': traffic, 'frac_actual_bytes': frac_actual, 'traffic_source': pmc_source,
                         # what binds the step: VALU issue, not HBM (PMC passes of tools/profile_round.sh on THESE kernel sources,
                         # null when none were taken): wave64 VALU instructions per pixel, issue cycles = SQ_ACTIVE_INST_VALU x 4 / 1024
                         # SIMDs, frac = issue cycles / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs), per hot kernel and for the step
                         'valu': valu},
            'parity': parity,
            'cpu_baseline': cpu,
        }
        if per_rank_ms is not None:
            out['per_rank_ms_per_step'] = per_rank_ms
            out['rank_devices'] = rank_devices
        if single_elapsed is not None:   # the same work without the overlap: the step the per-kernel figures add up to
            out['single_stream'] = {'value': round(my_px_per_step * world * k_steps / single_elapsed / 1e6, 1), 'unit': 'Mpixels/s',
                                    'ms_per_step': round(single_elapsed / k_steps * 1e3, 3)}
            out['roofline']['measured_in'] = 'single-stream pass (see config.kernels_ms_from)'
        if present < world:
            out['gpus_present'] = present
            out['note'] = '%d ranks wrapped onto %d GPU(s) over gloo: exercises the N > 1 path, not a scaling measurement' % (world, present)

    # ---- after the headline: the other BASELINE configurations, on the same box, library and timing scheme -------------------------
    # the headline's scratch is released first (every rank)
    for c in closers:
        c.close()
    want_other = args.other == 'on' or (args.other == 'auto' and default_workload)
    if world > 1 and want_other:
        # N > 1: the number north_star gates on -- BASELINE config 4 under --scaling strong: E1M1..E1M9, 1024 poses per level, rank g
        # renders poses [g n/G, (g+1) n/G) of EVERY level (one level set, mixed sub-batches), barriers on both sides, slowest rank
        # counts.  Then rank 0 ALONE renders the full batch the same way (the others wait): the one-GPU time of the same run, box and
        # library -- measured_speedup = that / the G-GPU step.
        spec = OTHER_WORKLOADS[0]
        wad4 = rd.Wad(synthetic.ensure_wad(), synthetic.META_PATH)
        line = quick_line(rd, torch, sharding, wad4, spec, args.streams, steps=20, warmup=3, kernels=False, rank=rank, world=world, dist=dist,
                          backend_device=dist_device)
        gathered = [None] * world
        dist.all_gather_object(gathered, line.get('rank_ms_per_step'))
        alone = None
        if rank == 0:
            alone = quick_line(rd, torch, sharding, wad4, spec, args.streams, steps=20, warmup=3, kernels=False)
        dist.barrier()
        if rank == 0:
            out['strong_config4'] = {
                'workload': line['workload'] + ', cut into %d contiguous pose ranges per level (--scaling strong)' % world,
                'value': line['value'], 'unit': 'Mpixels/s', 'ms_per_step': line['ms_per_step'], 'steps': line['steps'], 'per_rank_ms': gathered,
                'rank_devices': rank_devices, 'poses_per_level_and_rank': line['poses_per_level_and_rank'], 'streams': line['streams'], 'stream_plan': line['stream_plan'],
                'one_gpu_same_run': {'value': alone['value'], 'ms_per_step': alone['ms_per_step'], 'what': 'rank 0 alone, the full batch, the other ranks idle'},
                'measured_speedup': round(alone['ms_per_step'] / line['ms_per_step'], 2),
                'speedup_vs_share_model': 'the share model (other_workloads[0].scaling_proxy of the N = 1 line) predicts full_ms / share_ms; this is the measurement it predicts',
                **({'gpus_present': present, 'note': 'ranks wrapped onto %d GPU(s): the path, not a scaling measurement' % present} if present < world else {})}
    if rank == 0:
        other_lines = None
        if world == 1 and want_other:
            other_lines = []
            wads = {}
            for spec in OTHER_WORKLOADS:
                key = bool(spec['big'])
                if key not in wads:
                    wads[key] = rd.Wad(synthetic.ensure_big_wad() if key else synthetic.ensure_wad(), synthetic.META_PATH)
                try:
                    ws = 1 if cli_streams == 1 else spec.get('streams', resolve_streams(0, len(spec['levels'])))   # (--streams 1: everything on one stream)
                    # (the line a strong-scaling share is compared with gets as many steps as the share: the first step after a
                    # synchronisation starts on an empty GPU, and that ramp weighs more on a 5 ms step than on a 37 ms one)
                    line = quick_line(rd, torch, sharding, wads[key], spec, ws, **({'steps': 20} if spec.get('share') else {}))
                    if spec.get('share'):   # the strong-scaling share a one-GPU box can time (BASELINE config 4 is strong scaling)
                        line['scaling_proxy'] = scaling_proxy(rd, torch, sharding, wads[key], spec, spec['share'])
                    other_lines.append(line)
                except Exception as e:  # noqa: BLE001  (a failing extra must not take the headline with it: it says so instead)
                    other_lines.append({'workload': spec['name'], 'error': repr(e)})
            if args.rich_line:   # the texture-rich stand-in (tools/mkwad.py --rich): a texel store beyond one XCD's L2
                try:
                    rwad = rd.Wad(synthetic.ensure_rich_wad(), synthetic.META_PATH)
                    other_lines.append(quick_line(rd, torch, sharding, rwad, dict(name='texture-rich E1M1 stand-in (every wall and flat its own texture)', levels=[0], big=False,
                                                                                     width=1920, height=1080, poses=1024, tv=False, parity=2), 3))
                except Exception as e:  # noqa: BLE001
                    other_lines.append({'workload': 'texture-rich E1M1 stand-in', 'error': repr(e)})
        out['other_workloads'] = other_lines
        if world == 1 and args.share > 1:
            spec = dict(name=metric_label(args, levels), levels=levels, big=args.big, width=args.width, height=args.height, poses=args.poses, tv=args.time_varying)
            out['scaling_proxy'] = scaling_proxy(rd, torch, sharding, wad, spec, args.share, steps=max(8, min(steps_requested, 20)))
        if args.long > 0:
            out['long'] = {'asked_seconds': args.long, 'timed_seconds': round(elapsed, 3)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        bad = [pp for pp in [out.get('parity')] + [o.get('parity') for o in (out.get('other_workloads') or [])] if pp and pp['mismatching_pixels']]
        if bad:
            raise SystemExit('bench.py: frames of the timed poses differ from the oracle: %r' % bad)


if __name__ == '__main__':
    main()

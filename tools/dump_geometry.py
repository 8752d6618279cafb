#!/usr/bin/env python3
"""BASELINE config 1: the CPU-only geometry build of one level -> wall / flat triangle dump, with the timings the
reference logs for the same phases.  No GPU is touched: rdoom_wad_open + rdoom_wad_build_level(use_gpu_tessellation = 0)
is the product's C++ restatement of `wad` + `game::level` running on one host thread.

    python tools/dump_geometry.py <iwad> <metadata.toml> <level index> <out dir> [--repeat N]

Written into <out dir> (SURVEY 8(d) "Config 1"):
    verts.bin                         the StaticVertex stream handed to the renderer, 48-byte records (game/src/vertex.rs:5-16)
    sky_verts.bin                     SkyVertex: 3 x f32 per record (vertex.rs:18-28)
    decor_verts.bin                   SpriteVertex, 44-byte records (vertex.rs:30-51)
    indices_{obj}_{flat|wall|sky|decor}.bin
                                      u32 index lists per ObjectId, as Builder keeps them (game/src/level.rs:275-305); flat
                                      and wall lists index verts.bin, sky lists sky_verts.bin, decor lists decor_verts.bin
    counters.json                     the counters of the reference's "Level built in ..." log line (level.rs:384-422), the
                                      timings below and the sizes of everything written
Timings (single thread, steady clock inside the library; median of --repeat runs, default 9; the first run -- cold page
cache, first-touch allocations -- is reported separately):
    t_load_ms = open + textures + level lumps + atlases      (rows a1-a7: what the reference times at wad/src/tex.rs:67-88,
                                                               371-408, 479-495 and loads in GameShaders::load_level)
    t_walk_ms = analysis + walk + Builder + index lists       (rows a9-a15: "Level built in {:.2}ms", level.rs:333, 384-396)
"""
import argparse
import json
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rust_doom_amd as rd  # noqa: E402

KIND_NAMES = {0: 'flat', 1: 'wall', 2: 'decor', 3: 'sky'}   # RDOOM_KIND_* (include/rdoom.h)
LOAD_KEYS = ('open_ms', 'textures_ms', 'level_lumps_ms', 'atlases_ms')
WALK_KEYS = ('analysis_ms', 'walk_ms')


def time_build(iwad, meta, level, repeat):
    """per-phase medians over `repeat` complete load + build runs on this thread; returns (first run, medians, last built)"""
    runs, built = [], None
    for _ in range(max(1, repeat)):
        wad = rd.Wad(iwad, meta)
        built = wad.build_level(level, gpu_tessellation=False)
        t = dict(wad.timings())
        t.update({k: v for k, v in built.timings().items() if k in ('level_lumps_ms', 'atlases_ms', 'analysis_ms', 'walk_ms')})
        runs.append(t)
    med = {k: statistics.median(r[k] for r in runs) for k in runs[0]}
    return runs[0], med, built


def host_timings(iwad, meta, level, repeat=9):
    """{'t_load_ms', 't_walk_ms', 'phases_ms', 'first_run_ms', 'repeat'} for bench.py's cpu_baseline and counters.json"""
    first, med, _built = time_build(iwad, meta, level, repeat)
    return {'t_load_ms': round(sum(med[k] for k in LOAD_KEYS), 3), 't_walk_ms': round(sum(med[k] for k in WALK_KEYS), 3),
            'phases_ms': {k: round(v, 3) for k, v in med.items()},
            'first_run_ms': {'t_load_ms': round(sum(first[k] for k in LOAD_KEYS), 3), 't_walk_ms': round(sum(first[k] for k in WALK_KEYS), 3)},
            'repeat': repeat, 'threads': 1}


def dump(built, out_dir):
    """writes the arrays; returns {file name: bytes}"""
    os.makedirs(out_dir, exist_ok=True)
    a = built.arrays()
    sizes = {}

    def write(name, arr):
        path = os.path.join(out_dir, name)
        np.ascontiguousarray(arr).tofile(path)
        sizes[name] = os.path.getsize(path)

    write('verts.bin', a['static_vertices'])
    write('sky_verts.bin', a['sky_vertices'].astype(np.float32))
    write('decor_verts.bin', a['decor_vertices'])
    source = {0: a['static_indices'], 1: a['static_indices'], 2: a['decor_indices'], 3: a['sky_indices']}
    for kind, obj, first, count in a['draws']:
        write('indices_%d_%s.bin' % (obj, KIND_NAMES[int(kind)]), source[int(kind)][first:first + count].astype(np.uint32))
    return sizes


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('iwad')
    ap.add_argument('metadata')
    ap.add_argument('level', type=int)
    ap.add_argument('out_dir')
    ap.add_argument('--repeat', type=int, default=9)
    args = ap.parse_args()
    first, med, built = time_build(args.iwad, args.metadata, args.level, args.repeat)
    sizes = dump(built, args.out_dir)
    c = built.counters()
    wad = rd.Wad(args.iwad, args.metadata)
    info = {'iwad': os.path.basename(args.iwad), 'level': args.level, 'level_name': wad.level_name(args.level),
            'path': 'CPU only: rdoom_wad_open + rdoom_wad_build_level(use_gpu_tessellation = 0), one thread',
            'counters': c, 't_load_ms': round(sum(med[k] for k in LOAD_KEYS), 3), 't_walk_ms': round(sum(med[k] for k in WALK_KEYS), 3),
            'phases_ms': {k: round(v, 3) for k, v in med.items()},
            'first_run_ms': {k: round(v, 3) for k, v in first.items()}, 'repeat': args.repeat, 'files': sizes}
    with open(os.path.join(args.out_dir, 'counters.json'), 'w') as f:
        json.dump(info, f, indent=1, sort_keys=True)
    # the reference's wording (game/src/level.rs:384-396)
    print('Level built in %.2fms:' % info['t_walk_ms'])
    for key in ('num_wall_quads', 'num_floor_polys', 'num_ceil_polys', 'num_sky_wall_quads', 'num_sky_floor_polys',
                'num_sky_ceil_polys', 'num_decors', 'num_static_tris', 'num_sky_tris', 'num_sprite_tris'):
        print('\t%s = %d' % (key, c[key]))
    print('t_load %.3f ms (open %.3f + textures %.3f + level lumps %.3f + atlases %.3f), t_walk %.3f ms (analysis %.3f + walk %.3f); '
          '%d files, %d bytes -> %s' % (info['t_load_ms'], med['open_ms'], med['textures_ms'], med['level_lumps_ms'], med['atlases_ms'],
                                        info['t_walk_ms'], med['analysis_ms'], med['walk_ms'], len(sizes), sum(sizes.values()), args.out_dir))


if __name__ == '__main__':
    main()

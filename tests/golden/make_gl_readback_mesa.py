#!/usr/bin/env python3
"""Generates tests/golden/gl_readback/{census_mesa.json,frames_mesa.npz}: the SAME frames as make_gl_readback.py -- the 46
stored ones and the 158 of the extended census -- read back from a SECOND GL implementation: Mesa's llvmpipe (desktop OpenGL 4.5
core profile, opened headless through tests/mesa_headless.c), which runs the reference's six GLSL programs UNPATCHED behind the
engine's own `#version 140` line, `samplerBuffer` included (tests/gl_readback.py, backend 'mesa').

Stored: per frame the mismatch census of the ORACLE's frame against Mesa's readback (tests/gl_census.py: every differing pixel
labelled with the discontinuity that explains it, `other` = unexplained) and the zero-tolerance fragment-stage counts; and, so
that the GPU box -- which has neither Mesa's headers nor the reference checkout to run this -- can hold the HIP frames against
what Mesa drew, the readbacks themselves (RGB + the primitive Mesa's rasteriser chose) of the 27 golden poses at 320x200 and of
pose 0 of the benchmark sweep at 1920x1080.

    python tests/golden/make_gl_readback_mesa.py [--jobs N]     # needs /root/reference + Mesa's swrast_dri.so
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import gl_census  # noqa: E402
import gl_readback  # noqa: E402
import make_gl_readback as gen  # noqa: E402
from oracle import raster, wad_oracle  # noqa: E402
from util import GOLDEN, META_PATH, ensure_wad  # noqa: E402

OUT = os.path.join(GOLDEN, 'gl_readback')
STORED_READBACKS = ['L%d_P%d' % (i, k) for i in range(9) for k in range(3)] + ['L0_bench0_1080p']


def stored_frames():
    """the frame list of make_gl_readback.main(): golden poses, moving objects, targeted views, four 1080p poses of the sweep"""
    frames = gen.frame_list()
    frames.append(('L0_bench0_1080p', 0, 1920, 1080, gen.bench_pose(1920, 1080), None))
    for i in (341, 682, 1000):
        frames.append(('L0_bench%d_1080p' % i, 0, 1920, 1080, gen.bench_pose(1920, 1080, i), None))
    lv0 = wad_oracle.build_level(ensure_wad(), META_PATH, 0)
    frames += gen.targeted_frames(lambda i: lv0)
    return frames


def one_frame(lv, glref, oracle, pose, w, h, obj_seed, keep=False):
    mv, pr, t = pose[:16], pose[16:32], float(pose[32])
    lights = lv.lights.fill_buffer_at(t)
    om = None if obj_seed is None else gen.moving_object_views(lv, mv, obj_seed)
    rgb = glref.render(mv, pr, t, lights, w, h, object_modelviews=om)
    gid = glref.render(mv, pr, t, lights, w, h, mode='ids', object_modelviews=om)
    var = glref.render(mv, pr, t, lights, w, h, mode='varyings', object_modelviews=om)
    fb, prim = oracle.render(mv, pr, t, lights, w, h, want_prim=True, object_modelviews=om)
    c = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var, object_modelviews=om)
    r = gl_census.fragment_exact(oracle, lv, t, lights, rgb, gid, var)
    c['fragment_exact'] = {k: r[k] for k in ('pixels', 'disagree', 'by_kind', 'sky_sampler_boundary', 'row_division_boundary')}
    return c, ((rgb, gid) if keep else None)


def main():
    G = None
    levels, gls, oracles = {}, {}, {}
    census = {'frames': {}, 'extended': {}}
    arrays = {}

    def ctx(key):
        if key not in levels:
            path, index = gen.wad_of(key)
            levels[key] = wad_oracle.build_level(path, META_PATH, index)
        if key not in gls:
            gls[key] = gl_readback.GLReference(levels[key], backend='mesa')
            oracles[key] = raster.RasterOracle(levels[key])
        return levels[key], gls[key], oracles[key]

    for part, frames in (('frames', stored_frames()), ('extended', gen.extended_frames())):
        for key, index, w, h, pose, obj_seed in frames:
            lv, glref, oracle = ctx(index)
            c, kept = one_frame(lv, glref, oracle, np.asarray(pose, np.float32), w, h, obj_seed, keep=(part == 'frames' and key in STORED_READBACKS))
            c.update(level=index, width=w, height=h, time=float(pose[32]), objects_seed=obj_seed)
            census[part][key] = c
            if kept:
                arrays[key + '_rgb'], arrays[key + '_prim'] = kept
                arrays[key + '_pose'] = np.asarray(pose, np.float32)
            print(key, {k: v for k, v in c.items() if k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES}, flush=True)
    G = gl_readback.gl()
    census.update(gl_version=G.version, gl_renderer=G.renderer, glsl_version=G.glsl_version, subpixel_bits=G.subpixel_bits,
                  shader_head=gl_readback.ENGINE_VERSION_LINE, jitter_px=gl_census.JITTER)
    for part, tot_key, fe_key in (('frames', 'total', 'fragment_exact_total'), ('extended', 'extended_total', 'extended_fragment_exact_total')):
        census[tot_key] = {k: sum(f[k] for f in census[part].values()) for k in ('pixels', 'mismatch', 'winner_mismatch') + gl_census.CLASSES}
        census[fe_key] = gen.fragment_exact_total(census[part])
        census[fe_key]['row_division_boundary'] = sum(f['fragment_exact']['row_division_boundary'] for f in census[part].values())
        print(tot_key, census[tot_key], 'mismatch fraction %.5f' % (census[tot_key]['mismatch'] / census[tot_key]['pixels']))
        print(fe_key, census[fe_key])
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'frames_mesa.npz'), **arrays)
    with open(os.path.join(OUT, 'census_mesa.json'), 'w') as f:
        json.dump(census, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()

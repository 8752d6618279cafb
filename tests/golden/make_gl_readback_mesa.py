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

    python tests/golden/make_gl_readback_mesa.py            # needs /root/reference + Mesa's swrast_dri.so
    python tests/golden/make_gl_readback_mesa.py --extra    # census_mesa_extra.json: a wider net, counts only -- 60 IWADs of fresh generator
                                                            # seeds x 3 levels x 2 poses at random sizes / times, half with displaced doors
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


def describe_others(lv, mv, pr, t, lights, w, h, om, where, prim, gid, var, ovar):
    """what the census could NOT attribute to a discontinuity, pixel by pixel: same winner or not, how thin the winning triangle is
    on the screen (its height over its longest edge, in pixels), how far the ORACLE's own binary32 varyings are from the float64 ones"""
    f64 = gl_census.F64Frame(lv, mv, pr, t, lights, w, h, om)
    out = []
    for ix, iy, label in where:
        if label != 'other':
            continue
        op, gp = int(prim[iy, ix]), int(gid[iy, ix])
        rec = {'x': ix, 'y': iy, 'same_winner': (op & 0xFFFFFF) == (gp & 0xFFFFFF)}
        if rec['same_winner'] and op != 0xFFFFFFFF:
            s, c = f64._tri(op), f64.sample(op, ix + 0.5, iy + 0.5)
            rec['edge_margin_px'] = float(c['margin']) if c is not None else None
            if c is not None and 'tuv' in c and ovar is not None:
                rec['oracle_uv_off_texels'] = float(max(abs(float(ovar[iy, ix, 0]) - c['tuv'][0]), abs(float(ovar[iy, ix, 1]) - c['tuv'][1])))
                rec['gl_uv_off_texels'] = float(max(abs(float(var[iy, ix, 0]) - c['tuv'][0]), abs(float(var[iy, ix, 1]) - c['tuv'][1])))
                rec['oracle_dist_off_rel'] = float(abs(float(ovar[iy, ix, 2]) - c['dist']) / c['dist'])
                rec['crosses_eye_plane'] = bool(s['wmin'] <= 0.0)
        out.append(rec)
    return out


def one_frame(lv, glref, oracle, pose, w, h, obj_seed, keep=False, others=False):
    mv, pr, t = pose[:16], pose[16:32], float(pose[32])
    lights = lv.lights.fill_buffer_at(t)
    om = None if obj_seed is None else gen.moving_object_views(lv, mv, obj_seed)
    rgb = glref.render(mv, pr, t, lights, w, h, object_modelviews=om)
    gid = glref.render(mv, pr, t, lights, w, h, mode='ids', object_modelviews=om)
    var = glref.render(mv, pr, t, lights, w, h, mode='varyings', object_modelviews=om)
    fb, prim = oracle.render(mv, pr, t, lights, w, h, want_prim=True, object_modelviews=om)
    c = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var, object_modelviews=om, detail=others)
    if others:
        where = c.pop('where')
        c['others'] = describe_others(lv, mv, pr, t, lights, w, h, om, where, prim, gid, var,
                                      oracle.render_varyings(mv, pr, t, lights, w, h)[2] if om is None else None) if c['other'] else []
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


EXTRA_SEEDS = range(31000, 31060)
EXTRA_SIZES = [(320, 200), (640, 400), (964, 540), (1280, 720), (712, 296), (1366, 768)]


def extra_frames():
    """(key, level key, width, height, pose[33], object seed or None): levels no committed fixture was produced from"""
    out = []
    for seed in EXTRA_SEEDS:
        rng = np.random.RandomState(seed)
        for index in range(3):
            key = 'seed%d:%d' % (seed, index)
            for k in range(2):
                w, h = EXTRA_SIZES[rng.randint(len(EXTRA_SIZES))]
                i = int(rng.randint(1024))
                t = float(rng.choice([0.0, round(rng.uniform(0.0, 40.0), 1)]))
                if t > 0.0 and abs(t * 35.0 / 8.0 - round(t * 35.0 / 8.0)) < 1e-3:
                    t += 0.05   # not ON a boundary of static.vert's animation period (8 / 35 s): binary32 (GL, oracle) and the census's float64 model floor t / period differently there
                out.append(('seed%d_L%d_sweep%d_t%.1f_%dx%d' % (seed, index, i, t, w, h), key, w, h, gen.sweep_pose(key, w, h, i, time=t),
                            (5000 + seed + index) if rng.randint(2) else None))
    return out


def main_extra():
    census = {'frames': {}}
    for key, index, w, h, pose, obj_seed in extra_frames():
        path, li = gen.wad_of(index)
        lv = wad_oracle.build_level(path, META_PATH, li)
        c, _ = one_frame(lv, gl_readback.GLReference(lv, backend='mesa'), raster.RasterOracle(lv), np.asarray(pose, np.float32), w, h, obj_seed, others=True)
        c.update(level=index, width=w, height=h, time=float(pose[32]), objects_seed=obj_seed)
        census['frames'][key] = c
        print(key, {k: v for k, v in c.items() if k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES}, flush=True)
    G = gl_readback.gl()
    census.update(gl_version=G.version, gl_renderer=G.renderer, shader_head=gl_readback.ENGINE_VERSION_LINE, jitter_px=gl_census.JITTER)
    census['total'] = {k: sum(f[k] for f in census['frames'].values()) for k in ('pixels', 'mismatch', 'winner_mismatch') + gl_census.CLASSES}
    census['fragment_exact_total'] = gen.fragment_exact_total(census['frames'])
    census['fragment_exact_total']['row_division_boundary'] = sum(f['fragment_exact']['row_division_boundary'] for f in census['frames'].values())
    print('total', census['total'], 'mismatch fraction %.5f' % (census['total']['mismatch'] / census['total']['pixels']))
    print('fragment stage', census['fragment_exact_total'])
    with open(os.path.join(OUT, 'census_mesa_extra.json'), 'w') as f:
        json.dump(census, f, indent=1, sort_keys=True)


def extreme_frames():
    """poses the sweep rarely produces (tests/stress_extreme_poses.py): eyes within centimetres of vertices, walls and floors or far
    outside the level, pitches up to straight up / down -- triangles that cross the eye plane, leave the depth range inside a pixel
    block, cover the whole frame"""
    from util import reference_projection, view_matrix
    out = []
    for index, (w, h) in zip(range(9), [(640, 400), (964, 540), (324, 180), (1280, 720)] * 3):
        rng = np.random.RandomState(77000 + index)
        lv = wad_oracle.build_level(ensure_wad(), META_PATH, index)
        verts = lv.static_vertices['a_pos']
        for i in range(20):
            v = verts[rng.randint(len(verts))].astype(np.float64)
            kind = i % 4
            if kind == 0:
                eye = v + rng.uniform(-0.02, 0.02, 3)
            elif kind == 1:
                eye = v + np.array([rng.uniform(-0.3, 0.3), rng.choice([0.0, 1e-4, -1e-4]), rng.uniform(-0.3, 0.3)])
            elif kind == 2:
                eye = v * 1.0 + np.array([rng.uniform(-40, 40), rng.uniform(5, 60), rng.uniform(-40, 40)])
            else:
                eye = v + rng.uniform(-0.5, 0.5, 3)
            pitch = rng.choice([rng.uniform(-1.57, 1.57), 1.5707, -1.5707, 0.0])
            t = float(rng.choice([0.0, round(rng.uniform(0, 30), 2) + 0.013]))
            pose = np.zeros(33, np.float32)
            pose[:16], pose[16:32], pose[32] = view_matrix(eye, rng.uniform(0, 2 * np.pi), pitch), reference_projection(w, h), t
            out.append(('L%d_extreme%d_k%d_%dx%d' % (index, i, kind, w, h), index, w, h, pose, None))
    return out


def main_net(frames, out_name):
    census = {'frames': {}}
    for key, index, w, h, pose, obj_seed in frames:
        path, li = gen.wad_of(index)
        lv = wad_oracle.build_level(path, META_PATH, li)
        c, _ = one_frame(lv, gl_readback.GLReference(lv, backend='mesa'), raster.RasterOracle(lv), np.asarray(pose, np.float32), w, h, obj_seed, others=True)
        c.update(level=index, width=w, height=h, time=float(pose[32]), objects_seed=obj_seed)
        census['frames'][key] = c
        print(key, {k: v for k, v in c.items() if k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES}, flush=True)
    G = gl_readback.gl()
    census.update(gl_version=G.version, gl_renderer=G.renderer, shader_head=gl_readback.ENGINE_VERSION_LINE, jitter_px=gl_census.JITTER)
    census['total'] = {k: sum(f[k] for f in census['frames'].values()) for k in ('pixels', 'mismatch', 'winner_mismatch') + gl_census.CLASSES}
    census['fragment_exact_total'] = gen.fragment_exact_total(census['frames'])
    census['fragment_exact_total']['row_division_boundary'] = sum(f['fragment_exact']['row_division_boundary'] for f in census['frames'].values())
    print('total', census['total'], 'mismatch fraction %.5f' % (census['total']['mismatch'] / census['total']['pixels']))
    print('fragment stage', census['fragment_exact_total'])
    with open(os.path.join(OUT, out_name), 'w') as f:
        json.dump(census, f, indent=1, sort_keys=True)


def large_frames():
    """the 10 x E1M1 level (38 262 triangles) and the texture-rich stand-in (320 wall textures, a 4096 x 2048 atlas): other content than the nine levels"""
    out = [('big_sweep%d_t%.1f_640x400' % (i, t), 'big', 640, 400, gen.sweep_pose('big', 640, 400, i, time=t), None)
           for i, t in zip(range(5, 1024, 26), [0.0, 3.3, 0.0, 17.9] * 10)]
    return out


if __name__ == '__main__':
    if '--large' in sys.argv:
        main_net(large_frames(), 'census_mesa_large.json')
    elif '--extreme' in sys.argv:
        main_net(extreme_frames(), 'census_mesa_extreme.json')
    elif '--extra' in sys.argv:
        main_extra()
    else:
        main()

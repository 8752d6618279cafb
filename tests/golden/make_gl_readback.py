#!/usr/bin/env python3
"""Generates tests/golden/gl_readback/{frames.npz,census.json}: GL readbacks of the REFERENCE's own shaders
(assets/shaders/*.{vert,frag} of /root/reference, executed by SwiftShader through tests/gl_readback.py) for

  * the 27 golden poses (tests/golden/poses.npy: 9 levels x {spawn view = BASELINE config 2's pose at 320x200, two
    seeded views, the third at time 1.7 s with its own light table}),
  * three frames with moving objects (per-object u_modelview, engine/src/renderer.rs:120-132),
  * twelve targeted views of E1M1: under open sky looking up, close to decorations, along scrolling / animated
    textures at non-zero times,
  * poses 0, 341, 682 and 1000 of the benchmark sweep at 1920x1080 (BASELINE config 3's frame size),

each with the auxiliary winner-id pass, and the mismatch census of the ORACLE's frame against them
(tests/gl_census.py).  The readbacks are committed so that the GPU box -- which has neither the reference checkout nor
needs SwiftShader -- can compare the HIP frames with what the reference's shaders produced.

    python tests/golden/make_gl_readback.py          # rewrites the fixtures (needs /root/reference + SwiftShader)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import gl_census  # noqa: E402
import gl_readback  # noqa: E402
from oracle import raster, wad_oracle  # noqa: E402
from util import GOLDEN, META_PATH, ROOT, ensure_big_wad, ensure_wad  # noqa: E402

OUT = os.path.join(GOLDEN, 'gl_readback')


def moving_object_views(lv, modelview, seed):
    """view o model per object: every dynamic object shifted vertically (tests/test_gpu_raster_parity.py)."""
    rng = np.random.RandomState(seed)
    view = np.asarray(modelview, np.float64).reshape(4, 4).T
    om = np.zeros((int(lv.num_objects), 16), np.float32)
    for o in range(len(om)):
        model = np.eye(4)
        model[1, 3] = 0.0 if o == 0 else rng.uniform(-0.6, 0.6)
        om[o] = (view @ model).T.astype(np.float32).reshape(16)
    return om


def frame_list():
    """(key, level index, width, height, pose[33], object seed or None)"""
    poses = np.load(os.path.join(GOLDEN, 'poses.npy'))
    out = []
    for index in range(poses.shape[0]):
        for i in range(poses.shape[1]):
            out.append(('L%d_P%d' % (index, i), index, 320, 200, poses[index, i], None))
    for index, i in ((0, 1), (2, 0), (4, 2)):
        out.append(('L%d_P%d_objects' % (index, i), index, 320, 200, poses[index, i], 77 + index))
    return out


def targeted_frames(levels):
    """views the golden poses do not guarantee: under open sky looking up (sky.frag's mirrored / tiled bands), close to
    decorations (sprite.vert / sprite.frag), and along scrolling / animated textures at non-zero times."""
    from util import reference_projection, view_matrix
    out = []
    lv = levels(0)
    rng = np.random.RandomState(3)
    sky_verts = np.asarray(lv.sky_vertices, np.float32).reshape(-1, 3)
    for i in range(4):
        c = sky_verts[rng.randint(len(sky_verts))]
        eye = np.array([c[0] + rng.uniform(-0.5, 0.5), 0.45 + rng.uniform(0, 0.3), c[2] + rng.uniform(-0.5, 0.5)])
        pose = np.zeros(33, np.float32)
        pose[:16], pose[16:32] = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(0.3, 1.3)), reference_projection(320, 200)
        out.append(('L0_sky%d' % i, 0, 320, 200, pose, None))
    dv = lv.decor_vertices
    for k, q in enumerate(range(0, min(len(dv) // 4, 8), 2)):
        c = dv['a_pos'][4 * q:4 * q + 4].mean(0)
        ang = (0.3, 2.4, 4.5, 1.1)[k % 4]
        eye = c + np.array([np.sin(ang) * 1.3, 0.1, np.cos(ang) * 1.3])
        d = c - eye
        pose = np.zeros(33, np.float32)
        pose[:16], pose[16:32] = view_matrix(eye, np.arctan2(-d[0], -d[2]), 0.05), reference_projection(320, 200)
        pose[32] = (0.0, 0.4, 2.9, 11.3)[k % 4]   # decor animation frames (sprite.vert:25-37)
        out.append(('L0_decor%d' % k, 0, 320, 200, pose, None))
    # scrolling walls (special 0x30, visitor.rs:922) and animated flats / walls at several times
    sv = lv.static_vertices
    tri = sv[np.asarray(lv.static_indices).reshape(-1, 3)]
    moving = np.nonzero((tri['a_scroll_rate'][:, 2] != 0) | (tri['a_num_frames'][:, 2] > 1))[0]
    for k in range(4):
        t = tri[moving[rng.randint(len(moving))]]
        c = t['a_pos'].mean(0)
        ang = rng.uniform(0, 2 * np.pi)
        eye = c + np.array([np.sin(ang) * 1.0, 0.2, np.cos(ang) * 1.0])
        d = c - eye
        pose = np.zeros(33, np.float32)
        pose[:16], pose[16:32] = view_matrix(eye, np.arctan2(-d[0], -d[2]), -0.1), reference_projection(320, 200)
        pose[32] = (0.7, 3.3, 9.1, 27.5)[k]
        out.append(('L0_anim%d' % k, 0, 320, 200, pose, None))
    return out


def bench_pose(width, height, index=0):
    """pose `index` of the benchmark sweep of E1M1 (rust-doom_amd/sharding.py: pose_sweep)"""
    import importlib
    import rust_doom_amd as rd
    sharding = importlib.import_module('rust-doom_amd.sharding')
    built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
    p = sharding.pose_sweep(rd, built, 1, width, height, first=index)[0]
    out = np.zeros(33, np.float32)
    out[:16], out[16:32], out[32] = p['modelview'], p['projection'], p['time']
    return out


_BUILT = {}


OTHER_SEEDS = (7, 4242, 90210)   # tests/test_other_seeds.py


def other_seed_wad(seed):
    """an IWAD from another seed of tools/mkwad.py with three small random levels (tests/test_other_seeds.py writes the same)"""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mkwad
    path = os.path.join(tempfile.gettempdir(), 'rdoom_other_seed_%d.wad' % seed)
    if not os.path.exists(path):
        rng = np.random.RandomState(seed)
        specs = [('E1M%d' % (k + 1), ('gen', seed * 31 + k, int(rng.randint(24, 41)), int(rng.randint(4, 10)))) for k in range(3)]
        data, _ = mkwad.build_wad(seed, specs=specs)
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'wb') as f:
            f.write(data)
        os.replace(tmp, path)
    return path


def shapes_wad():
    """the IWAD with the lump shapes real files have (TEXTURE2, duplicated names, MAPxx, paired-rotation sprites, textures of many
    overlapping patches: tests/test_iwad_shapes.py, tests/golden/make_golden_shapes.py)"""
    import tempfile
    sys.path.insert(0, GOLDEN)
    from make_golden_shapes import build_shapes_wad
    path = os.path.join(tempfile.gettempdir(), 'rdoom_shapes_gl.wad')
    if not os.path.exists(path):
        tmp = '%s.%d.tmp' % (path, os.getpid())
        build_shapes_wad(tmp)
        os.replace(tmp, path)
    return path


def wad_of(key):
    """level key -> (IWAD path, level index): an index of the synthetic IWAD, 'big', 'seed<S>:<index>' or 'shapes:<index>'"""
    if key == 'big':
        return ensure_big_wad(), 0
    if isinstance(key, str) and key.startswith('shapes:'):
        return shapes_wad(), int(key[7:])
    if isinstance(key, str) and key.startswith('seed'):
        seed, index = key[4:].split(':')
        return other_seed_wad(int(seed)), int(index)
    return ensure_wad(), key


def sweep_pose(key, width, height, i, time=0.0):
    """pose i of the benchmark sweep (rust-doom_amd/sharding.py: pose_sweep) of the level `key` names (wad_of), at `time`"""
    import importlib
    import rust_doom_amd as rd
    sharding = importlib.import_module('rust-doom_amd.sharding')
    if key not in _BUILT:
        path, index = wad_of(key)
        _BUILT[key] = rd.Wad(path, META_PATH).build_level(index)
    p = sharding.pose_sweep(rd, _BUILT[key], 1, width, height, first=i, time=time)[0]
    out = np.zeros(33, np.float32)
    out[:16], out[16:32], out[32] = p['modelview'], p['projection'], p['time']
    return out


def extended_frames():
    """(key, level key, width, height, pose[33], object seed or None); level key = index, or 'big' for the 10 x E1M1 level"""
    out = [('L0_bench%d_1080p' % i, 0, 1920, 1080, sweep_pose(0, 1920, 1080, i), None) for i in range(16, 1024, 32)]
    for index in range(1, 9):
        out += [('L%d_sweep%d_640' % (index, i), index, 640, 400, sweep_pose(index, 640, 400, i), None) for i in range(0, 1024, 128)]
    # time-varying state (animated flats, scrolling walls, the light table of that time) with every door / lift displaced
    for index in range(9):
        for k, i in enumerate((40, 424, 808)):
            t = (1.7, 9.1, 27.5)[k] + index
            out.append(('L%d_sweep%d_t%.1f_objects_640' % (index, i, t), index, 640, 400, sweep_pose(index, 640, 400, i, time=t), 1000 + 16 * index + k))
    # BASELINE config 5's frame size, time-varying (the synthetic E1M3 and E1M1)
    for index, i, t in ((2, 5, 2.3), (2, 517, 6.9), (0, 261, 12.4), (0, 773, 0.0)):
        out.append(('L%d_sweep%d_t%.1f_2160p' % (index, i, t), index, 3840, 2160, sweep_pose(index, 3840, 2160, i, time=t), None))
    # the 10 x E1M1 level (38 k triangles: the MAP29 stand-in)
    out += [('big_sweep%d_1080p' % i, 'big', 1920, 1080, sweep_pose('big', 1920, 1080, i), None) for i in (3, 259, 515, 771)]
    # IWADs from other seeds of the generator: other textures, other atlas packings, other maps
    for seed in OTHER_SEEDS:
        for index in range(3):
            key = 'seed%d:%d' % (seed, index)
            out += [('seed%d_L%d_sweep%d_t%.1f_640' % (seed, index, i, t), key, 640, 400, sweep_pose(key, 640, 400, i, time=t), None)
                    for i, t in ((11, 0.0), (523, 5.3))]
    # the IWAD with the lump shapes of real files: TEXTURE2's textures, many-patch textures with holes, sprites without an ..A0 lump
    for index in range(3):
        key = 'shapes:%d' % index
        out += [('shapes_L%d_sweep%d_t%.1f_640' % (index, i, t), key, 640, 400, sweep_pose(key, 640, 400, i, time=t), None)
                for i, t in ((7, 0.0), (301, 4.1), (655, 0.0))]
    return out


def extended_census(lv, glref, oracle, pose, w, h, obj_seed=None):
    mv, pr, t = pose[:16], pose[16:32], float(pose[32])
    lights = lv.lights.fill_buffer_at(t)
    om = None if obj_seed is None else moving_object_views(lv, mv, obj_seed)
    rgb = glref.render(mv, pr, t, lights, w, h, object_modelviews=om)
    gid = glref.render(mv, pr, t, lights, w, h, mode='ids', object_modelviews=om)
    var = glref.render(mv, pr, t, lights, w, h, mode='varyings', object_modelviews=om)
    fb, prim = oracle.render(mv, pr, t, lights, w, h, want_prim=True, object_modelviews=om)
    c = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var, object_modelviews=om)
    c['fragment_exact'] = frag_exact_counts(oracle, lv, t, lights, rgb, gid, var)
    return c


def frag_exact_counts(oracle, lv, t, lights, rgb, gid, var):
    """the zero-tolerance fragment-stage pin (tests/gl_census.py: fragment_exact), counts only"""
    r = gl_census.fragment_exact(oracle, lv, t, lights, rgb, gid, var)
    return {k: r[k] for k in ('pixels', 'disagree', 'by_kind', 'sky_sampler_boundary')}


def fragment_exact_total(frames):
    tot = {'pixels': 0, 'disagree': 0, 'sky_sampler_boundary': 0, 'by_kind': {'flat': 0, 'wall': 0, 'decor': 0, 'sky': 0}}
    for f in frames.values():
        fe = f['fragment_exact']
        for k in ('pixels', 'disagree', 'sky_sampler_boundary'):
            tot[k] += fe[k]
        for k in tot['by_kind']:
            tot['by_kind'][k] += fe['by_kind'][k]
    return tot


def extended_level(levels, key):
    if key not in levels:
        path, index = wad_of(key)
        levels[key] = wad_oracle.build_level(path, META_PATH, index)
    return levels[key]


def main():
    wad = ensure_wad()
    frames = frame_list()
    frames.append(('L0_bench0_1080p', 0, 1920, 1080, bench_pose(1920, 1080), None))
    for i in (341, 682, 1000):  # three more poses of the sweep at BASELINE config 3's frame size
        frames.append(('L0_bench%d_1080p' % i, 0, 1920, 1080, bench_pose(1920, 1080, i), None))
    levels, gls, oracles = {}, {}, {}
    levels[0] = wad_oracle.build_level(wad, META_PATH, 0)
    frames += targeted_frames(lambda i: levels[i])
    arrays, census = {}, {'swiftshader': gl_readback.gl().version, 'subpixel_bits': gl_readback.gl().subpixel_bits,
                          'jitter_px': gl_census.JITTER, 'frames': {}}
    for key, index, w, h, pose, obj_seed in frames:
        if index not in levels:
            levels[index] = wad_oracle.build_level(wad, META_PATH, index)
        if index not in gls:
            gls[index] = gl_readback.GLReference(levels[index])
            oracles[index] = raster.RasterOracle(levels[index])
        lv = levels[index]
        mv, pr, t = pose[:16], pose[16:32], float(pose[32])
        lights = lv.lights.fill_buffer_at(t)
        om = None if obj_seed is None else moving_object_views(lv, mv, obj_seed)
        rgb = gls[index].render(mv, pr, t, lights, w, h, object_modelviews=om)
        gid = gls[index].render(mv, pr, t, lights, w, h, mode='ids', object_modelviews=om)
        var = gls[index].render(mv, pr, t, lights, w, h, mode='varyings', object_modelviews=om)
        fb, prim = oracles[index].render(mv, pr, t, lights, w, h, want_prim=True, object_modelviews=om)
        c = gl_census.census(lv, mv, pr, t, lights, w, h, fb, prim, rgb, gid, var, object_modelviews=om)
        c['fragment_exact'] = frag_exact_counts(oracles[index], lv, t, lights, rgb, gid, var)
        c.update(level=index, width=w, height=h, time=t, objects_seed=obj_seed)
        census['frames'][key] = c
        arrays[key + '_rgb'] = rgb
        arrays[key + '_prim'] = gid
        arrays[key + '_pose'] = np.asarray(pose, np.float32)
        print(key, {k: v for k, v in c.items() if k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES})
    tot = {k: sum(f[k] for f in census['frames'].values()) for k in ('pixels', 'mismatch', 'winner_mismatch') + gl_census.CLASSES}
    census['total'] = tot
    census['fragment_exact_total'] = fragment_exact_total(census['frames'])
    print('fragment stage, zero tolerance:', census['fragment_exact_total'])
    print('total', tot, 'mismatch fraction %.4f' % (tot['mismatch'] / tot['pixels']))
    # Extended census: counts only (no readbacks are stored for these frames), a wider net for systematic differences --
    # every 32nd pose of the benchmark sweep at 1920x1080, eight poses of each other level's sweep at 640x400, time-varying
    # frames with displaced doors / lifts on every level, 3840x2160 frames, the 10 x E1M1 level, IWADs from other seeds.
    census['extended'] = {}
    for key, index, w, h, pose, obj_seed in extended_frames():
        lv = extended_level(levels, index)
        if index not in gls:
            gls[index] = gl_readback.GLReference(lv)
            oracles[index] = raster.RasterOracle(lv)
        census['extended'][key] = c = extended_census(lv, gls[index], oracles[index], pose, w, h, obj_seed)
        c.update(level=index, width=w, height=h, time=float(pose[32]), objects_seed=obj_seed)
        print(key, {k: v for k, v in c.items() if k in ('mismatch', 'winner_mismatch') + gl_census.CLASSES})
    etot = {k: sum(f[k] for f in census['extended'].values()) for k in ('pixels', 'mismatch', 'winner_mismatch') + gl_census.CLASSES}
    census['extended_total'] = etot
    census['extended_fragment_exact_total'] = fragment_exact_total(census['extended'])
    print('fragment stage, zero tolerance (extended):', census['extended_fragment_exact_total'])
    print('extended total', etot, 'mismatch fraction %.4f' % (etot['mismatch'] / etot['pixels']))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'frames.npz'), **arrays)
    with open(os.path.join(OUT, 'census.json'), 'w') as f:
        json.dump(census, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()

"""GPU: level SETS (rdoom_levelset_create + rdoom_batch_render_levels) -- several levels resident together, ONE launch set per
render over poses of different levels (BASELINE config 4's share of one GPU: 128 poses of each of E1M1..E1M9).  Every pose of a
mixed batch must be, bit for bit, the frame (and the winning primitive ids) the oracle renders of THAT pose's level -- with
random times, per-level light tables and moving objects --, and the frame a single-level batch renders of it."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from util import reference_projection, render_checked, view_matrix

pytestmark = pytest.mark.gpu


def random_poses(lv, n, w, h, rng, moving=True):
    tri = lv.static_vertices['a_pos'][lv.static_indices.reshape(-1, 3)].mean(1)
    n_obj = int(lv.num_objects)
    poses, om, lights = np.zeros(n, rd.POSE), np.zeros((n, n_obj, 16), np.float32), np.zeros((n, 256), np.uint8)
    for i in range(n):
        c = tri[rng.randint(len(tri))]
        eye = np.array([c[0] + rng.uniform(-0.4, 0.4), c[1] + rng.uniform(-0.1, 0.7), c[2] + rng.uniform(-0.4, 0.4)])
        view = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(-1.0, 1.0))
        t = float(rng.choice([0.0, rng.uniform(0, 30)]))
        poses[i]['modelview'], poses[i]['projection'], poses[i]['time'] = view, reference_projection(w, h), t
        lights[i] = lv.lights.fill_buffer_at(t)
        v64 = view.astype(np.float64).reshape(4, 4).T
        for o in range(n_obj):
            m = np.eye(4)
            m[1, 3] = 0.0 if (o == 0 or i % 2 == 0 or not moving) else rng.uniform(-0.8, 0.8)
            om[i, o] = (v64 @ m).T.astype(np.float32).reshape(16)
    return poses, om, lights


def mixed_batch(levels, per_level, w, h, seed, moving):
    """poses of every level, shuffled: (poses, level_of_pose, lights, object modelviews padded to the set's object count)"""
    rng = np.random.RandomState(seed)
    parts = [random_poses(lv, per_level, w, h, rng, moving) for lv in levels]
    n_obj = max(p[1].shape[1] for p in parts)
    poses = np.concatenate([p[0] for p in parts])
    lights = np.concatenate([p[2] for p in parts])
    om = np.zeros((len(poses), n_obj, 16), np.float32)
    lop = np.zeros(len(poses), np.uint32)
    at = 0
    for k, p in enumerate(parts):
        om[at:at + per_level, :p[1].shape[1]] = p[1]
        lop[at:at + per_level] = k
        at += per_level
    order = rng.permutation(len(poses))
    return poses[order], lop[order], lights[order], om[order]


def check_against_oracles(levels, poses, lop, lights, om, fbs, prim, w, h):
    oracles = [raster.RasterOracle(lv) for lv in levels]

    def check(i):
        lv = levels[lop[i]]
        o = None if om is None else om[i, :int(lv.num_objects)]
        ofb, oprim = oracles[lop[i]].render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h,
                                            want_prim=True, object_modelviews=o)
        return sum(int((ofb != fb[i]).sum()) for fb in fbs), int((oprim != prim[i]).sum())

    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(check, range(len(poses))))
    bad = [(i, int(lop[i]), r) for i, r in enumerate(res) if r != (0, 0)]
    assert not bad, bad[:6]


@pytest.mark.parametrize('moving', [False, True])
@pytest.mark.parametrize('size', [(640, 400), (324, 180)])
def test_mixed_level_batch_equals_the_oracle_of_each_poses_level(oracle_levels, size, moving):
    """all nine levels in one set, three poses of each in one render, shuffled; times vary, so do the per-level light tables"""
    w, h = size
    levels = [oracle_levels(i) for i in range(9)]
    poses, lop, lights, om = mixed_batch(levels, 3, w, h, seed=4200 + w + int(moving), moving=moving)
    lset = rd.DeviceLevelSet(levels)
    assert lset.num_levels() == 9 and lset.num_objects() == max(int(lv.num_objects) for lv in levels)
    batch = rd.Batch(lset, w, h, len(poses))
    kw = {'level_of_pose': lop}
    if moving:
        kw['object_modelviews'] = om
    fb_plain, fb, prim = render_checked(batch, poses, lights, **kw)
    assert (prim != 0xFFFFFFFF).mean() > 0.3
    check_against_oracles(levels, poses, lop, lights, om if moving else None, (fb_plain, fb), prim, w, h)


def test_set_render_equals_single_level_batches_at_1080p(oracle_levels):
    """device against device at the BENCH frame size: a pose rendered through the set == the same pose through a batch of its level
    alone (which test_gpu_full_size.py holds against the oracle); a slice of the frames also against the oracle directly"""
    w, h = 1920, 1080
    idx = [0, 3, 8]
    levels = [oracle_levels(i) for i in idx]
    poses, lop, lights, _om = mixed_batch(levels, 4, w, h, seed=77, moving=False)
    lset = rd.DeviceLevelSet(levels)
    batch = rd.Batch(lset, w, h, len(poses))
    fb_plain, fb, prim = render_checked(batch, poses, lights, level_of_pose=lop)
    assert np.array_equal(fb_plain, fb)
    for k, lv in enumerate(levels):
        sel = np.nonzero(lop == k)[0]
        single = rd.Batch(rd.DeviceLevel(lv), w, h, len(sel))
        s_plain, s_fb, s_prim = render_checked(single, poses[sel], lights[sel])
        assert np.array_equal(s_plain, fb[sel]) and np.array_equal(s_fb, fb[sel]) and np.array_equal(s_prim, prim[sel]), k
    sel = np.array([int(np.nonzero(lop == k)[0][0]) for k in range(len(levels))])
    check_against_oracles(levels, poses[sel], lop[sel], lights[sel], None, (fb[sel],), prim[sel], w, h)


def test_a_set_of_one_is_the_level(oracle_levels):
    """rdoom_level_create IS a set of one: the same frames with and without level_of_pose"""
    w, h = 400, 240
    lv = oracle_levels(2)
    poses, _om, lights = random_poses(lv, 6, w, h, np.random.RandomState(5), moving=False)
    level = rd.DeviceLevel(lv)
    assert rd.DeviceLevelSet([lv]).num_levels() == 1
    a = rd.Batch(level, w, h, len(poses))
    a.render(poses, lights)
    fa = a.read_framebuffer()
    b = rd.Batch(rd.DeviceLevelSet([lv]), w, h, len(poses))
    b.render(poses, lights, level_of_pose=np.zeros(len(poses), np.uint32))
    assert np.array_equal(fa, b.read_framebuffer())


def test_profiled_set_renders_report_their_kernel_times(oracle_levels):
    w, h = 640, 400
    levels = [oracle_levels(i) for i in (0, 1)]
    poses, lop, lights, _om = mixed_batch(levels, 8, w, h, seed=9, moving=False)
    batch = rd.Batch(rd.DeviceLevelSet(levels), w, h, len(poses))
    for _ in range(3):
        batch.render_profiled(poses, lights, level_of_pose=lop)
    t = batch.collect_timings()
    assert t['renders'] == 3 and t['fragment_ms'] > 0 and t['pixels'] == 3 * len(poses) * w * h


def test_bad_arguments(oracle_levels):
    lv0, lv1 = oracle_levels(0), oracle_levels(1)
    w, h = 128, 80
    lset = rd.DeviceLevelSet([lv0, lv1])
    batch = rd.Batch(lset, w, h, 4)
    poses, _om, lights = random_poses(lv0, 4, w, h, np.random.RandomState(1), moving=False)
    with pytest.raises(rd.RdoomError) as e:   # a level the set does not hold
        batch.render(poses, lights, level_of_pose=np.array([0, 1, 2, 0], np.uint32))
    assert e.value.status == -1 and 'level 2' in str(e.value)
    batch.render(poses, lights, level_of_pose=np.array([0, 0, 0, 0], np.uint32))   # (and the batch still renders afterwards)
    batch.finish()
    other = {k: getattr(lv1, k) for k in ('static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices', 'draws',
                                          'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture', 'sky_band', 'palette')}
    other['colormap'] = np.asarray(lv1.colormap, np.uint8)[::-1].copy()   # another IWAD's COLORMAP
    with pytest.raises(rd.RdoomError) as e:
        rd.DeviceLevelSet([lv0, other])
    assert e.value.status == -1 and 'COLORMAP' in str(e.value)
    with pytest.raises(rd.RdoomError):
        rd.DeviceLevelSet([])

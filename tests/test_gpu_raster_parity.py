"""GPU parity (raster half): HIP setup+raster+fragment kernels vs the C oracle on the SAME level
arrays (the oracle-built arrays are passed through the C ABI, so only the kernels are under test).
Bit-exact: u8 palette-index framebuffers and u32 winning-primitive ids."""
import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from util import META_PATH, reference_projection, render_checked, view_matrix

pytestmark = pytest.mark.gpu


def sweep_poses(lv, n, width, height, seed=7, time=0.0):
    rng = np.random.RandomState(seed)
    cents = []
    for s in range(0, len(lv.static_indices) // 3, 37):
        tri = lv.static_vertices['a_pos'][lv.static_indices[3 * s:3 * s + 3]]
        cents.append(tri.mean(0))
    poses = np.zeros(n, rd.POSE)
    sp = np.array([float(x) for x in lv.start_pos])
    for i in range(n):
        if i == 0:  # the reference's spawn view in the reference's own arithmetic (rdoom_pose_from_player)
            p0 = rd.pose_from_player(sp, float(lv.start_yaw), 1e-8, width, height, time)
            poses[0]['modelview'], poses[0]['projection'], poses[0]['time'] = p0['modelview'], p0['projection'], time
            continue
        else:
            c = cents[rng.randint(len(cents))]
            eye = np.array([c[0] + rng.uniform(-0.3, 0.3), c[1] + rng.uniform(0.2, 0.6), c[2] + rng.uniform(-0.3, 0.3)])
            yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.5, 0.5)
        poses[i]['modelview'] = view_matrix(eye, yaw, pitch)
        poses[i]['projection'] = reference_projection(width, height)
        poses[i]['time'] = time
    return poses


def run_case(lv, width, height, n, kinds, time=0.0):
    poses = sweep_poses(lv, n, width, height, time=time)
    lights = lv.lights.fill_buffer_at(time)
    dev = rd.DeviceLevel(lv)
    batch = rd.Batch(dev, width, height, n)
    fb_plain, fb, prim = render_checked(batch, poses, lights, kinds=kinds)  # after a dirtying render; without and with primitive ids
    ro = raster.RasterOracle(lv)
    bad = []
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], time, lights, width, height, kinds=kinds,
                               want_prim=True)
        npx = int((oprim != prim[i]).sum()), int((ofb != fb[i]).sum()), int((ofb != fb_plain[i]).sum())
        if npx != (0, 0, 0):
            bad.append((i, npx))
    assert not bad, 'mismatching (pose, (prim px, colour px, colour px of the path without ids)): %r' % bad[:10]
    return fb


def test_static_320x200_single_pose(oracle_levels):
    """BASELINE config 2: E1M1, reference spawn pose, 320x200, static walls+flats."""
    lv = oracle_levels(0)
    fb = run_case(lv, 320, 200, 1, (1 << rd.KIND_FLAT) | (1 << rd.KIND_WALL))
    assert (fb != 0).mean() > 0.5


def test_static_sweep_e1m1(oracle_levels):
    run_case(oracle_levels(0), 320, 200, 24, (1 << rd.KIND_FLAT) | (1 << rd.KIND_WALL))


def test_static_plus_sky_sweep(oracle_levels):
    run_case(oracle_levels(0), 320, 200, 16, rd.ALL_KINDS)


def test_kat_level_and_odd_sizes(oracle_levels):
    lv = oracle_levels(1)
    run_case(lv, 64, 33, 6, rd.ALL_KINDS)      # partial tiles in both directions
    run_case(lv, 324, 201, 4, rd.ALL_KINDS)    # width a multiple of 4 but not of 8: one quad per lane
    run_case(lv, 456, 120, 3, rd.ALL_KINDS)    # 7 x 64 + 8 pixels: the last 64x8 block of a row has one valid unit (no wave-uniform path there)
    run_case(lv, 1920, 1080, 2, rd.ALL_KINDS)  # BASELINE resolution


def test_any_window_size(oracle_levels):
    """The reference takes any --resolution WxH (src/main.rs:41): widths that are not a multiple of 4 render with a padded row
    pitch (rdoom.h: rdoom_batch_framebuffer_pitch) and read back tightly packed; heights that are not a multiple of 4 have pixel
    rows below the frame in their bottom block row (raster.hip: they start at depth 0).  1366x768 and 1600x900 are the sizes the
    round-4 review names; 322 / 321 / 323 pad by 6 / 7 / 5 columns (a whole 4-pixel block of padding in the first two)."""
    lv = oracle_levels(1)
    for w, h, n in ((322, 200, 4), (321, 199, 3), (323, 130, 3), (1366, 768, 2), (1600, 900, 2), (1921, 1082, 1), (5, 3, 2), (9, 70, 2),
                    (1284, 724, 2)):   # (a large frame whose pitch is a multiple of 4 only: one quad per lane, 32 x 16 blocks all the same)
        run_case(lv, w, h, n, rd.ALL_KINDS)
    lv = oracle_levels(0)
    run_case(lv, 1366, 768, 3, rd.ALL_KINDS, time=0.9)
    run_case(lv, 1920, 1082, 2, rd.ALL_KINDS)   # height % 4 == 2: the bottom quadrant row crosses the frame's edge inside a block


def test_framebuffer_pitch(oracle_levels):
    lv = oracle_levels(1)
    dev = rd.DeviceLevel(lv)
    for w, want in ((320, 320), (324, 324), (322, 328), (321, 328), (1366, 1368), (4, 8), (1, 8)):
        b = rd.Batch(dev, w, 16, 1)
        assert b.framebuffer_pitch() == want, (w, b.framebuffer_pitch())
        b.close()


def test_path_stats(oracle_levels):
    """rdoom_batch_path_stats: overflowed poses, split tile lists, described quadrants of the last render -- against what the
    hooks force (entry_cap: every pose overflows; no_split: no split lists) and against the table's own arithmetic"""
    lv = oracle_levels(0)
    dev = rd.DeviceLevel(lv)
    w, h, n = 320, 200, 6
    poses = sweep_poses(lv, n, w, h)
    lights = lv.lights.fill_buffer_at(0.0)
    b = rd.Batch(dev, w, h, n)
    b.render(poses, lights)
    s = b.path_stats()
    assert s['poses'] == n and s['bins_overflowed_poses'] == 0
    assert s['tiles'] == n * 5 * 4 and s['quadrants'] == n * (10 * 7)         # 320x200: 5 x 4 tiles, 10 x 7 quadrants inside the frame (200 = 6.25 x 32)
    assert 0 < s['described_quadrants'] < s['quadrants'] and s['tile_entries'] > s['tiles']
    assert s['split_tiles'] > 0                                                # far tiles hold more than 64 entries at this size
    fb = b.read_framebuffer()
    try:
        rd.debug_set('no_split', 1)
        b.render(poses, lights)
        s2 = b.path_stats()
        assert s2['split_tiles'] == 0 and s2['tile_entries'] == s['tile_entries'] and s2['described_quadrants'] >= 1
        assert np.array_equal(b.read_framebuffer(), fb)
        rd.debug_set('reset', 0)
        rd.debug_set('entry_cap', 300)
        b2 = rd.Batch(dev, w, h, n)                                            # (entry_cap is read when a batch is created)
        b2.render(poses, lights)
        s3 = b2.path_stats()
        assert 1 <= s3['bins_overflowed_poses'] <= n and np.array_equal(b2.read_framebuffer(), fb)   # (poses that look at little fit 300 entries)
    finally:
        rd.debug_set('reset', 0)


def test_4k_time_varying(oracle_levels):
    """BASELINE config 5's frame size (3840x2160) with animated flats, scrolling walls and the per-pose light
    table at t != 0, on the synthetic E1M3 (no DOOM2.WAD exists here): 2040 tiles per frame, 8.3 Mpixel."""
    run_case(oracle_levels(2), 3840, 2160, 2, rd.ALL_KINDS, time=2.3)


def test_time_varying(oracle_levels):
    """animated flats / scrolling walls / light table at t != 0 (static.vert:26-39)."""
    run_case(oracle_levels(0), 320, 200, 8, rd.ALL_KINDS, time=1.7)


def test_deterministic(oracle_levels):
    lv = oracle_levels(0)
    a = run_case(lv, 320, 200, 4, rd.ALL_KINDS)
    b = run_case(lv, 320, 200, 4, rd.ALL_KINDS)
    assert np.array_equal(a, b)


def test_decor_billboards(oracle_levels):
    """sprite.vert / sprite.frag: poses placed around decorations so that sprites fill part of
    the frame; alpha-tested, camera-facing, lit with min(v_light, 2 v_light - dist_term)."""
    lv = oracle_levels(0)
    dv = lv.decor_vertices
    assert len(dv) >= 8
    w, h = 320, 200
    poses = []
    for q in range(0, len(dv) // 4, 2):
        c = dv['a_pos'][4 * q:4 * q + 4].mean(0)
        for ang in (0.3, 2.4, 4.5):
            eye = c + np.array([np.sin(ang) * 1.3, 0.1, np.cos(ang) * 1.3])
            d = c - eye
            p = np.zeros(1, rd.POSE)[0]
            p['modelview'] = view_matrix(eye, np.arctan2(-d[0], -d[2]), 0.05)
            p['projection'] = reference_projection(w, h)
            poses.append(p)
    poses = np.array(poses[:24], rd.POSE)
    lights = lv.lights.fill_buffer_at(0.0)
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, len(poses))
    fb_plain, fb, prim = render_checked(batch, poses, lights)
    ro = raster.RasterOracle(lv)
    first = np.cumsum([0] + [int(d[3]) // 3 for d in lv.draws])
    decor_px = 0
    for i in range(len(poses)):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h, want_prim=True)
        assert np.array_equal(oprim, prim[i]) and np.array_equal(ofb, fb[i]) and np.array_equal(ofb, fb_plain[i]), i
        for di, d in enumerate(lv.draws):
            if d[0] == rd.KIND_DECOR:
                decor_px += int(((oprim >= first[di]) & (oprim < first[di + 1])).sum())
    assert decor_px > 10000  # the sprites really are in view


def test_moving_objects(oracle_levels):
    """doors / lifts move their object's transform (game/src/level.rs:203-255); the reference then
    draws that object with u_modelview = view o model (engine/src/renderer.rs:120-132).  Every object gets its own
    vertical offset per pose; the per-object matrices go through rdoom_batch_render_objects."""
    lv = oracle_levels(0)
    w, h, n = 320, 200, 10
    n_obj = int(lv.num_objects)
    assert n_obj > 1 and int(lv.draws[:, 1].max()) == n_obj - 1
    poses = sweep_poses(lv, n, w, h, seed=21)
    lights = lv.lights.fill_buffer_at(0.0)
    rng = np.random.RandomState(5)
    om = np.zeros((n, n_obj, 16), np.float32)
    for p in range(n):
        view = poses[p]['modelview'].astype(np.float64).reshape(4, 4).T  # row-major
        for o in range(n_obj):
            model = np.eye(4)
            model[1, 3] = 0.0 if o == 0 else rng.uniform(-0.6, 0.6)  # object 0 is the static world
            om[p, o] = (view @ model).T.astype(np.float32).reshape(16)
    dev = rd.DeviceLevel(lv)
    assert dev.num_objects() == n_obj
    batch = rd.Batch(dev, w, h, n)
    fb_plain, fb, prim = render_checked(batch, poses, lights, object_modelviews=om)
    ro = raster.RasterOracle(lv)
    moved = 0
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h, want_prim=True,
                               object_modelviews=om[i])
        assert np.array_equal(oprim, prim[i]) and np.array_equal(ofb, fb[i]) and np.array_equal(ofb, fb_plain[i]), i
        still = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h)
        moved += int((still != ofb).sum())
    assert moved > 2000  # the offsets are visible
    # at rest (every object with the pose's own modelview) the image is the ordinary one
    rest = np.repeat(poses['modelview'][:, None, :], n_obj, axis=1)
    batch.render(poses, lights, object_modelviews=rest)
    fb_rest = batch.read_framebuffer()
    batch.render(poses, lights)
    assert np.array_equal(fb_rest, batch.read_framebuffer())


def test_edge_cases(oracle_levels):
    """empty level, nothing selected, smallest frame, batch bounds"""
    from test_kat_analytic import kat_level
    lvl, lights = kat_level()
    empty = dict(lvl)
    empty['draws'] = np.zeros((0, 4), np.uint32)
    poses = np.zeros(2, rd.POSE)
    poses['modelview'] = np.eye(4, dtype=np.float32).reshape(16)
    poses['projection'] = reference_projection(64, 40)
    b = rd.Batch(rd.DeviceLevel(empty), 64, 40, 2)
    b.render(poses, lights)
    assert not b.read_framebuffer().any()                       # no triangles: background everywhere
    dev = rd.DeviceLevel(lvl)
    b = rd.Batch(dev, 64, 40, 2)
    b.render(poses, lights, kinds=0)
    assert not b.read_framebuffer().any()                       # no kind selected
    b.render(poses[:1], lights)                                 # fewer poses than the batch holds
    one = b.read_framebuffer()
    assert one.shape == (1, 40, 64) and one.any()
    with pytest.raises(rd.RdoomError):
        b.render(np.zeros(3, rd.POSE), lights)                  # more poses than max_poses
    with pytest.raises(rd.RdoomError):
        rd.Batch(dev, 0, 40, 1)                                 # an empty frame
    with pytest.raises(rd.RdoomError):
        rd.Batch(dev, 16388, 40, 1)                             # beyond 16384 on a side
    odd = rd.Batch(dev, 66, 40, 1)                              # any width (round 4: "width must be a multiple of 4")
    p66 = np.zeros(1, rd.POSE)
    p66['modelview'] = np.eye(4, dtype=np.float32).reshape(16)
    p66['projection'] = reference_projection(66, 40)
    odd.render(p66, lights)
    assert np.array_equal(odd.read_framebuffer()[0], raster.RasterOracle(lvl).render(p66[0]['modelview'], p66[0]['projection'], 0.0, lights, 66, 40))
    small = rd.Batch(dev, 8, 1, 1)                              # one row of eight pixels
    p = np.zeros(1, rd.POSE)
    p['modelview'] = np.eye(4, dtype=np.float32).reshape(16)
    p['projection'] = reference_projection(8, 1)
    small.render(p, lights)
    want = raster.RasterOracle(lvl).render(p[0]['modelview'], p[0]['projection'], 0.0, lights, 8, 1)
    assert np.array_equal(small.read_framebuffer()[0], want)


def test_sky_heavy_views(oracle_levels):
    """looking up from the sub-sectors under open sky: long runs of sky pixels (the fragment kernel's sky-run path,
    sky.frag:12-26 incl. the mirrored / tiled bands above the texture)"""
    lv = oracle_levels(0)
    w, h = 320, 200
    sky_verts = np.asarray(lv.sky_vertices, np.float32).reshape(-1, 3)
    assert len(sky_verts) > 0
    rng = np.random.RandomState(3)
    poses = np.zeros(20, rd.POSE)
    for i in range(len(poses)):
        c = sky_verts[rng.randint(len(sky_verts))]
        eye = np.array([c[0] + rng.uniform(-0.5, 0.5), 0.45 + rng.uniform(0, 0.3), c[2] + rng.uniform(-0.5, 0.5)])
        poses[i]['modelview'] = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(0.3, 1.3))
        poses[i]['projection'] = reference_projection(w, h)
    lights = lv.lights.fill_buffer_at(0.0)
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, len(poses))
    fb_plain, fb, prim = render_checked(batch, poses, lights)
    ro = raster.RasterOracle(lv)
    first = np.cumsum([0] + [int(d[3]) // 3 for d in lv.draws])
    sky_px = 0
    for i in range(len(poses)):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h, want_prim=True)
        assert np.array_equal(oprim, prim[i]) and np.array_equal(ofb, fb[i]) and np.array_equal(ofb, fb_plain[i]), i
        for di, d in enumerate(lv.draws):
            if d[0] == rd.KIND_SKY:
                sky_px += int(((oprim >= first[di]) & (oprim < first[di + 1])).sum())
    assert sky_px > 0.1 * len(poses) * w * h, sky_px


def test_explicit_stream(oracle_levels):
    """rdoom_batch_render on a caller-provided hipStream_t == on the default stream"""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    side = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(side)) == 0
    lv = oracle_levels(1)
    w, h, n = 128, 80, 4
    poses = sweep_poses(lv, n, w, h, seed=9)
    lights = lv.lights.fill_buffer_at(0.0)
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, n)
    batch.render(poses, lights)
    want = batch.read_framebuffer()
    batch.render(poses[::-1].copy(), lights, stream=side.value)  # different content in between
    assert hip.hipStreamSynchronize(side) == 0
    batch.render(poses, lights, stream=side.value)
    assert hip.hipStreamSynchronize(side) == 0
    assert np.array_equal(batch.read_framebuffer(), want)
    assert hip.hipStreamDestroy(side) == 0

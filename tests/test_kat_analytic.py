"""Analytic known-answer tests: expected pixels derived BY HAND (float64 geometry) from the reference's shader
source, not from any implementation in this repository.

Scene: camera at the origin looking down -z (identity modelview, the reference projection of
game/src/player.rs:84-89 / engine/src/projections.rs:93-101), in front of it
  * one wall quad at z = -d spanning x in [-1, 1], y in [-0.5, 0.5]  (wall vertex order and index pattern of
    game/src/level.rs:620-634, 676-680), textured with a 64x128 tile whose texel (x, y) = (x + 3 y) & 255;
  * one floor polygon at y = -0.41 (flat tile_uv = (-x*100, -z*100), game/src/level.rs:541), 64x64 tile with
    texel (x, y) = (5 x + y) & 255.
Expected pixel (static.vert:25-45, static.frag:18-28):
    uv    = mod(tile_uv, tile_size) + atlas_uv ; texel = atlas[floor(uv.y)][floor(uv.x)]
    dist  = clip.w = -z_eye ;  light = lights[a_light]/255 * 2 - min(1, 1 - 0.9/(dist + 0.9))
    row   = clamp(floor((1 - light) * 32), 0, 31) ;  out = COLORMAP[row][texel]
Pixels whose texel coordinate or colormap row lies within a guard band of a boundary are skipped (float32 vs
float64 may legitimately differ there); every other pixel must match exactly.  The same scene is rendered by
the C oracle (CPU test) and by the HIP kernels (GPU test)."""
import numpy as np
import pytest

from util import reference_projection

W, H = 320, 200
D_WALL = 2.5
FLOOR_Y = -0.41
LIGHT_WALL, LIGHT_FLOOR = 200, 144


def colormap():
    """a synthetic COLORMAP: row r maps index i -> (i * 7 + r * 13) & 255 (injective per row)"""
    i = np.arange(256, dtype=np.uint32)
    return np.stack([((i * 7 + r * 13) & 255).astype(np.uint8) for r in range(32)])


def kat_level():
    from oracle.wad_oracle import STATIC_VERTEX
    wall_atlas = np.zeros((128, 64), np.uint16)
    yy, xx = np.mgrid[0:128, 0:64]
    wall_atlas[:, :] = (xx + 3 * yy) & 255
    flat_atlas = np.zeros((64, 64), np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    flat_atlas[:, :] = (5 * xx + yy) & 255
    v = np.zeros(4 + 4, STATIC_VERTEX)
    # wall: (v1,low,s1,t1), (v2,low,s2,t1), (v2,high,s2,t2), (v1,high,s1,t2); s along x * 100, t top-down
    lo, hi = -0.5, 0.5
    for k, (x, y, s, t) in enumerate([(-1, lo, 0.0, 100.0), (1, lo, 200.0, 100.0), (1, hi, 200.0, 0.0),
                                      (-1, hi, 0.0, 0.0)]):
        v[k]['a_pos'] = (x, y, -D_WALL)
        v[k]['a_tile_uv'] = (s, t)
        v[k]['a_atlas_uv'] = (0, 0)
        v[k]['a_tile_size'] = (64, 128)
        v[k]['a_num_frames'] = 1
        v[k]['a_light'] = 0
        v[k]['a_row_height'] = 128
    # floor: fan over a big square around the camera at y = FLOOR_Y, CCW seen from above
    for k, (x, z) in enumerate([(-4, 1), (4, 1), (4, -9), (-4, -9)]):
        v[4 + k]['a_pos'] = (x, FLOOR_Y, z)
        v[4 + k]['a_tile_uv'] = (-x * 100.0, -z * 100.0)
        v[4 + k]['a_atlas_uv'] = (0, 0)
        v[4 + k]['a_tile_size'] = (64, 64)
        v[4 + k]['a_num_frames'] = 1
        v[4 + k]['a_light'] = 1
    # floor fan incl. the degenerate first triangle (level.rs:636-645), wall quad (0,1,3),(1,2,3)
    idx = np.array([4, 4, 5, 4, 5, 6, 4, 6, 7, 0, 1, 3, 1, 2, 3], np.uint32)
    draws = np.array([[0, 0, 0, 9], [1, 0, 9, 6]], np.uint32)  # flats first, then walls (level.rs:445-470)
    lights = np.zeros(256, np.uint8)
    lights[0], lights[1] = LIGHT_WALL, LIGHT_FLOOR
    lvl = dict(static_vertices=v, static_indices=idx, sky_vertices=np.zeros((0, 3), np.float32),
               sky_indices=np.zeros(0, np.uint32), draws=draws, flat_atlas=flat_atlas, wall_atlas=wall_atlas,
               sky_texture=np.zeros((1, 1), np.uint16), sky_band=np.float32(0.0), colormap=colormap().reshape(-1),
               palette=np.zeros(768, np.uint8))
    return lvl, lights


def expected():
    """float64 prediction: (H, W) expected index, (H, W) bool 'safe to compare'."""
    proj = reference_projection(W, H).astype(np.float64).reshape(4, 4).T  # row-major P[r][c]
    fx, fy = proj[0, 0], proj[1, 1]
    cm = colormap()
    lvl, _ = kat_level()
    out = np.zeros((H, W), np.uint8)
    safe = np.zeros((H, W), bool)
    guard = 0.02
    for iy in range(H):
        for ix in range(W):
            nx, ny = (ix + 0.5) / (W / 2) - 1.0, (iy + 0.5) / (H / 2) - 1.0
            dx, dy = nx / fx, ny / fy  # eye ray (dx, dy, -1) * t
            hits = []
            # wall plane z = -D_WALL: t = D_WALL
            x, y = dx * D_WALL, dy * D_WALL
            if -1 < x < 1 and -0.5 < y < 0.5:
                u = (x + 1.0) * 100.0
                vv = (0.5 - y) * 100.0
                tex = lvl['wall_atlas']
                hits.append((D_WALL, u % 64.0, vv % 128.0, tex, LIGHT_WALL,
                             min(1 - abs(x), 0.5 - abs(y))))
            if dy < 0:  # floor plane
                t = FLOOR_Y / dy
                x, z = dx * t, -t
                if -4 < x < 4 and -9 < z < 1:
                    hits.append((t, (-x * 100.0) % 64.0, (-z * 100.0) % 64.0, lvl['flat_atlas'], LIGHT_FLOOR,
                                 min(4 - abs(x), z + 9, 1 - z)))
            if not hits:
                continue
            hits.sort(key=lambda h: h[0])
            dist, u, vv, tex, lightb, edge = hits[0]
            ok = edge > 0.02 and (len(hits) == 1 or hits[1][0] - hits[0][0] > 1e-3)
            fu, fv = u - np.floor(u), vv - np.floor(vv)
            ok &= guard < fu < 1 - guard and guard < fv < 1 - guard
            texel = int(tex[int(np.floor(vv)), int(np.floor(u))]) & 255
            light = lightb / 255.0 * 2.0 - min(1.0, 1.0 - 0.9 / (dist + 0.9))
            tt = (1.0 - light) * 32.0
            ok &= abs(tt - np.round(tt)) > 1e-3
            row = int(min(31, max(0, np.floor(tt))))
            out[iy, ix] = cm[row, texel]
            safe[iy, ix] = ok
    return out, safe


@pytest.fixture(scope='module')
def prediction():
    return expected()


def pose():
    mv = np.eye(4, dtype=np.float32).reshape(16)
    return mv, reference_projection(W, H)


def test_prediction_is_meaningful(prediction):
    out, safe = prediction
    assert safe.mean() > 0.35         # a large part of the frame is comparable (the rest is background or guard band)
    assert len(np.unique(out[safe])) > 100  # and it is not a flat colour


def test_oracle_matches_hand_derivation(prediction):
    from oracle import raster
    lvl, lights = kat_level()
    mv, pr = pose()
    fb = raster.RasterOracle(lvl).render(mv, pr, 0.0, lights, W, H)
    out, safe = prediction
    bad = (fb != out) & safe
    assert bad.sum() == 0, 'oracle differs from the hand derivation at %r' % (np.argwhere(bad)[:5],)
    # uncovered pixels (above the wall, beside it) are background 0
    assert fb[H - 1, 0] == 0


def test_colormap_row_vs_light_at_fixed_distance():
    """static.frag:24-26 at the wall's distance d = 2.5: dist_term = 1 - 0.9/3.4 = 0.73529..., so
    row = clamp(floor((1 - (2 L/255 - 0.73529)) * 32), 0, 31).  By hand: L=255 -> -8.47 -> 0; L=200 -> 5.33 -> 5;
    L=144 -> 19.39 -> 19; L=100 -> 30.43 -> 30; L=50 -> 42.98 -> 31 (clamped).  The synthetic COLORMAP is
    invertible, so the row the oracle used is decoded from the centre pixel of the wall."""
    from oracle import raster
    lvl, lights = kat_level()
    mv, pr = pose()
    ro = raster.RasterOracle(lvl)
    inv13 = pow(13, -1, 256)
    for L, row in [(255, 0), (200, 5), (144, 19), (100, 30), (50, 31)]:
        lights[0] = L
        fb = ro.render(mv, pr, 0.0, lights, W, H)
        iy, ix = H // 2 + 3, W // 2 + 5  # a pixel on the wall (the wall covers the frame centre)
        # texel at that pixel, by hand: x = ndc_x * d / fx, u = (x + 1) * 100 ; y likewise, v = (0.5 - y) * 100
        proj = reference_projection(W, H).astype(np.float64).reshape(4, 4).T
        x = ((ix + 0.5) / (W / 2) - 1.0) / proj[0, 0] * D_WALL
        y = ((iy + 0.5) / (H / 2) - 1.0) / proj[1, 1] * D_WALL
        u, v = ((x + 1.0) * 100.0) % 64.0, ((0.5 - y) * 100.0) % 128.0
        texel = (int(u) + 3 * int(v)) & 255
        got_row = ((int(fb[iy, ix]) - texel * 7) * inv13) & 255
        assert got_row == row, (L, got_row, row)


def test_light_table_kats():
    """wad/src/light.rs:113-115 (>>3, /31), :81-91 (fake contrast +-2/31, clamped), game/src/lights.rs:26-30
    ((x*255) as u8 truncates)"""
    from oracle import wad_oracle as wo
    F = np.float32
    assert wo.light_to_f32(255) == F(1.0) and wo.light_to_f32(160) == F(20) / F(31) and wo.light_to_f32(7) == F(0)
    assert wo.light_to_f32(256) > F(1.0)  # exceeds 1, clamped only at table fill
    base = wo.LightInfo(wo.light_to_f32(160))
    assert wo.with_contrast(base, True).level == F(20) / F(31) + F(2) / F(31)
    assert wo.with_contrast(base, False).level == F(20) / F(31) + F(-2) / F(31)
    assert wo.with_contrast(wo.LightInfo(F(1.0)), True).level == F(1.0)
    assert wo.with_contrast(wo.LightInfo(F(0.0)), False).level == F(0.0)
    lights = wo.Lights()
    assert lights.push(base) == 0 and lights.push(wo.LightInfo(F(1.0))) == 1 and lights.push(base) == 0
    table = lights.fill_buffer_at(0.0)
    assert table[0] == int(F(20) / F(31) * F(255)) == 164 and table[1] == 255 and table[2] == 0


@pytest.mark.gpu
def test_hip_matches_hand_derivation(prediction):
    import rust_doom_amd as rd
    lvl, lights = kat_level()
    mv, pr = pose()
    poses = np.zeros(1, rd.POSE)
    poses[0]['modelview'], poses[0]['projection'] = mv, pr
    batch = rd.Batch(rd.DeviceLevel(lvl), W, H, 1)
    from util import render_checked
    fb_plain, fb_ids, _prim = render_checked(batch, poses, lights)  # after a dirtying render (the view turned by a radian)
    assert np.array_equal(fb_plain, fb_ids)
    fb = fb_plain[0]
    out, safe = prediction
    bad = (fb != out) & safe
    assert bad.sum() == 0, 'HIP path differs from the hand derivation at %r' % (np.argwhere(bad)[:5],)


def test_zero_area_fan_triangles_never_win(oracle_levels):
    """The reference's floor / ceiling fans start with a triangle that repeats its first vertex (game/src/level.rs:636-645).  Since
    round 6 the triangle set-up S3-S5 runs in binary64 on the binary32 inputs: the products of S3 are exact there,
    the edge coefficients of a repeated vertex cancel EXACTLY, the determinant is 0 and S4 culls the triangle, as GL does.  (In
    binary32 the rounded products left noise of either sign: half of those triangles were set up, binned and walked -- 7 % of the
    visible triangles of the benchmark sweep -- and a few won pixels on the strength of that noise.)"""
    from oracle import raster
    poses = np.load(__import__('os').path.join(__import__('util').GOLDEN, 'poses.npy'))
    for index in (0, 3, 7):
        lv = oracle_levels(index)
        tri = np.asarray(lv.static_indices).reshape(-1, 3)
        degenerate = (tri[:, 0] == tri[:, 1]) | (tri[:, 1] == tri[:, 2]) | (tri[:, 0] == tri[:, 2])
        assert degenerate.sum() > 50   # every flat polygon brings one
        # static primitive ids = positions in the draw list; flats and walls are drawn from static_indices in draw order
        draws = np.asarray(lv.draws).reshape(-1, 4)
        first_prim = np.cumsum([0] + [int(c) // 3 for c in draws[:, 3]])
        for k in range(poses.shape[1]):
            p = poses[index, k]
            t = float(p[32])
            _fb, prim = raster.RasterOracle(lv).render(p[:16], p[16:32], t, lv.lights.fill_buffer_at(t), 320, 200, want_prim=True)
            for pid in np.unique(prim[prim != raster.NO_PRIM]):
                d = int(np.searchsorted(first_prim, pid, side='right') - 1)
                if int(draws[d, 0]) in (0, 1):   # flat / wall: indices into static_indices
                    i = int(draws[d, 2]) // 3 + int(pid - first_prim[d])
                    assert not degenerate[i], (index, k, int(pid))


@pytest.mark.gpu
def test_hip_culls_the_zero_area_fan_triangle():
    """the hand-built scene holds five triangles, one of them the fan's degenerate first: the renderer must call FOUR visible"""
    import rust_doom_amd as rd
    lvl, lights = kat_level()
    mv, pr = pose()
    poses = np.zeros(1, rd.POSE)
    poses[0]['modelview'], poses[0]['projection'] = mv, pr
    batch = rd.Batch(rd.DeviceLevel(lvl), W, H, 1)
    batch.render(poses, lights, timed=True)
    assert batch.render(poses, lights, timed=True)['visible_triangles'] == 4

"""The texture-rich stand-in (tools/mkwad.py build_wad(rich=True)): E1M1's geometry with a wall texture of its own on every linedef
side (out of 320) and its own flats in every sector (out of 192).  build_texture_atlas then yields a 4096 x 2048 wall atlas (16 MB as
u16) and build_flat_atlas 1024 x 512 (wad/src/tex.rs:168-333) -- a texel store several times one XCD's 4 MiB of L2, which the nine
default levels (512 x 512 + 256 x 256: 1.1 MB) never leave.  Same bar as every level: product == oracle byte for byte on every
array, committed digests, HIP == oracle on frames and winning primitive ids."""
import hashlib
import importlib

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster, wad_oracle
from util import META_PATH

synthetic = importlib.import_module('rust-doom_amd.synthetic')
ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture']
# sha256 of the generated IWAD and of the level's arrays (product == oracle == these): the stand-in may not drift silently
WAD_SHA = 'a1311361518068868f6f7d1446fe0c4198a96d2c5affb86f3d3962480ca8c9fa'
ARRAYS_SHA = '26eaf0532fa5f24ff17ea675b45236ebedc874256db3a89852d5ced3ebd02ace'


@pytest.fixture(scope='module')
def rich():
    path = synthetic.ensure_rich_wad()
    return path, wad_oracle.build_level(path, META_PATH, 0)


def arrays_digest(get):
    h = hashlib.sha256()
    for k in ARRAYS:
        h.update(k.encode() + b'\0' + np.ascontiguousarray(get(k)).tobytes())
    return h.hexdigest()


def test_rich_level_is_rich_and_host_builder_matches_oracle(rich):
    path, lv = rich
    assert hashlib.sha256(open(path, 'rb').read()).hexdigest() == WAD_SHA
    built = rd.Wad(path, META_PATH).build_level(0)
    d = built.desc
    assert d.wall_w * d.wall_h >= 2048 * 2048 and d.flat_w * d.flat_h >= 512 * 512
    got = built.arrays()
    for k in ARRAYS:
        assert np.asarray(got[k]).tobytes() == np.asarray(getattr(lv, k)).tobytes(), k
    assert arrays_digest(lambda k: got[k]) == ARRAYS_SHA
    # really rich: hundreds of distinct texture rectangles in use, against a few dozen on the default E1M1
    sv = got['static_vertices']
    assert len({(float(u), float(v)) for u, v in sv['a_atlas_uv']}) >= 300
    # and the SAME geometry as the default E1M1 (same generator seed): positions of the floor / ceiling polygons agree
    base = rd.Wad(synthetic.ensure_wad(), META_PATH).build_level(0)
    assert built.counters()['num_floor_polys'] == base.counters()['num_floor_polys']
    assert np.array_equal(built.floor_centroids(), base.floor_centroids())


@pytest.mark.gpu
@pytest.mark.parametrize('width,height,n', [(640, 400, 16), (1920, 1080, 3)])
def test_rich_level_frames(rich, width, height, n):
    from test_gpu_raster_parity import sweep_poses
    from util import render_checked
    path, lv = rich
    built = rd.Wad(path, META_PATH).build_level(0, gpu_tessellation=True)
    got = built.arrays()
    for k in ARRAYS:   # (device tessellation: byte-identical level arrays)
        assert np.asarray(got[k]).tobytes() == np.asarray(getattr(lv, k)).tobytes(), k
    poses = sweep_poses(lv, n, width, height, seed=23, time=0.6)
    lights = lv.lights.fill_buffer_at(0.6)
    batch = rd.Batch(rd.DeviceLevel(built), width, height, n)
    fb_plain, fb, prim = render_checked(batch, poses, lights)
    ro = raster.RasterOracle(lv)
    bad = []
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.6, lights, width, height, want_prim=True)
        d = int((oprim != prim[i]).sum()), int((ofb != fb[i]).sum()) + int((ofb != fb_plain[i]).sum())
        if d != (0, 0):
            bad.append((i, d))
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_rich_level_in_a_set_with_the_default_levels(rich, oracle_levels):
    """a level set whose texel store holds a 16 MB atlas next to small ones: texel bases far beyond 2^16"""
    from test_gpu_levelset import check_against_oracles, mixed_batch
    from util import render_checked
    _path, lv = rich
    levels = [oracle_levels(0), lv, oracle_levels(5)]
    w, h = 640, 400
    poses, lop, lights, _om = mixed_batch(levels, 4, w, h, seed=31, moving=False)
    batch = rd.Batch(rd.DeviceLevelSet(levels), w, h, len(poses))
    fb_plain, fb, prim = render_checked(batch, poses, lights, level_of_pose=lop)
    check_against_oracles(levels, poses, lop, lights, None, (fb_plain, fb), prim, w, h)

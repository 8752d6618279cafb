"""A level ten times E1M1 (7.2 k linedefs, 11.9 k segs, 3 k sub-sectors, 38 k static triangles, 350 objects):
larger than anything in DOOM / DOOM2 (BASELINE config 5 names MAP29, ~1.9 k linedefs; no DOOM2.WAD exists here).
Exercises what small levels do not: multi-block device tessellation, depth-sort key counts in the thousands per
pose, bin-entry counts near the global cap, 16-bit visibility words near their limit."""
import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster, wad_oracle
from util import META_PATH, ensure_big_wad

ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture']


@pytest.fixture(scope='module')
def big():
    path = ensure_big_wad()
    return path, wad_oracle.build_level(path, META_PATH, 0)


def same_arrays(built, lv):
    got = built.arrays()
    for k in ARRAYS:
        assert np.asarray(got[k]).tobytes() == np.asarray(getattr(lv, k)).tobytes(), k
    for t in (0.0, 1.7):
        assert built.lights_at(t).tobytes() == lv.lights.fill_buffer_at(t).tobytes()


def test_big_level_host_builder_matches_oracle(big):
    path, lv = big
    built = rd.Wad(path, META_PATH).build_level(0)
    assert built.counters()['num_static_tris'] > 30000 and int(lv.num_objects) > 300
    same_arrays(built, lv)


@pytest.mark.gpu
def test_big_level_device_tessellation(big):
    """SSECTOR->polygon and SEG->quad kernels over ~3 k sub-sectors / ~12 k segs: byte-identical level arrays"""
    path, lv = big
    same_arrays(rd.Wad(path, META_PATH).build_level(0, gpu_tessellation=True), lv)


@pytest.mark.gpu
@pytest.mark.parametrize('width,height,n', [(320, 200, 24), (1920, 1080, 3)])
def test_big_level_frames(big, width, height, n):
    from test_gpu_raster_parity import sweep_poses
    path, lv = big
    built = rd.Wad(path, META_PATH).build_level(0, gpu_tessellation=True)
    poses = sweep_poses(lv, n, width, height, seed=11, time=0.9)
    # a bird's-eye pose and a long diagonal: the most triangles in view this level offers
    lo, hi = lv.static_vertices['a_pos'].min(0), lv.static_vertices['a_pos'].max(0)
    from util import reference_projection, view_matrix
    poses[1]['modelview'] = view_matrix(((lo[0] + hi[0]) / 2, hi[1] + 30.0, (lo[2] + hi[2]) / 2), 0.3, -1.5)
    poses[2]['modelview'] = view_matrix((lo[0] - 1.0, hi[1] + 4.0, lo[2] - 1.0), -2.4, -0.25)
    lights = lv.lights.fill_buffer_at(0.9)
    batch = rd.Batch(rd.DeviceLevel(built), width, height, n)
    from util import render_checked
    fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render; without and with primitive ids
    ro = raster.RasterOracle(lv)
    bad = []
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.9, lights, width, height, want_prim=True)
        d = int((oprim != prim[i]).sum()), int((ofb != fb[i]).sum()) + int((ofb != fb_plain[i]).sum())
        if d != (0, 0):
            bad.append((i, d))
    assert not bad, bad[:10]


@pytest.mark.gpu
def test_big_level_4k_time_varying_with_per_pose_lights(big):
    """BASELINE config 5's own configuration on its stand-in: the 10 x E1M1 level at 3840 x 2160, pose i at its own time
    (animated flats / walls, scrolling walls) with its own light table fill_buffer_at(time) -- the line `bench.py --big
    --width 3840 --height 2160 --time-varying` measures.  Frames and winning primitive ids against the oracle."""
    import importlib
    sharding = importlib.import_module('rust-doom_amd.sharding')
    path, lv = big
    built = rd.Wad(path, META_PATH).build_level(0, gpu_tessellation=True)
    width, height, n = 3840, 2160, 4
    poses = sharding.pose_sweep(rd, built, n, width, height, first=37)
    times = np.array([(37 + 64 * i) / 35.0 for i in range(n)], np.float32)   # pose i of the sweep at time i / 35 s, spread out
    poses['time'] = times
    lights = np.stack([built.lights_at(float(t)) for t in times])
    assert len({li.tobytes() for li in lights}) > 1                          # the tables really differ
    batch = rd.Batch(rd.DeviceLevel(built), width, height, n)
    from util import render_checked
    fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render (other poses, other light tables)
    ro = raster.RasterOracle(lv)
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        assert lights[i].tobytes() == lv.lights.fill_buffer_at(float(times[i])).tobytes()
        return ro.render(poses[i]['modelview'], poses[i]['projection'], float(times[i]), lights[i], width, height, want_prim=True)

    with ThreadPoolExecutor(n) as ex:   # (the C oracle releases the GIL: one pose per host thread)
        want = list(ex.map(one, range(n)))
    for i, (ofb, oprim) in enumerate(want):
        assert int((oprim != prim[i]).sum()) == 0 and int((ofb != fb[i]).sum()) == 0 and int((ofb != fb_plain[i]).sum()) == 0, i
        assert (fb[i] != 0).mean() > 0.5

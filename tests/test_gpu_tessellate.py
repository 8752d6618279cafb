"""GPU: the tessellation kernels (SSECTOR -> convex polygon, SEG -> wall / sky quads) == the host walk, byte for
byte on every array of every level; product path end to end."""
import numpy as np
import pytest

import rust_doom_amd as rd
from util import META_PATH

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('index', range(9))
def test_device_tessellation_matches_host(wad_path, index):
    wad = rd.Wad(wad_path, META_PATH)
    host = wad.build_level(index, gpu_tessellation=False).arrays()
    dev = wad.build_level(index, gpu_tessellation=True).arrays()
    for k in host:
        assert np.asarray(host[k]).tobytes() == np.asarray(dev[k]).tobytes(), k


def test_product_path_end_to_end(wad_path):
    """Product path only (C++ loader -> device tessellation -> HBM -> kernels) vs a level built from
    scratch by the numpy oracle and rendered by the C oracle."""
    from oracle import raster, wad_oracle
    wad = rd.Wad(wad_path, META_PATH)
    built = wad.build_level(2, gpu_tessellation=True)
    level = rd.DeviceLevel(built)
    w, h, n = 256, 160, 6
    batch = rd.Batch(level, w, h, n)
    cents = built.floor_centroids()
    poses = np.zeros(n, rd.POSE)
    for i in range(n):
        c = cents[(i * 7) % len(cents)]
        poses[i] = rd.pose_look((c[0], c[1] + 0.41, c[2]), 1.1 * i, 0.1 * (i - 3), w, h, 0.5)
    lights = built.lights_at(0.5)
    from util import render_checked
    fb, fb_ids, _prim = render_checked(batch, poses, lights)  # after a dirtying render; both instantiations give the same frames
    assert np.array_equal(fb, fb_ids)
    olv = wad_oracle.build_level(wad_path, META_PATH, 2)
    ro = raster.RasterOracle(olv)
    for i in range(n):
        want = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.5, olv.lights.fill_buffer_at(0.5), w, h)
        assert np.array_equal(want, fb[i]), i

"""IWADs from OTHER seeds of the generator (tools/mkwad.py), each with three small random levels of a different size:
inputs none of the committed fixtures was produced from.
  * CPU: the product's C++ loader + builder against the numpy oracle, every array byte for byte;
  * GPU: poses of each level's sweep, HIP framebuffer and winning primitive ids against the C oracle, bit-exact."""
import importlib
import os
import sys

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster, wad_oracle
from test_host_builder_parity import ARRAYS
from util import META_PATH, ROOT

sharding = importlib.import_module('rust-doom_amd.sharding')
SEEDS = (7, 4242, 90210)


def _wad(tmp_path_factory, seed):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mkwad
    rng = np.random.RandomState(seed)
    specs = [('E1M%d' % (k + 1), ('gen', seed * 31 + k, int(rng.randint(24, 41)), int(rng.randint(4, 10)))) for k in range(3)]
    data, _ = mkwad.build_wad(seed, specs=specs)
    path = str(tmp_path_factory.mktemp('seed%d' % seed) / 'other.wad')
    open(path, 'wb').write(data)
    return path


@pytest.fixture(scope='module', params=SEEDS)
def other_wad(request, tmp_path_factory):
    return _wad(tmp_path_factory, request.param)


def test_builder_matches_oracle_on_other_seeds(other_wad):
    product = rd.Wad(other_wad, META_PATH)
    assert product.num_levels() == 3
    for index in range(3):
        got = product.build_level(index).arrays()
        want = wad_oracle.build_level(other_wad, META_PATH, index)
        for name in ARRAYS:
            a, b = got[name], np.asarray(getattr(want, name))
            assert a.shape == b.shape and a.tobytes() == b.tobytes(), (index, name)
        c = product.build_level(index).counters()
        for k, v in want.counters.items():
            assert c[k] == v, (index, k)


@pytest.mark.gpu
def test_hip_matches_oracle_on_other_seeds(other_wad):
    product = rd.Wad(other_wad, META_PATH)
    for index in range(3):
        built = product.build_level(index, gpu_tessellation=True)
        assert built.arrays()['static_vertices'].tobytes() == product.build_level(index).arrays()['static_vertices'].tobytes()
        ro = raster.RasterOracle(built.arrays())
        level = rd.DeviceLevel(built)
        for w, h, n, t in ((320, 200, 3, 0.0), (712, 296, 2, 3.3)):
            poses = sharding.pose_sweep(rd, built, n, w, h, first=17 * index, time=t)
            lights = built.lights_at(t)
            batch = rd.Batch(level, w, h, n)
            from util import render_checked
            fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render; without and with primitive ids
            for i in range(n):
                ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], t, lights, w, h, want_prim=True)
                assert np.array_equal(fb[i], ofb) and np.array_equal(fb_plain[i], ofb) and np.array_equal(prim[i], oprim), (index, w, i)

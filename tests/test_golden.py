"""Committed golden fixtures (tests/golden/digests.json + poses.npy, generator tests/golden/make_golden.py):
sha256 of every level array of the nine synthetic levels and of 27 oracle-rendered frames (palette-index
framebuffer + winning primitive ids).  Checked against (a) the oracle itself (regression pin), (b) the
product's C++ loader/builder, (c, GPU) the HIP renderer."""
import hashlib
import json
import os

import numpy as np
import pytest

import rust_doom_amd as rd
from util import GOLDEN, META_PATH, wad_digest

G = json.load(open(os.path.join(GOLDEN, 'digests.json')))
POSES = np.load(os.path.join(GOLDEN, 'poses.npy'))
W, H = G['width'], G['height']


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_synthetic_iwad_is_the_pinned_one():
    assert wad_digest() == G['wad_sha256']


@pytest.mark.parametrize('index', range(9))
def test_oracle_arrays_match_golden(oracle_levels, index):
    lv = oracle_levels(index)
    g = G['levels'][index]
    for k, d in g['arrays'].items():
        assert sha(getattr(lv, k)) == d, k
    assert dict(lv.counters) == g['counters'] and int(lv.num_objects) == g['num_objects']
    assert sha(lv.lights.fill_buffer_at(0.0)) == g['lights_t0'] and sha(lv.lights.fill_buffer_at(1.7)) == g['lights_t1.7']


@pytest.mark.parametrize('index', range(9))
def test_product_builder_matches_golden(wad_path, index):
    built = rd.Wad(wad_path, META_PATH).build_level(index)
    got = built.arrays()
    g = G['levels'][index]
    for k, d in g['arrays'].items():
        a = got[k]
        if k in ('sky_vertices',):
            a = np.asarray(a, np.float32).reshape(-1, 3)
        assert sha(a) == d, k
    assert sha(built.lights_at(0.0)) == g['lights_t0'] and sha(built.lights_at(1.7)) == g['lights_t1.7']


@pytest.mark.parametrize('index', [0, 1, 4, 8])
def test_oracle_frames_match_golden(oracle_levels, index):
    from oracle import raster
    lv = oracle_levels(index)
    ro = raster.RasterOracle(lv)
    for p, g in zip(POSES[index], G['levels'][index]['frames']):
        t = float(p[32])
        fb, prim = ro.render(p[:16], p[16:32], t, lv.lights.fill_buffer_at(t), W, H, want_prim=True)
        assert sha(fb) == g['fb'] and sha(prim) == g['prim']


@pytest.mark.gpu
@pytest.mark.parametrize('index', range(9))
def test_hip_frames_match_golden(wad_path, index):
    """product path end to end (C++ loader -> HBM -> kernels) against the committed digests"""
    built = rd.Wad(wad_path, META_PATH).build_level(index)
    level = rd.DeviceLevel(built)
    n = len(POSES[index])
    batch = rd.Batch(level, W, H, n)
    poses = np.zeros(n, rd.POSE)
    lights = np.zeros((n, 256), np.uint8)
    for i, p in enumerate(POSES[index]):
        poses[i]['modelview'], poses[i]['projection'], poses[i]['time'] = p[:16], p[16:32], p[32]
        lights[i] = built.lights_at(float(p[32]))
    from util import render_checked
    fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render; without and with primitive ids
    for i, g in enumerate(G['levels'][index]['frames']):
        assert sha(fb[i]) == g['fb'] and sha(fb_plain[i]) == g['fb'], (index, i)
        assert sha(prim[i]) == g['prim'], (index, i)

"""GPU: a bounded, seeded slice of the randomized sweeps tests/stress_parity.py and tests/stress_extreme_poses.py run at
length by hand (those scripts stay for long runs): every synthetic level, random poses / times / pitches / per-object
offsets, and poses the ordinary sweep rarely produces -- eyes within centimetres of walls and on floor planes, far
outside the level, straight up / down.  HIP vs oracle, palette-index framebuffers AND winning primitive ids, bit for bit.
About 700 poses in all."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from util import reference_projection, render_checked, view_matrix

pytestmark = pytest.mark.gpu
SIZES = [(640, 400), (324, 180), (1280, 720), (200, 120)]


def compare(lv, poses, lights, w, h, om=None):
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, len(poses))
    kw = {} if om is None else {'object_modelviews': om}
    fb_plain, fb, prim = render_checked(batch, poses, lights, **kw)  # after a dirtying render; without and with primitive ids
    ro = raster.RasterOracle(lv)

    def check(i):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h,
                               want_prim=True, object_modelviews=None if om is None else om[i])
        return int((ofb != fb[i]).sum()) + int((ofb != fb_plain[i]).sum()), int((oprim != prim[i]).sum())

    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(check, range(len(poses))))
    bad = [(i, r) for i, r in enumerate(res) if r != (0, 0)]
    assert not bad, bad[:6]
    return float(np.mean(prim != 0xFFFFFFFF))


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('index', range(9))
def test_random_poses_times_and_moving_objects(oracle_levels, index, seed):
    lv, n = oracle_levels(index), 20
    rng = np.random.RandomState(100 + index + 1000 * seed)
    w, h = SIZES[(index + seed) % len(SIZES)]
    tri = lv.static_vertices['a_pos'][lv.static_indices.reshape(-1, 3)].mean(1)
    n_obj = int(lv.num_objects)
    poses, om, lights = np.zeros(n, rd.POSE), np.zeros((n, n_obj, 16), np.float32), np.zeros((n, 256), np.uint8)
    for i in range(n):
        c = tri[rng.randint(len(tri))]
        eye = np.array([c[0] + rng.uniform(-0.4, 0.4), c[1] + rng.uniform(-0.1, 0.7), c[2] + rng.uniform(-0.4, 0.4)])
        view = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(-1.2, 1.2))
        t = float(rng.choice([0.0, rng.uniform(0, 30)]))
        poses[i]['modelview'], poses[i]['projection'], poses[i]['time'] = view, reference_projection(w, h), t
        lights[i] = lv.lights.fill_buffer_at(t)
        v64 = view.astype(np.float64).reshape(4, 4).T
        for o in range(n_obj):
            m = np.eye(4)
            m[1, 3] = 0.0 if (o == 0 or i % 2 == 0) else rng.uniform(-0.8, 0.8)
            om[i, o] = (v64 @ m).T.astype(np.float32).reshape(16)
    assert compare(lv, poses, lights, w, h, om) > 0.3


@pytest.mark.parametrize('index', range(9))
def test_extreme_poses(oracle_levels, index):
    lv = oracle_levels(index)
    w, h = [(640, 400), (1920, 1080), (324, 180), (1280, 720)][index % 4]
    n = 4 if (w, h) == (1920, 1080) else 20
    rng = np.random.RandomState(900 + index)
    verts = lv.static_vertices['a_pos']
    poses, lights = np.zeros(n, rd.POSE), np.zeros((n, 256), np.uint8)
    for i in range(n):
        v = verts[rng.randint(len(verts))].astype(np.float64)
        kind = i % 4
        if kind == 0:    # a hair's breadth from a vertex of the level
            eye = v + rng.uniform(-0.02, 0.02, 3)
        elif kind == 1:  # on the floor / ceiling plane itself, looking along it
            eye = v + np.array([rng.uniform(-0.3, 0.3), rng.choice([0.0, 1e-4, -1e-4]), rng.uniform(-0.3, 0.3)])
        elif kind == 2:  # far outside, looking back
            eye = v + np.array([rng.uniform(-40, 40), rng.uniform(5, 60), rng.uniform(-40, 40)])
        else:
            eye = v + rng.uniform(-0.5, 0.5, 3)
        pitch = rng.choice([rng.uniform(-1.57, 1.57), 1.5707, -1.5707, 0.0])
        t = float(rng.choice([0.0, rng.uniform(0, 30)]))
        poses[i]['modelview'] = view_matrix(eye, rng.uniform(0, 2 * np.pi), pitch)
        poses[i]['projection'], poses[i]['time'] = reference_projection(w, h), t
        lights[i] = lv.lights.fill_buffer_at(t)
    compare(lv, poses, lights, w, h)

"""One-off randomized stress on the 10 x E1M1 level (GPU box): tile lists of ~90 entries, i.e. the rasteriser's multi-batch
path.  Not collected by pytest; run as  python tests/stress_big_level.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import conftest  # noqa
import rust_doom_amd as rd
from oracle import raster, wad_oracle
from util import META_PATH, apply_stress_hooks, ensure_big_wad, reference_projection, render_checked, view_matrix
print('# hooks:', ' '.join(apply_stress_hooks()))
lv = wad_oracle.build_level(ensure_big_wad(), META_PATH, 0)
rng = np.random.RandomState(123)
tri = lv.static_vertices['a_pos'][lv.static_indices.reshape(-1, 3)].mean(1)
for (w, h, n) in [(640, 400, 160), (1920, 1080, 24)]:
    poses = np.zeros(n, rd.POSE); lights = np.zeros((n, 256), np.uint8)
    for i in range(n):
        c = tri[rng.randint(len(tri))]
        eye = np.array([c[0] + rng.uniform(-0.4, 0.4), c[1] + rng.uniform(-0.1, 3.0), c[2] + rng.uniform(-0.4, 0.4)])
        t = float(rng.choice([0.0, rng.uniform(0, 30)]))
        poses[i]['modelview'], poses[i]['projection'], poses[i]['time'] = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(-1.2, 1.2)), reference_projection(w, h), t
        lights[i] = lv.lights.fill_buffer_at(t)
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, n)
    fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render; without and with primitive ids
    assert np.array_equal(fb_plain, fb)
    ro = raster.RasterOracle(lv)
    def check(i):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h, want_prim=True)
        return int((ofb != fb[i]).sum()), int((oprim != prim[i]).sum())
    with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
        res = list(ex.map(check, range(n)))
    bad = [(i, r) for i, r in enumerate(res) if r != (0, 0)]
    print('big level %dx%d poses %d:' % (w, h, n), 'ok' if not bad else bad[:5])

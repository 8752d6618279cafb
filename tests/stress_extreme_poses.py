"""One-off randomized stress (GPU box) on poses the ordinary sweep rarely produces: eyes within centimetres of walls and
floors or outside the level, pitches up to straight up / down, so that triangles cross the eye plane, leave the depth
range inside a block and cover whole tiles at once.  Not collected by pytest; run as  python tests/stress_extreme_poses.py"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401,E402
import rust_doom_amd as rd  # noqa: E402
from oracle import raster, wad_oracle  # noqa: E402
from util import META_PATH, ensure_wad, reference_projection, view_matrix  # noqa: E402


def main():
    from util import apply_stress_hooks
    hooks = apply_stress_hooks()
    if hooks:
        print('# hooks:', ' '.join(hooks))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
    total_bad = 0
    for index, (w, h) in zip(range(9), [(640, 400), (1920, 1080), (324, 180), (1280, 720)] * 3):
        if (w, h) == (1920, 1080):
            n_here = max(8, n // 8)
        else:
            n_here = n
        lv = wad_oracle.build_level(ensure_wad(), META_PATH, index)
        verts = lv.static_vertices['a_pos']
        poses = np.zeros(n_here, rd.POSE)
        lights = np.zeros((n_here, 256), np.uint8)
        for i in range(n_here):
            v = verts[rng.randint(len(verts))].astype(np.float64)
            kind = i % 4
            if kind == 0:    # a hair's breadth from a vertex of the level
                eye = v + rng.uniform(-0.02, 0.02, 3)
            elif kind == 1:  # on the floor / ceiling plane itself, looking along it
                eye = v + np.array([rng.uniform(-0.3, 0.3), rng.choice([0.0, 1e-4, -1e-4]), rng.uniform(-0.3, 0.3)])
            elif kind == 2:  # far outside, looking back
                eye = v * 1.0 + np.array([rng.uniform(-40, 40), rng.uniform(5, 60), rng.uniform(-40, 40)])
            else:
                eye = v + rng.uniform(-0.5, 0.5, 3)
            pitch = rng.choice([rng.uniform(-1.57, 1.57), 1.5707, -1.5707, 0.0])
            t = float(rng.choice([0.0, rng.uniform(0, 30)]))
            poses[i]['modelview'] = view_matrix(eye, rng.uniform(0, 2 * np.pi), pitch)
            poses[i]['projection'], poses[i]['time'] = reference_projection(w, h), t
            lights[i] = lv.lights.fill_buffer_at(t)
        batch = rd.Batch(rd.DeviceLevel(lv), w, h, n_here)
        from util import render_checked
        fb_plain, fb, prim = render_checked(batch, poses, lights)  # after a dirtying render; without and with primitive ids
        assert np.array_equal(fb_plain, fb)
        ro = raster.RasterOracle(lv)

        def check(i):
            ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h,
                                   want_prim=True)
            return int((ofb != fb[i]).sum()), int((oprim != prim[i]).sum())

        with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
            res = list(ex.map(check, range(n_here)))
        bad = [(i, r) for i, r in enumerate(res) if r != (0, 0)]
        total_bad += len(bad)
        covered = float(np.mean(prim != 0xFFFFFFFF))
        print('level %d %dx%d poses %d covered %.2f: %s' % (index, w, h, n_here, covered, 'ok' if not bad else bad[:6]))
    sys.exit(1 if total_bad else 0)


if __name__ == '__main__':
    main()

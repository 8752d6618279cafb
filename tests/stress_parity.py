"""One-off randomized stress (GPU box): every synthetic level x random poses / times / resolutions / object offsets,
HIP vs oracle, framebuffers and winning primitives.  Not collected by pytest; run as
    python tests/stress_parity.py [poses_per_level] [seed] [width height]
Prints one line per level and exits non-zero on any mismatch."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    sizes = [(640, 400), (324, 180), (1280, 720), (200, 120)]
    if len(sys.argv) > 4:  # one frame size for every level, e.g. 1920 1080
        sizes = [(int(sys.argv[3]), int(sys.argv[4]))]
    total_bad = 0
    for index in range(9):
        lv = wad_oracle.build_level(ensure_wad(), META_PATH, index)
        w, h = sizes[index % len(sizes)]
        tri = lv.static_vertices['a_pos'][lv.static_indices.reshape(-1, 3)].mean(1)
        n_obj = int(lv.num_objects)
        poses = np.zeros(n, rd.POSE)
        om = np.zeros((n, n_obj, 16), np.float32)
        lights = np.zeros((n, 256), np.uint8)
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
        batch = rd.Batch(rd.DeviceLevel(lv), w, h, n)
        # (each checked render follows one of the poses in reverse order: the batch's scratch holds another frame's state)
        batch.render(poses[::-1].copy(), lights[::-1].copy(), object_modelviews=om[::-1].copy())
        batch.render(poses, lights, object_modelviews=om)  # without primitive ids: the path bench.py times
        fb_plain = batch.read_framebuffer()
        batch.enable_primitive_ids()
        batch.render(poses[::-1].copy(), lights[::-1].copy(), object_modelviews=om[::-1].copy())
        batch.render(poses, lights, object_modelviews=om)
        fb, prim = batch.read_framebuffer(), batch.read_primitive_ids()
        ro = raster.RasterOracle(lv)

        def check(i):
            ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h,
                                   want_prim=True, object_modelviews=om[i])
            return int((ofb != fb[i]).sum()) + int((ofb != fb_plain[i]).sum()), int((oprim != prim[i]).sum())

        with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
            res = list(ex.map(check, range(n)))
        bad = [(i, r) for i, r in enumerate(res) if r != (0, 0)]
        total_bad += len(bad)
        print('level %d %dx%d poses %d objects %d: %s' % (index, w, h, n, n_obj, 'ok' if not bad else 'MISMATCH %r' % bad[:6]))
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())

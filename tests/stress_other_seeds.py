"""One-off randomized stress over LEVELS rather than poses: IWADs from many generator seeds (tools/mkwad.py), three random
levels each, none of them an input any committed fixture was produced from.  Per level:
  * CPU half: the product's C++ loader + builder against the numpy oracle, every array byte for byte (and the counters);
  * with a GPU: the device tessellation against the CPU one, then random poses / times, HIP framebuffers and winning
    primitives against the C oracle, bit-exact.
Not collected by pytest; run as
    python tests/stress_other_seeds.py [n_seeds] [first_seed] [poses_per_level] [--cpu-only]
Prints one line per level and exits non-zero on any mismatch."""
import os
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401,E402
import rust_doom_amd as rd  # noqa: E402
from oracle import raster, wad_oracle  # noqa: E402
from test_host_builder_parity import ARRAYS  # noqa: E402
from util import META_PATH, ROOT, reference_projection, view_matrix  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, 'tools'))
import mkwad  # noqa: E402

SIZES = [(320, 200), (712, 296), (1280, 720), (200, 120), (964, 540)]


def make_wad(seed, directory):
    rng = np.random.RandomState(seed)
    specs = [('E1M%d' % (k + 1), ('gen', seed * 31 + k, int(rng.randint(20, 61)), int(rng.randint(3, 19)))) for k in range(3)]
    data, _ = mkwad.build_wad(seed, specs=specs)
    path = os.path.join(directory, 'seed%d.wad' % seed)
    with open(path, 'wb') as f:
        f.write(data)
    return path


def builder_mismatches(product, path, index):
    got = product.build_level(index).arrays()
    want = wad_oracle.build_level(path, META_PATH, index)
    bad = [name for name in ARRAYS
           if got[name].shape != np.asarray(getattr(want, name)).shape or got[name].tobytes() != np.asarray(getattr(want, name)).tobytes()]
    c = product.build_level(index).counters()
    bad += ['counter ' + k for k, v in want.counters.items() if c[k] != v]
    return bad, want


def render_mismatches(product, index, lv, n, rng, size):
    w, h = size
    built = product.build_level(index, gpu_tessellation=True)
    if built.arrays()['static_vertices'].tobytes() != np.asarray(lv.static_vertices).tobytes():
        return ['device tessellation']
    tri = lv.static_vertices['a_pos'][lv.static_indices.reshape(-1, 3)].mean(1)
    poses = np.zeros(n, rd.POSE)
    lights = np.zeros((n, 256), np.uint8)
    for i in range(n):
        c = tri[rng.randint(len(tri))]
        eye = np.array([c[0] + rng.uniform(-0.4, 0.4), c[1] + rng.uniform(-0.1, 0.7), c[2] + rng.uniform(-0.4, 0.4)])
        t = float(rng.choice([0.0, rng.uniform(0, 30)]))
        poses[i]['modelview'] = view_matrix(eye, rng.uniform(0, 2 * np.pi), rng.uniform(-1.2, 1.2))
        poses[i]['projection'], poses[i]['time'] = reference_projection(w, h), t
        lights[i] = lv.lights.fill_buffer_at(t)
    batch = rd.Batch(rd.DeviceLevel(built), w, h, n)
    # (each checked render follows one of the poses in reverse order: the batch's scratch holds another frame's state)
    batch.render(poses[::-1].copy(), lights[::-1].copy())
    batch.render(poses, lights)  # without primitive ids: the path bench.py times
    fb_plain = batch.read_framebuffer()
    batch.enable_primitive_ids()
    batch.render(poses[::-1].copy(), lights[::-1].copy())
    batch.render(poses, lights)
    fb, prim = batch.read_framebuffer(), batch.read_primitive_ids()
    ro = raster.RasterOracle(lv)

    def check(i):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], float(poses[i]['time']), lights[i], w, h, want_prim=True)
        return int((ofb != fb[i]).sum()) + int((ofb != fb_plain[i]).sum()), int((oprim != prim[i]).sum())

    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(check, range(n)))
    return ['pose %d: %d pixels, %d winners' % (i, a, b) for i, (a, b) in enumerate(res) if (a, b) != (0, 0)]


def main():
    from util import apply_stress_hooks
    hooks = apply_stress_hooks()
    if hooks:
        print('# hooks:', ' '.join(hooks))
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    cpu_only = '--cpu-only' in sys.argv
    n_seeds = int(args[0]) if len(args) > 0 else 8
    first = int(args[1]) if len(args) > 1 else 1000
    n = int(args[2]) if len(args) > 2 else 16
    total_bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(first, first + n_seeds):
            path = make_wad(seed, tmp)
            product = rd.Wad(path, META_PATH)
            rng = np.random.RandomState(seed)
            for index in range(3):
                bad, lv = builder_mismatches(product, path, index)
                size = SIZES[(seed + index) % len(SIZES)]
                if not bad and not cpu_only:
                    bad = render_mismatches(product, index, lv, n, rng, size)
                total_bad += len(bad)
                print('seed %d level %d: %d static triangles, %d objects, %s: %s' % (
                    seed, index, len(lv.static_indices) // 3, int(lv.num_objects),
                    'builder only' if cpu_only else '%d poses at %dx%d' % (n, size[0], size[1]),
                    'ok' if not bad else 'MISMATCH %r' % bad[:6]), flush=True)
    return 1 if total_bad else 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""Generates tests/golden/digests_shapes.json from the ORACLE on the "shapes" IWAD (tools/mkwad.py build_wad(5150, shapes=True):
TEXTURE2, duplicated lump names, MAPxx markers, sprite lumps with paired rotations, textures of many overlapping patches --
tests/test_iwad_shapes.py).  Same layout as digests.json; the poses are make_golden.level_poses' (seeded), not stored.

    python tests/golden/make_golden_shapes.py        # rewrites the fixture
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import mkwad  # noqa: E402
from make_golden import ARRAYS, H, TIMES, W, level_poses, sha  # noqa: E402
from oracle import raster, wad_oracle  # noqa: E402
from util import GOLDEN, META_PATH  # noqa: E402

SEED = 5150
SPECS = [('MAP%02d' % (k + 1), ('gen', SEED * 31 + k, 28 + 6 * k, 5 + 2 * k)) for k in range(3)]


def build_shapes_wad(path):
    data, _ = mkwad.build_wad(SEED, specs=SPECS, shapes=True)
    with open(path, 'wb') as f:
        f.write(data)
    return hashlib.sha256(data).hexdigest()


def main():
    path = os.path.join('/tmp', 'rdoom_shapes_%d.wad' % os.getpid())
    out = {'wad_sha256': build_shapes_wad(path), 'width': W, 'height': H, 'levels': []}
    for index in range(len(SPECS)):
        lv = wad_oracle.build_level(path, META_PATH, index)
        ro = raster.RasterOracle(lv)
        frames = []
        for p in level_poses(lv, index):
            t = float(p[32])
            fb, prim = ro.render(p[:16], p[16:32], t, lv.lights.fill_buffer_at(t), W, H, want_prim=True)
            frames.append({'fb': sha(fb), 'prim': sha(prim), 'covered': int((fb != 0).sum())})
        out['levels'].append({'arrays': {a: sha(getattr(lv, a)) for a in ARRAYS}, 'counters': lv.counters, 'frames': frames})
    os.remove(path)
    with open(os.path.join(GOLDEN, 'digests_shapes.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote digests_shapes.json', out['wad_sha256'], TIMES)


if __name__ == '__main__':
    main()

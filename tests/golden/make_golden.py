#!/usr/bin/env python3
"""Generates tests/golden/digests.json + tests/golden/poses.npy from the ORACLE (oracle/wad_oracle.py +
oracle/raster_oracle.c) on the seeded synthetic IWAD.  Committed fixtures; tests/test_golden.py checks the
oracle, the product's C++ builder and the HIP renderer against them.

The reference (cristicbz/rust-doom) cannot be built or run here (no rustc, no GL) and ships no golden
vectors for this path, so these digests pin OUR restatement against regressions, not the reference binary.

    python tests/golden/make_golden.py        # rewrites the fixtures
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import raster, wad_oracle  # noqa: E402
from util import GOLDEN, META_PATH, ensure_wad, reference_projection, view_matrix, wad_digest  # noqa: E402

ARRAYS = ['static_vertices', 'static_indices', 'sky_vertices', 'sky_indices', 'decor_vertices', 'decor_indices',
          'draws', 'flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture', 'colormap', 'palette']
W, H, POSES_PER_LEVEL, TIMES = 320, 200, 3, (0.0, 0.0, 1.7)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def level_poses(lv, index):
    """3 deterministic poses per level: the spawn view + two seeded views from triangle centroids."""
    rng = np.random.RandomState(1000 + index)
    out = np.zeros((POSES_PER_LEVEL, 33), np.float32)
    sp = np.array([float(x) for x in lv.start_pos])
    ntri = len(lv.static_indices) // 3
    for i in range(POSES_PER_LEVEL):
        if i == 0:
            eye, yaw, pitch = sp + [0, 0.12, 0], float(lv.start_yaw), 1e-8
        else:
            t = rng.randint(ntri)
            c = lv.static_vertices['a_pos'][lv.static_indices[3 * t:3 * t + 3]].mean(0)
            eye = np.array([c[0] + rng.uniform(-0.3, 0.3), c[1] + rng.uniform(0.2, 0.6), c[2] + rng.uniform(-0.3, 0.3)])
            yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.5, 0.5)
        out[i, :16] = view_matrix(eye, yaw, pitch)
        out[i, 16:32] = reference_projection(W, H)
        if i == 0:
            # the spawn view in the REFERENCE'S OWN ARITHMETIC (binary32 Decomposed / Quaternion, player.rs:124-131 +
            # renderer.rs:78-87) -- BASELINE config 2's pose; it differs from the float64 helper above by a few units in the last
            # place.  From the oracle's own transcription (oracle/camera.py: numpy binary32 + glibc sinf / cosf / tanf), NOT from
            # the library under test: tests/test_pose_helpers.py holds rdoom_pose_from_player to it bit for bit
            from oracle import camera
            out[i, :16], out[i, 16:32] = camera.pose_from_player(sp, float(lv.start_yaw), 1e-8, W, H)
        out[i, 32] = TIMES[i]
    return out


def main():
    wad = ensure_wad()
    n_levels = 9
    digests = {'wad_sha256': wad_digest(), 'width': W, 'height': H, 'levels': []}
    all_poses = np.zeros((n_levels, POSES_PER_LEVEL, 33), np.float32)
    for index in range(n_levels):
        lv = wad_oracle.build_level(wad, META_PATH, index)
        entry = {'arrays': {k: sha(getattr(lv, k)) for k in ARRAYS}, 'counters': dict(lv.counters),
                 'num_objects': int(lv.num_objects), 'lights_t0': sha(lv.lights.fill_buffer_at(0.0)),
                 'lights_t1.7': sha(lv.lights.fill_buffer_at(1.7)), 'frames': []}
        poses = level_poses(lv, index)
        all_poses[index] = poses
        ro = raster.RasterOracle(lv)
        for p in poses:
            t = float(p[32])
            fb, prim = ro.render(p[:16], p[16:32], t, lv.lights.fill_buffer_at(t), W, H, want_prim=True)
            entry['frames'].append({'fb': sha(fb), 'prim': sha(prim), 'covered': int((prim != raster.NO_PRIM).sum())})
        digests['levels'].append(entry)
        print('level %d: %s' % (index, [f['covered'] for f in entry['frames']]))
    os.makedirs(GOLDEN, exist_ok=True)
    np.save(os.path.join(GOLDEN, 'poses.npy'), all_poses)
    with open(os.path.join(GOLDEN, 'digests.json'), 'w') as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    print('wrote', os.path.join(GOLDEN, 'digests.json'))


if __name__ == '__main__':
    main()

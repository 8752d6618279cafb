"""GPU: the 10 x E1M1 level (MAP29 stand-in, 38 262 triangles) at 320x200 -- a large level at a small frame, the case the tile-list
budget was not sized for until round 6 (entry_cap scaled with the frame only: split lists store an entry once per quadrant, and
such a pose could overflow into the every-tile-scans-every-triangle path without anybody noticing).  Asserts that no pose
overflows (rdoom_batch_path_stats) and that the frames equal the oracle's; then the same poses with a forced overflow
(hook entry_cap) must give the same bytes through the record-list path."""
import importlib
import os

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from util import META_PATH, ensure_big_wad

pytestmark = pytest.mark.gpu
sharding = importlib.import_module('rust-doom_amd.sharding')
W, H, N = 320, 200, 48


def test_large_level_at_a_small_frame_keeps_its_tile_lists():
    built = rd.Wad(ensure_big_wad(), META_PATH).build_level(0, gpu_tessellation=True)
    level = rd.DeviceLevel(built)
    poses = sharding.pose_sweep(rd, built, N, W, H)
    lights = built.lights_at(0.0)
    batch = rd.Batch(level, W, H, N)
    batch.render(poses[::-1].copy(), lights)  # other frames first: the checked render meets their scratch
    batch.render(poses, lights)
    s = batch.path_stats()
    assert s['poses'] == N and s['bins_overflowed_poses'] == 0, s
    assert s['split_tiles'] > 0  # the horizon tiles of this level hold hundreds of entries: lists per quadrant
    fb = batch.read_framebuffer()
    sample = np.zeros((N, 33), np.float32)
    sample[:, :16] = poses['modelview']
    sample[:, 16:32] = poses['projection']
    sample[:, 32] = poses['time']
    want = raster.RasterOracle(built.arrays()).render_batch(sample, np.tile(lights, (N, 1)), W, H, threads=os.cpu_count() or 1)
    bad = [(i, int((want[i] != fb[i]).sum())) for i in range(N) if not np.array_equal(want[i], fb[i])]
    assert not bad, 'poses (index, differing pixels): %r' % bad[:8]
    try:  # every pose through the fallback (no bins: each tile scans the pose's records near to far): same bytes
        rd.debug_set('entry_cap', 64)
        b2 = rd.Batch(level, W, H, N)
        b2.render(poses, lights)
        assert b2.path_stats()['bins_overflowed_poses'] == N
        assert np.array_equal(b2.read_framebuffer(), fb)
    finally:
        rd.debug_set('reset', 0)

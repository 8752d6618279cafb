"""Child process of tests/test_gpu_debug_paths.py: renders a small sweep through the C ABI with the test hooks
(rdoom_debug_set: name=value arguments after the four numbers) the parent asked for and checks it against the oracle."""
import sys

import numpy as np

import conftest  # noqa: F401  (sys.path)
import rust_doom_amd as rd
from oracle import raster, wad_oracle
from test_gpu_raster_parity import sweep_poses
from util import META_PATH, ensure_wad


def main():
    index, w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    for setting in sys.argv[5:]:
        name, value = setting.split('=')
        rd.debug_set(name, int(value))
    lv = wad_oracle.build_level(ensure_wad(), META_PATH, index)
    poses = sweep_poses(lv, n, w, h, seed=11, time=0.4)
    lights = lv.lights.fill_buffer_at(0.4)
    # Every checked render follows a render of OTHER poses into the same batch: visibility words, quadrant table and tile
    # lists then hold another frame's values wherever the checked render does not write them (the rasteriser leaves out
    # the visibility words of quadrants its table describes) -- a reader of a stale word gets a wrong record, not a lucky one.
    other = sweep_poses(lv, n, w, h, seed=12, time=0.4)
    batch = rd.Batch(rd.DeviceLevel(lv), w, h, n)
    batch.render(other, lights)
    batch.render(poses, lights)  # the path without primitive ids (the one bench.py times)
    fb_plain = batch.read_framebuffer()
    batch.enable_primitive_ids()
    batch.render(other, lights)
    t = batch.render(poses, lights, timed=True)
    fb, prim = batch.read_framebuffer(), batch.read_primitive_ids()
    ro = raster.RasterOracle(lv)
    bad = 0
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.4, lights, w, h, want_prim=True)
        bad += int((ofb != fb[i]).sum()) + int((oprim != prim[i]).sum()) + int((ofb != fb_plain[i]).sum())
    print('RESULT bad=%d fixups=%d' % (bad, t['fixup_pixels']))
    return 0 if bad == 0 else 1


if __name__ == '__main__':
    sys.exit(main())

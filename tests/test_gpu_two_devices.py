"""GPU, two or more devices in one process (skipped on a one-GPU box): batches are bound to their level's device
(rdoom_batch_create / render / finish / read_* all make it current), so a host thread may drive several GPUs in turn and
read any batch back whatever device happens to be current.  Round 2's advisor found the read paths acting on the
caller's current device."""
import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from test_gpu_raster_parity import sweep_poses

pytestmark = pytest.mark.gpu


def test_batches_on_two_devices_interleaved(oracle_levels):
    if rd.device_count() < 2:
        pytest.skip('needs two GPUs')
    lv = oracle_levels(0)
    w, h, n = 320, 200, 4
    poses = sweep_poses(lv, n, w, h, seed=5, time=0.0)
    lights = lv.lights.fill_buffer_at(0.0)
    batches = []
    for dev in (0, 1):
        rd.set_device(dev)
        level = rd.DeviceLevel(lv)
        batch = rd.Batch(level, w, h, n)
        batches.append((level, batch))
    rd.set_device(0)
    batches[1][1].enable_primitive_ids()      # allocates on device 1 although device 0 is current
    batches[1][1].render(poses, lights)
    batches[0][1].render(poses[::-1].copy(), lights)
    rd.set_device(1)
    fb0 = batches[0][1].read_framebuffer()    # device 1 is current: the read must still wait for / copy from device 0
    rd.set_device(0)
    fb1, prim1 = batches[1][1].read_framebuffer(), batches[1][1].read_primitive_ids()
    ro = raster.RasterOracle(lv)
    for i in range(n):
        ofb, oprim = ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h, want_prim=True)
        assert np.array_equal(fb1[i], ofb) and np.array_equal(prim1[i], oprim)
        assert np.array_equal(fb0[n - 1 - i], ofb)

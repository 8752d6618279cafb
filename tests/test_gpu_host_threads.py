"""GPU: the library driven from several HOST threads (SURVEY 8(b) "Threading": "a rdoom_level* is
immutable after create => shareable; a rdoom_batch* / stream context is single-owner; one host thread per GPU").  This is
the shape a Rust host would use -- one process, a std::thread per device / stream -- rather than one process per GPU.
  * ONE rdoom_level shared by T = 4 threads, each with its own rdoom_batch and its own hipStream_t: concurrent render /
    finish / read_framebuffer for several rounds (ctypes releases the GIL around every call), every frame compared with the
    oracle;
  * rdoom_last_error is thread-local: an error provoked on one thread is not seen by another;
  * one thread per DEVICE when the box has at least two (skipped otherwise; tests/test_gpu_two_devices.py drives two devices
    from one thread).
The host-only half (rdoom_wad_open / build_level on T threads) runs under ThreadSanitizer in tests/test_host_threads.py."""
import ctypes
import threading

import numpy as np
import pytest

import rust_doom_amd as rd
from oracle import raster
from test_gpu_raster_parity import sweep_poses
from util import dirtying_poses

pytestmark = pytest.mark.gpu
T, ROUNDS = 4, 3


def test_shared_level_four_threads_own_batches_and_streams(oracle_levels):
    lv = oracle_levels(3)
    w, h, n = 320, 200, 5
    level = rd.DeviceLevel(lv)                       # immutable after create: shared by every thread
    lights = lv.lights.fill_buffer_at(0.25)
    ro = raster.RasterOracle(lv)
    hip = ctypes.CDLL('libamdhip64.so')
    work, want = [], []
    for t in range(T):
        poses = sweep_poses(lv, n, w, h, seed=100 + t, time=0.25)
        work.append(poses)
        want.append(np.stack([ro.render(p['modelview'], p['projection'], 0.25, lights, w, h) for p in poses]))
    errors, barrier = [], threading.Barrier(T)

    def worker(t):
        try:
            rd.set_device(level_device)              # (hipSetDevice is per host thread)
            stream = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
            batch = rd.Batch(level, w, h, n)         # single-owner: this thread's own scratch
            for r in range(ROUNDS):
                barrier.wait()                       # all threads render at the same time
                batch.render(dirtying_poses(work[t]), lights, stream=stream.value)
                batch.render(work[t], lights, stream=stream.value)
                batch.finish()
                fb = batch.read_framebuffer()
                if not np.array_equal(fb, want[t]):
                    errors.append((t, r, int((fb != want[t]).sum())))
            batch.close()
            assert hip.hipStreamDestroy(stream) == 0
        except Exception as e:  # noqa: BLE001  (reported by the main thread)
            errors.append((t, repr(e)))
            barrier.abort()

    level_device = 0
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors


def test_read_back_does_not_wait_for_another_threads_stream(oracle_levels):
    """rdoom_batch_finish / rdoom_batch_read_* wait for THEIR batch's last render, on the stream it was queued on -- not for the
    device (round-4 review, "boundary warts": they used to call hipDeviceSynchronize, so a thread-per-stream host stalled on other
    threads' streams at every read-back).  Thread A keeps its stream busy with eight long renders (1080p x 1024 poses, 4-5 ms each)
    and then waits for them (rdoom_batch_finish); once the first two are queued the main thread renders a small batch on another
    stream and reads it back.  That read-back must return well before A's work is done -- by the clock: A's forty milliseconds
    against a few for the small batch, whose kernels share the device with A's -- and the small frames must be right."""
    import time
    lv = oracle_levels(0)
    level = rd.DeviceLevel(lv)
    lights = lv.lights.fill_buffer_at(0.0)
    hip = ctypes.CDLL('libamdhip64.so')
    big_n, w, h, n = 1024, 256, 160, 3
    big = rd.Batch(level, 1920, 1080, big_n)
    big_poses = np.resize(sweep_poses(lv, 64, 1920, 1080, seed=5), big_n)
    small = rd.Batch(level, w, h, n)
    poses = sweep_poses(lv, n, w, h, seed=6)
    ro = raster.RasterOracle(lv)
    want = np.stack([ro.render(p['modelview'], p['projection'], 0.0, lights, w, h) for p in poses])
    sa, sb = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(sa)) == 0 and hip.hipStreamCreate(ctypes.byref(sb)) == 0
    big.render(big_poses, lights, stream=sa.value)   # warm-up: first-use allocations and the one-off constant upload happen here
    small.render(poses, lights, stream=sb.value)
    big.finish(), small.finish()
    queued, errors, t = threading.Event(), [], {}

    def thread_a():
        try:
            rd.set_device(0)
            t['a_start'] = time.perf_counter()
            for k in range(8):
                big.render(big_poses, lights, stream=sa.value)   # (asynchronous; the two-deep staging lets the host run two ahead)
                if k == 1:
                    queued.set()
            big.finish()
            t['a_done'] = time.perf_counter()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            queued.set()

    th = threading.Thread(target=thread_a)
    th.start()
    assert queued.wait(120) and not errors, errors
    small.render(poses, lights, stream=sb.value)
    fb = small.read_framebuffer()
    t['small_done'] = time.perf_counter()
    th.join(120)
    assert not errors, errors
    assert np.array_equal(fb, want)
    a_ms, small_ms = (t['a_done'] - t['a_start']) * 1e3, (t['small_done'] - t['a_start']) * 1e3
    print('thread A busy for %.1f ms; the other stream\'s read-back returned after %.1f ms' % (a_ms, small_ms))
    assert t['small_done'] < t['a_done'] and small_ms < 0.7 * a_ms, (small_ms, a_ms)
    big.close(), small.close()
    assert hip.hipStreamDestroy(sa) == 0 and hip.hipStreamDestroy(sb) == 0


def test_bench_threads_launcher_on_this_gpu():
    """`bench.py --gpus 2 --launcher threads`: one process, a host thread per GPU through the C ABI only (INTEGRATION.md's shape
    for a Rust host) -- with one GPU present both threads wrap onto it; strong scaling: the two threads render the two halves of
    ONE batch.  The line must account for every pose and say that it is not a scaling measurement."""
    import json
    import os
    import subprocess
    import sys
    from util import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher', 'threads', '--scaling', 'strong', '--poses', '48',
                          '--width', '640', '--height', '400', '--steps', '3', '--warmup', '1'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['scaling'] == 'strong'
    assert [r[:2] for r in line['config']['pose_ranges']] == [[0, 24], [24, 48]]
    assert len(line['per_thread_ms_per_step']) == 2
    if rd.device_count() < 2:
        assert line['gpus_present'] == rd.device_count() and 'not a scaling measurement' in line['note']


def test_last_error_is_thread_local(oracle_levels):
    lib = rd.lib()
    lib.rdoom_last_error.restype = ctypes.c_char_p
    seen = {}
    first, second = threading.Event(), threading.Event()

    def failing():
        count = ctypes.c_int32()
        assert lib.rdoom_batch_finish(None) != 0     # BAD_ARG: "null argument" on THIS thread
        seen['failing'] = lib.rdoom_last_error()
        first.set()
        second.wait(60)
        seen['failing_again'] = lib.rdoom_last_error()   # still this thread's message after the other thread's calls
        del count

    def clean():
        first.wait(60)
        seen['clean_before'] = lib.rdoom_last_error()    # nothing failed on this thread
        n = ctypes.c_int32()
        assert lib.rdoom_device_count(ctypes.byref(n)) == 0
        assert lib.rdoom_debug_set(b'no such hook', 1) != 0
        seen['clean_after'] = lib.rdoom_last_error()
        second.set()

    a, b = threading.Thread(target=failing), threading.Thread(target=clean)
    a.start(), b.start()
    a.join(120), b.join(120)
    assert seen['failing'] and b'null' in seen['failing']
    assert not seen['clean_before']
    assert b'no such hook' in seen['clean_after']
    assert seen['failing_again'] == seen['failing']


def test_one_host_thread_per_device(oracle_levels):
    if rd.device_count() < 2:
        pytest.skip('needs two GPUs (the driver\'s box has one)')
    lv = oracle_levels(1)
    w, h, n = 256, 160, 4
    lights = lv.lights.fill_buffer_at(0.0)
    ro = raster.RasterOracle(lv)
    errors = []

    def worker(dev):
        try:
            rd.set_device(dev)
            level = rd.DeviceLevel(lv)               # one copy of the level per device (replicated)
            batch = rd.Batch(level, w, h, n)
            poses = sweep_poses(lv, n, w, h, seed=40 + dev)
            for _ in range(ROUNDS):
                batch.render(poses, lights)
                fb = batch.read_framebuffer()
                for i in range(n):
                    if not np.array_equal(fb[i], ro.render(poses[i]['modelview'], poses[i]['projection'], 0.0, lights, w, h)):
                        errors.append((dev, i))
        except Exception as e:  # noqa: BLE001
            errors.append((dev, repr(e)))

    threads = [threading.Thread(target=worker, args=(d,)) for d in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
    assert not errors, errors

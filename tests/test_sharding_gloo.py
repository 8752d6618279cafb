"""N > 1 path on CPU: two processes over gloo (world_size 2).  Poses shard; nothing is exchanged on the data
path, so what must hold is (1) the contiguous ranges partition the batch, (2) each rank generating ITS slice of
the seeded sweep yields exactly the slice of the single-process sweep (bench.py's weak-scaling shards are
disjoint slices of one sweep), (3) the barrier + max-over-ranks timing reduction works."""
import hashlib
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rust_doom_amd as rd
from util import META_PATH, ensure_wad

sharding = importlib.import_module('rust-doom_amd.sharding')

N_TOTAL, W, H = 64, 320, 200


def test_shard_ranges_partition():
    for n in (0, 1, 7, 64, 1024, 1025):
        for world in (1, 2, 3, 4, 8):
            r = [sharding.shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[g][1] == r[g + 1][0] for g in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
        lo, hi = sharding.shard_range(N_TOTAL, rank, world)
        poses = sharding.pose_sweep(rd, built, hi - lo, W, H, first=lo)
        # no data-path collective: each rank keeps its frames; here only digests travel, for the check
        digest = hashlib.sha256(poses.tobytes()).digest()
        t = torch.tensor(list(digest), dtype=torch.uint8)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        dist.barrier()
        slowest = sharding.max_over_ranks(1.0 + rank, dist, 'cpu')
        if rank == 0:
            np.save(os.path.join(out_dir, 'digests.npy'), np.stack([g.numpy() for g in gathered]))
            np.save(os.path.join(out_dir, 'slowest.npy'), np.array([slowest]))
    finally:
        dist.destroy_process_group()


def test_world_size_2_shards_equal_the_single_process_sweep(tmp_path):
    world = 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
    full = sharding.pose_sweep(rd, built, N_TOTAL, W, H, first=0)
    got = np.load(tmp_path / 'digests.npy')
    for g in range(world):
        lo, hi = sharding.shard_range(N_TOTAL, g, world)
        want = np.frombuffer(hashlib.sha256(full[lo:hi].tobytes()).digest(), np.uint8)
        assert np.array_equal(got[g], want), g
    assert float(np.load(tmp_path / 'slowest.npy')[0]) == 2.0  # max over ranks of (1 + rank)


def test_sweep_is_deterministic_and_slices_compose():
    built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
    a = sharding.pose_sweep(rd, built, 16, W, H, first=0)
    b = sharding.pose_sweep(rd, built, 8, W, H, first=8)
    assert a[8:].tobytes() == b.tobytes()
    assert sharding.max_over_ranks(3.5) == 3.5


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` with no torch.distributed environment launches two ranks itself (the way the driver
    calls it); --dry-run stops short of the device work so this runs on a host without a GPU."""
    import json
    import subprocess
    import sys
    from util import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    for scaling, want in (('strong', [[0, 12], [12, 24]]), ('weak', [[0, 24], [24, 48]])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--poses', '24',
                              '--width', '64', '--height', '40', '--scaling', scaling],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
        assert line['n_gpus'] == 2 and line['scaling'] == scaling
        assert [r[:2] for r in line['pose_ranges']] == want
        built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
        for lo, hi, digest in line['pose_ranges']:
            full = sharding.pose_sweep(rd, built, hi - lo, 64, 40, first=lo)
            assert hashlib.sha256(full.tobytes()).hexdigest()[:16] == digest


def test_bench_threads_launcher_partitions_like_the_process_launcher():
    """`--launcher threads`: one process, one host thread per GPU through the C ABI only (the shape INTEGRATION.md gives a Rust
    host).  Its dry run must hand every thread the pose range -- and the very poses -- the process launcher hands its ranks."""
    import json
    import subprocess
    import sys
    from util import ROOT
    lines = {}
    for launcher in ('processes', 'threads'):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--poses', '24', '--width', '64',
                              '--height', '40', '--scaling', 'strong', '--launcher', launcher], capture_output=True, text=True, timeout=600,
                             env={k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')})
        assert out.returncode == 0, out.stderr[-2000:]
        lines[launcher] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert lines['threads']['launcher'] == 'threads' and lines['threads']['n_gpus'] == 2
    assert lines['threads']['pose_ranges'] == lines['processes']['pose_ranges'] == [[0, 12, lines['threads']['pose_ranges'][0][2]],
                                                                                    [12, 24, lines['threads']['pose_ranges'][1][2]]]


def test_stream_plans():
    """one level: its poses as S sub-batches; several levels alternate over the stream pool as one batch each"""
    import sys
    from util import ROOT
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_streams(0, 1) == 3 and bench.resolve_streams(0, 9) == 3 and bench.resolve_streams(4, 9) == 4
    assert bench.parts_per_level(1, 2) == 2 and bench.parts_per_level(1, 3) == 3 and bench.parts_per_level(9, 3) == 1 and bench.parts_per_level(2, 4) == 2 and bench.parts_per_level(1, 1) == 1

"""Pose-batch sharding across the GPUs of one node: poses are independent units, so a batch
is cut into contiguous ranges, one per rank (one process per GPU); the level is replicated; NO data-path
collective exists.  torch.distributed is used only for the timing barrier and the max-over-ranks reduction
of the benchmark (RCCL on the GPU box, gloo in the CPU tests)."""
import math

import numpy as np

SWEEP_SEED = 0x19931210


def shard_range(n_total, rank, world):
    """Contiguous range [lo, hi) of rank `rank` out of `world`: [g*n/G, (g+1)*n/G)."""
    if not (0 <= rank < world):
        raise ValueError('rank %d outside 0..%d' % (rank, world - 1))
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def xorshift32(state):
    state ^= (state << 13) & 0xFFFFFFFF
    state ^= state >> 17
    state ^= (state << 5) & 0xFFFFFFFF
    return state & 0xFFFFFFFF


def pose_sweep(rd, built, n, width, height, first=0, seed=SWEEP_SEED, time=0.0):
    """SURVEY 8(d) config-3 pose generator (seeded, deterministic): poses [first, first+n) of the global sweep.
    Pose i: eye = centroid of a pseudo-random sub-sector floor polygon at floor + 0.41, yaw = 2 pi (i mod 1024)/1024
    + rnd, pitch = (rnd - 0.5) * 0.6.  Three xorshift32 draws per pose, so any slice can be generated alone."""
    cents = built.floor_centroids()
    poses = np.zeros(n, rd.POSE)
    s = seed
    for _ in range(first * 3):
        s = xorshift32(s)
    for k in range(n):
        i = first + k
        s = xorshift32(s)
        c = cents[s % len(cents)]
        s = xorshift32(s)
        yaw = 2.0 * math.pi * (i % 1024) / 1024.0 + s * 2.0 ** -32
        s = xorshift32(s)
        pitch = (s * 2.0 ** -32 - 0.5) * 0.6
        poses[k] = rd.pose_look((c[0], c[1] + 0.41, c[2]), yaw, pitch, width, height, time)
    return poses


def max_over_ranks(value, dist=None, device='cpu'):
    """max of a python float over all ranks (the benchmark's step time); identity without a process group"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

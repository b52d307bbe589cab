"""Multi-GPU path on CPU: world_size-2 gloo runs of the block sharding + result gather + membership all-reduce that
bench.py uses over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_matching_amd import sharding


def test_shard_blocks_partition():
    for G in (1, 2, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            ranges = [sharding.shard_blocks(G, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == G
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    lo, hi = sharding.shard_vectors(100000, 16384, 2, 1)
    assert (lo, hi) == (4 * 16384, 100000)
    assert sharding.global_indices([0, 5], 32768) == [32768, 32773]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank "computed" result ciphertexts for its own blocks: [blocks][2][1][N] of residues
    N, moduli = 64, [1152921504606584833]
    lo, hi = sharding.shard_blocks(6, world, rank)
    rng = np.random.default_rng(100 + rank)
    local = torch.from_numpy(rng.integers(0, moduli[0], size=(hi - lo, 2, 1, N), dtype=np.int64))
    got = sharding.gather_results(local.reshape(-1), dist, rank, world)
    if rank == 0:
        for r in range(world):
            rr = np.random.default_rng(100 + r)
            rlo, rhi = sharding.shard_blocks(6, world, r)
            want = rr.integers(0, moduli[0], size=(rhi - rlo, 2, 1, N), dtype=np.int64).reshape(-1)
            assert np.array_equal(got[r].numpy(), want)
    # membership: sum of per-rank partial ciphertexts modulo q
    part = torch.from_numpy(rng.integers(0, moduli[0], size=(2, N), dtype=np.int64))
    mine = part.clone()
    red = sharding.allreduce_membership_residues(part, moduli * 2, dist)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    tot = np.zeros((2, N), dtype=object)
    for p_ in parts:
        tot = tot + p_.numpy().astype(object)
    assert np.array_equal(red.numpy().astype(object), tot % moduli[0])
    q.put((rank, True))
    dist.destroy_process_group()


def test_gather_and_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == [0, 1]

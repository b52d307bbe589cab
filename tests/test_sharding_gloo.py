"""The multi-process sharded sender (image_matching_amd.sharding.DistDiagonalSender — the class bench.py --gpus N runs over
RCCL) on CPU: world-size-2 and -3 gloo groups drive the REAL host logic (block ranges from libhydia's hydia_shard_blocks,
query broadcast, padded gather of uneven shards into global block order, membership = local add-many -> integer reduce ->
mod q -> EvalSum) over a stand-in engine that does the per-rank ciphertext arithmetic with numpy on synthetic residues.
What the stand-in replaces is exactly the part that needs a GPU; tests/test_gpu_sharding.py runs the same class on real
contexts and requires bit-equality with the unsharded sender."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from image_matching_amd import sharding

Q = [1152921504606584833, 35184372744193]  # one 60-bit and one 45-bit prime: limb 0 and limb 1 of the fake results
N, SLOTS, NQ = 64, 32, 3


class FakeCt:
    def __init__(self, data, scale):
        self.data, self.scale = np.ascontiguousarray(data, dtype=np.uint64), scale

    def shape(self):
        c, p, l, _ = self.data.shape
        return c, p, l, self.scale

    def export(self):
        return self.data.copy()


class FakeContext:
    """numpy stand-in for the rank-local engine: same method names as image_matching_amd.Context"""
    N, slots, nQ, dim = N, SLOTS, NQ, 4

    def import_ct(self, data, scale):
        return FakeCt(data, scale)

    def add_many(self, ct):
        tot = np.zeros_like(ct.data[0], dtype=object)
        for x in ct.data:
            tot = tot + x.astype(object)
        for l in range(tot.shape[1]):
            tot[:, l] %= Q[l]
        return FakeCt(tot.astype(np.uint64)[None], ct.scale)

    def ct_mod_reduce(self, ct):
        for l in range(ct.data.shape[2]):
            ct.data[:, :, l] %= np.uint64(Q[l])

    def eval_sum(self, ct):  # stand-in for the rotate-and-add tree: any deterministic function of the reduced sum
        return FakeCt((ct.data * np.uint64(3)) % np.uint64(Q[0] if ct.data.shape[2] == 1 else 1 << 62), ct.scale)


def block_result(g, kind):
    """what `kind` gives for GLOBAL block g: [2][2 limbs][N] residues (depends on g only, so any sharding must agree)"""
    rng = np.random.default_rng(1000 * kind + g)
    return np.stack([rng.integers(0, Q[l], size=(2, N), dtype=np.uint64) for l in range(2)], axis=1)


def fake_rotation(query, i):
    """stand-in for rotation i of the query: a deterministic function of (query, i); rotation 0 is the query itself"""
    return query[0] if i == 0 else (query[0] * np.uint64(2 * i + 1) + np.uint64(i)) % np.uint64(Q[0])


class FakeSender:
    def __init__(self, lo, hi, query_check):
        self.lo, self.hi, self.query_check = lo, hi, query_check

    def _run(self, q, kind):
        assert np.array_equal(q.export(), self.query_check), "every rank must receive rank 0's query"
        return FakeCt(np.stack([block_result(g, kind) for g in range(self.lo, self.hi)]), 2.0 ** 45)

    def computeSimilarity(self, q):
        return self._run(q, 1)

    def indexScenario(self, q):
        return self._run(q, 2)

    # rotation-split loop A: a rank computes a range, the scenarios take the gathered set
    def rotateQueryRange(self, q, first, count):
        assert np.array_equal(q.export(), self.query_check), "every rank must receive rank 0's query"
        return FakeCt(np.stack([fake_rotation(q.export(), i) for i in range(first, first + count)]), q.scale)

    def _run_rot(self, rot, kind):
        want = np.stack([fake_rotation(self.query_check, i) for i in range(FakeContext.dim)])
        assert np.array_equal(rot.export(), want), "the gathered rotations must be the full set in rotation order"
        return FakeCt(np.stack([block_result(g, kind) for g in range(self.lo, self.hi)]), 2.0 ** 45)

    def computeSimilarityRotated(self, rot):
        return self._run_rot(rot, 1)

    def indexScenarioRotated(self, rot):
        return self._run_rot(rot, 2)


def test_shard_blocks_partition():
    for G in (1, 2, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            ranges = [sharding.shard_blocks(G, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == G
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert sharding.shard_vectors(100000, 16384, 2, 1) == (4 * 16384, 100000)
    assert sharding.shard_vectors(1024, 16384, 4, 2) == (1024, 1024)  # more ranks than blocks: empty shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, G, split, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cc = FakeContext()
    query = np.random.default_rng(5).integers(0, Q[0], size=(1, 2, NQ, N), dtype=np.uint64)
    n_total = G * SLOTS - 5  # ragged last block
    lo, hi = sharding.shard_blocks(G, world, rank)
    sender = sharding.DistDiagonalSender(cc, n_total, dist, rank, world, staging="host",
                                         make_sender=lambda c, n: FakeSender(lo, hi, query), rotation_split=split)
    assert sender.rotation_split == (split and len(sender.active) > 1)  # one rank with blocks: nothing to share out
    if sender.rotation_split:  # the ranks that hold blocks share the rotations 0 .. dim-1 between them, in rank order
        ranges = [sender.rot_ranges[r] for r in sender.active]
        assert ranges[0][0] == 0 and ranges[-1][1] == cc.dim and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert (sender.lo, sender.hi) == (lo, hi) and (sender.local is None) == (hi == lo)
    q = FakeCt(query, 2.0 ** 45) if rank == 0 else None
    for _ in range(2):  # second call reuses the cached buffers / metadata
        idx = sender.indexScenario(q)
        sim = sender.computeSimilarity(q)
        mem = sender.membershipScenario(q)
    if rank == 0:
        want_idx = np.stack([block_result(g, 2) for g in range(G)])
        want_sim = np.stack([block_result(g, 1) for g in range(G)])
        assert idx.shape() == (G, 2, 2, 2.0 ** 45) and np.array_equal(idx.export(), want_idx)  # global block order
        assert np.array_equal(sim.export(), want_sim)
        single = FakeContext()
        want_mem = single.eval_sum(single.add_many(FakeCt(want_idx, 2.0 ** 45)))  # what ONE context computes
        assert np.array_equal(mem.export(), want_mem.export())
    else:
        assert idx is None and sim is None and mem is None
    out.put((rank, True))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,G,split", [(2, 5, False), (2, 1, True), (3, 8, True), (2, 4, True), (3, 2, True)])
def test_dist_sender_host_logic(world, G, split):
    """split = loop A's rotations shared out over the ranks that hold blocks and all-gathered (SURVEY 8e option B); (2, 1): one of the
    two ranks holds no block — a single rank with blocks computes every rotation itself; (3, 2): the third rank holds no block and joins
    the all-gather with an empty range"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, G, split, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5)[0] for _ in range(world)) == list(range(world))


# ---- the mat-vec form of the resident databases: agreed by every rank, re-agreed when ANY rank's database changed
class FormContext(FakeContext):
    """FakeContext + what DistDiagonalSender reads the form off: db_kind / db_babies (0 = nothing enrolled) and the policy"""
    kind, babies = 0, 0

    def db_kind(self):
        return self.kind

    def db_babies(self):
        return self.babies

    def auto_babies(self, blocks):
        return self.dim


def _form_worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cc = FormContext()
    query = np.random.default_rng(5).integers(0, Q[0], size=(1, 2, NQ, N), dtype=np.uint64)
    G = 2
    lo, hi = sharding.shard_blocks(G, world, rank)
    holds = hi > lo
    # (a) the sender is built BEFORE anything is enrolled
    sender = sharding.DistDiagonalSender(cc, G * SLOTS, dist, rank, world, staging="host",
                                         make_sender=lambda c, n: FakeSender(lo, hi, query), rotation_split=True)
    assert sender._form in (None, (0, 0))
    q = FakeCt(query, 2.0 ** 45) if rank == 0 else None
    want = np.stack([block_result(g, 2) for g in range(G)])
    if holds:
        cc.kind, cc.babies = 5, cc.dim  # hoisted
    idx = sender.indexScenario(q)  # first call: every rank re-agrees (the block-less rank learns of it through the reduce)
    assert (idx is None) == (rank != 0) and (rank != 0 or np.array_equal(idx.export(), want))
    assert not sender.bsgs and sender.babies == cc.dim and sender.rotation_split == (len(sender.active) > 1)
    verdict = "ok"
    if case == "reenrol":  # (b) every block-holding rank re-enrols pre-rotated: the next call switches form by itself
        if holds:
            cc.kind, cc.babies = 6, 2
        idx = sender.indexScenario(q)
        assert (rank != 0 or np.array_equal(idx.export(), want)) and sender.bsgs and sender.babies == 2 and not sender.rotation_split
        if holds:
            cc.kind, cc.babies = 5, cc.dim
        sender.refresh_form()  # ... and the explicit form of the same
        assert not sender.bsgs and sender.rotation_split == (len(sender.active) > 1)
        idx = sender.indexScenario(q)
        assert rank != 0 or np.array_equal(idx.export(), want)
    elif case == "disagree":  # (c) more ranks than blocks, the two block-holding ranks end up with different forms: a clean error on EVERY rank
        if rank == 0:
            cc.kind, cc.babies = 6, 2
        try:
            sender.indexScenario(q)
            verdict = "no error"
        except ValueError as e:
            verdict = "ValueError" if "different mat-vec forms" in str(e) else "other: %s" % e
    out.put((rank, verdict))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "reenrol"), (3, "reenrol"), (3, "disagree")])
def test_dist_sender_form_changes_are_rank_symmetric(world, case):
    """a sender built before enrolment works at its first call; a re-enrolment in another form is picked up by every rank (world 3: the
    third rank holds no block and cannot see the change itself); disagreeing forms raise on every rank instead of hanging the others"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_form_worker, args=(r, world, port, case, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "a rank hung or failed"
    got = dict(out.get(timeout=5) for _ in range(world))
    assert got == {r: ("ValueError" if case == "disagree" else "ok") for r in range(world)}

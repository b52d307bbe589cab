"""GPU tests of the sharded DiagonalSender (run with -m gpu; one MI355X is enough): R shards — contexts that share the GPU —
must reproduce, bit for bit, the ciphertexts ONE context holding the whole database computes
(/root/reference/src/sender/sender_diag.cpp:28-30 blocks are independent; :46-49 membership tail), through both drivers:
the in-process group (libhydia hydia_group_*, what ./ImageMatching uses) and the one-process-per-GPU class bench.py uses
(here two ranks over gloo that both compute on GPU 0).  Also the re-key and handle-lifetime regressions of round 1's review."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def im():
    import image_matching_amd as im
    return im


def make_db(n, dim, planted, seed):
    rng = np.random.default_rng(seed)
    db = rng.integers(-99, 100, size=(n, dim)).astype(np.float64)
    for i in planted:
        db[i] = rng.integers(1, 4, size=dim)
    return db


def single_context_results(im, prm, n, db, planted, query, kseed=31, dseed=8, qseed=2, matvec="hoisted"):
    """matvec: the form of the mat-vec, fixed explicitly — "auto" would pick by the blocks a context holds, and a shard holds fewer
    than the single context it is compared with (results are bit-identical between runs of the SAME form)"""
    cc = im.Context(prm, 0)
    cc.set_matvec(matvec)
    cc.keygen(kseed)
    a = db.copy()
    im.DiagonalEnroller(cc, n).serializeDB(a, seed=dseed)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    q = receiver.encryptQuery(query, seed=qseed, nonce=9)
    idx = sender.indexScenario(q)
    found = receiver.decryptIndex(idx)  # (small vector_dim: a random row may legitimately clear 0.44)
    assert set(planted) <= set(found) and (cc.dim < 512 or found == sorted(planted))
    mem = sender.membershipScenario(q)
    out = dict(found=found, member=receiver.decryptMembership(mem), q=q.export(), qscale=q.shape()[3], sim=sender.computeSimilarity(q).export(), idx=idx.export(),
               mem=mem.export(), normalised=a,
               db_cts={t: cc.db_export_ct(t) for t in sorted({0, min(cc.dim + 1, cc.db_stats()[1] - 1), cc.db_stats()[1] - 1})})
    del q, idx, mem
    cc.close()
    return out


def check_group(im, prm, devices, n, db, planted, query, want, splits=(True, False), matvec="hoisted"):
    grp = im.ShardGroup(devices, prm)
    grp.ctx0.set_matvec(matvec)  # the group enrols every shard in shard 0's form
    grp.keygen(31)
    b = db.copy()
    im.ShardedDiagonalEnroller(grp, n).serializeDB(b, seed=8)
    assert np.array_equal(b, want["normalised"])  # normalised in place, like one enroller
    cc0 = grp.ctx0
    S, dim = cc0.slots, cc0.dim
    G = -(-n // S)
    # the shards together hold exactly the single context's ciphertexts (same nonces): spot-check through the shard that owns them
    for t, data in want["db_cts"].items():
        g = t // dim
        r = next(r for r in range(len(devices)) if im.shard_blocks(G, len(devices), r)[0] <= g < im.shard_blocks(G, len(devices), r)[1])
        lo = im.shard_blocks(G, len(devices), r)[0]
        assert np.array_equal(grp.shard_ctx(r).db_export_ct(t - lo * dim), data), (t, r)
    first, cnt = grp.shard_range(len(devices) - 1)
    assert first + cnt == n or cnt == 0
    sender, receiver = im.ShardedDiagonalSender(grp, n), im.DiagonalReceiver(cc0, n)
    q = receiver.encryptQuery(query, seed=2, nonce=9)
    assert np.array_equal(q.export(), want["q"])
    # loop A shared out over the shards and exchanged (default, SURVEY 8e option B), then recomputed by every shard (option A)
    for split in splits:
        grp.set_rotation_split(split)
        assert np.array_equal(sender.computeSimilarity(q).export(), want["sim"])
        idx = sender.indexScenario(q)
        assert np.array_equal(idx.export(), want["idx"])          # global block order
        assert receiver.decryptIndex(idx) == want["found"]         # global indices
        mem = sender.membershipScenario(q)
        assert np.array_equal(mem.export(), want["mem"])           # add-many -> integer sum -> mod q -> EvalSum
        assert receiver.decryptMembership(mem) == want["member"] and (want["member"] or not planted)
        del idx, mem
    del q
    grp.close()


@pytest.mark.parametrize("n,planted", [(5000, [0, 1024, 4999]), (700, [13]), (3100, [])])
def test_shard_group_bit_identical_small_ring(im, n, planted):
    """N = 2^11 (1024-vector blocks): 5, 1 and 4 blocks over 2, 3 and 5 shards — uneven and empty shards included."""
    prm = im.default_params(log_n=11, vector_dim=64)
    db = make_db(n, 64, planted, n)
    query = np.ones(64)
    for matvec in ("hoisted", "bsgs"):  # both forms of the mat-vec: shards and single context agree bit for bit within a form
        want = single_context_results(im, prm, n, db, planted, query, matvec=matvec)
        for R in (2, 3, 5):
            check_group(im, prm, [0] * R, n, db, planted, query, want, matvec=matvec, splits=(True, False) if matvec == "hoisted" else (True,))


@pytest.mark.parametrize("R", [2, 4, 8])
def test_shard_group_2p17_full_ring(im, R, full_ring_2p17):
    """BASELINE config 4's database (2^17 vectors = 8 blocks, N = 2^15) over R shards on one GPU: index ciphertexts identical to
    the single-context run, membership ciphertext identical after the mod."""
    n, db, planted, query, want = full_ring_2p17
    check_group(im, im.default_params(), [0] * R, n, db, planted, query, want)


@pytest.fixture(scope="module")
def full_ring_2p17(im):
    n = 1 << 17
    planted = [5, 16384 * 3 + 7, n - 1]
    rng = np.random.default_rng(17)
    db = rng.integers(-99, 100, size=(n, 512), dtype=np.int8).astype(np.float64)
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    return n, db, planted, query, single_context_results(im, im.default_params(), n, db, planted, query)


RANK_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch.distributed as dist
import image_matching_amd as im
from test_gpu_sharding import make_db
rank, world, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(sys.argv[2])
dist.init_process_group("gloo")
prm = im.default_params(log_n=11, vector_dim=64)
cc = im.Context(prm, 0)
cc.set_matvec("hoisted")
cc.keygen(31)
planted = [0, 1024, n - 1]
db = make_db(n, 64, planted, n)
enr = im.DistDiagonalEnroller(cc, n, rank, world)
enr.serializeDB(np.ascontiguousarray(db[enr.first:enr.last]), seed=8)
sender = im.DistDiagonalSender(cc, n, dist, rank, world, staging="host")
receiver = im.DiagonalReceiver(cc, n)
q = receiver.encryptQuery(np.ones(64), seed=2, nonce=9) if rank == 0 else None
sim, idx, mem = sender.computeSimilarity(q), sender.indexScenario(q), sender.membershipScenario(q)
if rank == 0:
    assert set(planted) <= set(receiver.decryptIndex(idx)) and receiver.decryptMembership(mem) is True
    np.savez(sys.argv[3], sim=sim.export(), idx=idx.export(), mem=mem.export())
dist.barrier()
dist.destroy_process_group()
cc.close()
'''


def test_dist_sender_two_ranks_on_one_gpu(im, tmp_path):
    """bench.py's multi-rank class with two real ranks (gloo, host staging; both contexts on GPU 0): rank 0's gathered index /
    similarity batches and reduced membership ciphertext equal the single-context ones bit for bit."""
    n = 5000
    prm = im.default_params(log_n=11, vector_dim=64)
    planted = [0, 1024, n - 1]
    db = make_db(n, 64, planted, n)
    want = single_context_results(im, prm, n, db, planted, np.ones(64))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    out = tmp_path / "rank0.npz"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n), str(out)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    for k in ("sim", "idx", "mem"):
        assert np.array_equal(got[k], want[k]), k


THREAD_DIST_SCRIPT = r'''
import os, sys, threading
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch
import image_matching_amd as im
from test_gpu_sharding import make_db, single_context_results


class ThreadDist:
    """torch.distributed stand-in for R ranks that are THREADS of this process sharing GPU 0, with NCCL's stream semantics: a collective
    is enqueued on the rank's own communication stream, which first waits for the rank's current stream; the call returns at once and
    the rank's current stream is made to wait for the collective (what Work.wait() does) — nothing blocks the host.  So the class under
    test must order its own stream, torch's stream and the collectives itself, exactly as under RCCL (which refuses two ranks on one
    GPU, so the real backend cannot run the world > 1 device-staging path on a one-GPU box)."""

    class ReduceOp:
        SUM = "sum"
        MAX = "max"

    def __init__(self, world):
        import threading
        import torch
        self.torch, self.world = torch, world
        self.bar = threading.Barrier(world)
        self.local = threading.local()
        self.slots = [None] * world
        self.comm = [torch.cuda.Stream() for _ in range(world)]

    def bind(self, rank):
        self.local.rank = rank

    def _run(self, fn):
        """publish -> everybody's operand is visible -> enqueue on the comm stream -> the others may move on"""
        torch, r = self.torch, self.local.rank
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        self.slots[r] = dict(self.slots[r], ready=ready)
        self.bar.wait()
        with torch.cuda.stream(self.comm[r]):
            for other in self.slots:
                self.comm[r].wait_event(other["ready"])   # the peers' operands are produced on THEIR current streams
            torch.cuda._sleep(50_000_000)                 # the collective completes LATE (~20 ms, longer than a local query): a consumer that does not wait reads stale data
            fn(r)
        cur.wait_stream(self.comm[r])
        done = torch.cuda.Event()
        done.record(self.comm[r])
        self.slots[r]["done"] = done
        self.bar.wait()
        for other in self.slots:                           # a peer must not overwrite its operand before every reader is through
            cur.wait_event(other["done"])
        self.bar.wait()

    def broadcast(self, t, src):
        self.slots[self.local.rank] = dict(t=t)
        self._run(lambda r: t.copy_(self.slots[src]["t"], non_blocking=True) if r != src else None)

    def gather(self, t, recv, dst):
        self.slots[self.local.rank] = dict(t=t)
        self._run(lambda r: [recv[k].copy_(self.slots[k]["t"], non_blocking=True) for k in range(self.world)] if r == dst else None)

    def reduce(self, t, dst, op):
        assert op == self.ReduceOp.SUM
        self.slots[self.local.rank] = dict(t=t)
        self._run(lambda r: [t.add_(self.slots[k]["t"]) for k in range(self.world) if k != dst] if r == dst else None)

    def all_reduce(self, t, op):
        assert op == self.ReduceOp.MAX
        self.slots[self.local.rank] = dict(t=t.clone())  # a snapshot: every rank reads every operand and overwrites its own
        self._run(lambda r: t.copy_(self.torch.stack([self.slots[k]["t"] for k in range(self.world)]).max(0).values))

    def all_gather_into_tensor(self, out, inp):
        m = inp.numel()
        self.slots[self.local.rank] = dict(t=inp)

        def fn(r):
            for k in range(self.world):
                if k != r:
                    out[k * m:(k + 1) * m].copy_(self.slots[k]["t"], non_blocking=True)
            if out[r * m:(r + 1) * m].data_ptr() != inp.data_ptr():
                out[r * m:(r + 1) * m].copy_(inp, non_blocking=True)
        self._run(fn)

    def all_gather(self, recv, t):
        self.slots[self.local.rank] = dict(t=t)
        self._run(lambda r: [recv[k].copy_(self.slots[k]["t"], non_blocking=True) for k in range(self.world)])

    def broadcast_object_list(self, objs, src):
        r = self.local.rank
        if r == src:
            self.obj = list(objs)
        self.bar.wait()
        objs[:] = self.obj
        self.bar.wait()


n, world, split, matvec = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5]
prm = im.default_params(log_n=11, vector_dim=64)
planted = sorted({0, min(1024, n - 1), n - 1})
db = make_db(n, 64, planted, n)
want = single_context_results(im, prm, n, db, planted, np.ones(64), matvec=matvec)
td = ThreadDist(world)
got, errors = {}, []


def rank_main(rank):
    try:
        td.bind(rank)
        with torch.cuda.stream(torch.cuda.Stream()):     # every rank has its own "current" stream, like separate processes
            cc = im.Context(prm, 0)
            cc.set_matvec(matvec)
            cc.keygen(31)
            enr = im.DistDiagonalEnroller(cc, n, rank, world)
            enr.serializeDB(np.ascontiguousarray(db[enr.first:enr.last]), seed=8)
            sender = im.DistDiagonalSender(cc, n, td, rank, world, staging="device", rotation_split=split)
            receiver = im.DiagonalReceiver(cc, n)
            for rep in range(2):
                q = receiver.encryptQuery(np.ones(64), seed=2, nonce=9) if rank == 0 else None
                sim, idx, mem = sender.computeSimilarity(q), sender.indexScenario(q), sender.membershipScenario(q)
                if rank == 0:
                    got[rep] = dict(sim=sim.export(), idx=idx.export(), mem=mem.export(), found=receiver.decryptIndex(idx),
                                    member=receiver.decryptMembership(mem))
                del q, sim, idx, mem
            td.bar.wait()
            cc.close()
    except Exception as e:  # noqa: BLE001
        errors.append((rank, repr(e)))
        td.bar.abort()


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=600)
assert not errors, errors
for rep in range(2):
    for k in ("sim", "idx", "mem"):
        assert np.array_equal(got[rep][k], want[k]), (rep, k)
    assert got[rep]["found"] == want["found"] and got[rep]["member"] is True
print("thread-dist ok")
'''


@pytest.mark.parametrize("n,world,split,matvec", [(5000, 3, True, "hoisted"), (1000, 3, True, "hoisted"), (5000, 2, True, "hoisted"),
                                                  (5000, 3, False, "hoisted"), (5000, 3, True, "bsgs")])
def test_dist_sender_device_staging_under_async_collectives(tmp_path, n, world, split, matvec):
    """DistDiagonalSender with staging="device" and world = 3 (n = 5000: uneven shards of 2 + 2 + 1 blocks of 1024 vectors; n = 1000: one
    block, ranks 1 and 2 own nothing): three threads, three
    contexts on GPU 0, collectives with NCCL's asynchronous stream semantics (ThreadDist in the script above: RCCL itself refuses two
    ranks on one GPU) that complete a few milliseconds LATE.  Rank 0's gathered similarity / index batches and its reduced membership
    ciphertext equal the single-context ones bit for bit, twice in a row (buffer reuse).  split = loop A's rotations shared out over
    the ranks and all-gathered (SURVEY 8e option B): world 3 takes the padded all_gather (64 rotations do not divide by 3), world 2 the
    in-place all_gather_into_tensor straight into the rotation buffer the mat-vec reads.  Runs in its own process (torch's HIP runtime
    next to the library's, started before anything forks)."""
    script = tmp_path / "thread_dist.py"
    script.write_text(THREAD_DIST_SCRIPT)
    r = subprocess.run([sys.executable, str(script), ROOT, str(n), str(world), "1" if split else "0", matvec], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "thread-dist ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_rotation_ranges_reassemble_loop_a(im):
    """hydia_rotate_query_range over any partition of 0 .. vector_dim-1 gives exactly the ciphertexts of hydia_rotate_query (full ring),
    and the *_rotated scenarios on them the ciphertexts of the plain scenarios — what the rotation-split multi-GPU sender relies on."""
    n = 20000
    cc = im.Context()
    cc.set_matvec("hoisted")  # the form whose loop A is worth sharing out (two blocks would otherwise pick baby-step / giant-step)
    cc.keygen(31)
    db = make_db(n, 512, [7, n - 1], 3)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=8)
    q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(512), seed=2, nonce=9)
    s = im.DiagonalSender(cc, n)
    full = s.rotateQuery(q)
    want = full.export()
    for parts in ([(0, 512)], [(0, 1), (1, 511)], [(0, 171), (171, 171), (342, 170)], [(64 * k, 64) for k in range(8)]):
        got = np.concatenate([s.rotateQueryRange(q, lo, cnt).export() for lo, cnt in parts])
        assert np.array_equal(got, want), parts
    assert np.array_equal(s.indexScenarioRotated(full).export(), s.indexScenario(q).export())
    assert np.array_equal(s.computeSimilarityRotated(full).export(), s.computeSimilarity(q).export())
    with pytest.raises(im.HydiaError):
        s.rotateQueryRange(q, 500, 13)  # past vector_dim
    del full, q
    cc.close()


def test_rekey_refreshes_loop_a_keys(im):
    """keygen(a), query, keygen(b) on the SAME context, query: must equal a fresh context keyed with b (loop A streams a packed
    shadow of the rotation keys that has to follow every re-key — round-1 advisor finding)."""
    prm = im.default_params(log_n=11, vector_dim=64)
    n = 1500
    db = make_db(n, 64, [3], 1)

    def run(cc):  # noqa: E306
        im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=8)
        q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(64), seed=2, nonce=9)
        s = im.DiagonalSender(cc, n)
        return s.rotateQuery(q).export(), s.indexScenario(q).export()
    cc = im.Context(prm, 0)
    cc.keygen(100)
    first = run(cc)
    cc.keygen(200)
    again = run(cc)
    fresh = im.Context(prm, 0)
    fresh.keygen(200)
    want = run(fresh)
    assert not np.array_equal(first[0], want[0])
    assert np.array_equal(again[0], want[0]) and np.array_equal(again[1], want[1])
    # the same through an imported key: overwrite rotation key 1 with another context's and expect that context's rotation
    other = im.Context(prm, 0)
    other.keygen(300)
    for r in range(1, 64):
        cc.import_eval_key(r, other.export_eval_key(r))
    cc.import_eval_key(0, other.export_eval_key(0))
    cc.import_public_key(other.export_public_key())
    assert np.array_equal(run(cc)[0], run(other)[0])
    for c in (cc, fresh, other):
        c.close()


def test_handles_outlive_context_close(im):
    """hydia_ct_free after hydia_ctx_destroy (round-1 advisor finding): the context is kept alive by its handles."""
    cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
    cc.keygen(1)
    ct = cc.encrypt(np.zeros((2, cc.slots)), seed=1)
    L = cc.L
    cc.close()
    assert ct.shape()[:3] == (2, 2, 12)
    del ct  # frees the handle, which completes the deferred destruction
    with pytest.raises(im.HydiaError):  # shape validation of device imports
        c2 = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
        try:
            c2.ct_from_device(1, 1, 7, 2, 1.0)
        finally:
            c2.close()
    with pytest.raises(im.HydiaError) as e:  # nonces are 40-bit
        c3 = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
        try:
            c3.keygen(1)
            c3.encrypt(np.zeros((1, c3.slots)), seed=1, nonce0=1 << 40)
        finally:
            c3.close()
    assert e.value.code == -1
    # two encryptions with default seeds differ (OS entropy), two with the same (seed, nonce) agree
    c4 = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
    c4.keygen(None)
    z = np.zeros((1, c4.slots))
    assert not np.array_equal(c4.encrypt(z).export(), c4.encrypt(z).export())
    assert np.array_equal(c4.encrypt(z, seed=5, nonce0=3).export(), c4.encrypt(z, seed=5, nonce0=3).export())
    c4.close()


def test_cli_sharded_matches_reference_answers(tmp_path):
    """./ImageMatching <2_10.dat> 5 with HYDIA_DEVICES=0,0 (two shards on the one GPU): the sharded roles behind the reference's
    driver flow give `true`, `[ 0 ]`."""
    exe = os.path.join(ROOT, "image_matching_amd", "ImageMatching")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    g = np.load(os.path.join(GOLDEN, "dataset_2_11.npz"))
    dat = tmp_path / "2_11.dat"
    with open(dat, "w") as f:
        f.write("%d\n" % int(g["n"]))
        f.write(" ".join(str(int(v)) for v in g["query"]) + " \n")
        for row in g["db"]:
            f.write(" ".join(str(int(v)) for v in row) + " \n")
    (tmp_path / "latency.csv").write_text("")
    out = subprocess.run([exe, str(dat), "5"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HYDIA_DEVICES="0,0", HYDIA_SEED="7"))
    assert out.returncode == 0, out.stderr
    assert "Membership scenario: true" in out.stdout and "Index scenario: [ 0 ]" in out.stdout
    row = (tmp_path / "latency.csv").read_text().strip().split(",")
    assert row[0] == "Diagonal" and row[1] == "2048" and row[10] == "true" and row[11] == "[ 0 ]"


def test_db_save_load_round_trip(im, tmp_path):
    """hydia_db_save / hydia_db_load: enrol, save, NEW context, load, and indexScenario is bit-identical (no re-enrolment);
    a file written for another prime chain or ring is refused."""
    prm = im.default_params(log_n=11, vector_dim=64)
    n, planted = 3000, [1, 2999]
    db = make_db(n, 64, planted, 5)
    cc = im.Context(prm, 0)
    cc.keygen(31)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=8)
    q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(64), seed=2, nonce=9)
    want = im.DiagonalSender(cc, n).indexScenario(q).export()
    path = tmp_path / "db.hydia"
    cc.db_save(path)
    stats = cc.db_stats()
    assert os.path.getsize(path) > stats[2]
    c2 = im.Context(prm, 0)
    c2.keygen(31)
    with pytest.raises(im.HydiaError):
        c2.db_save(tmp_path / "none.hydia")  # nothing resident
    c2.db_load(path)
    assert c2.db_stats() == stats
    q2 = c2.import_ct(q.export(), q.shape()[3])
    assert np.array_equal(im.DiagonalSender(c2, n).indexScenario(q2).export(), want)
    assert set(planted) <= set(im.DiagonalReceiver(c2, n).decryptIndex(im.DiagonalSender(c2, n).indexScenario(q2)))
    c3 = im.Context(im.default_params(log_n=12, vector_dim=64), 0)
    with pytest.raises(im.HydiaError):
        c3.db_load(path)
    with pytest.raises(im.HydiaError):
        c2.db_load(tmp_path / "missing.hydia")
    # the header is untrusted input (round-2 review): a ciphertext count that does not follow from the vector count, a truncated
    # file and trailing bytes are refused BEFORE anything is allocated or the resident database is touched
    raw = bytearray(path.read_bytes())
    hdr_ncts = 8 + 6 * 4 + 8   # magic, six u32 fields, n_vectors -> n_cts (u64)
    assert int.from_bytes(raw[hdr_ncts:hdr_ncts + 8], "little") == stats[1]
    for name, blob in (("count", raw[:hdr_ncts] + (stats[1] * 1000).to_bytes(8, "little") + raw[hdr_ncts + 8:]),
                       ("zero", raw[:hdr_ncts] + (0).to_bytes(8, "little") + raw[hdr_ncts + 8:]),
                       ("vectors", raw[:hdr_ncts - 8] + (1 << 40).to_bytes(8, "little") + raw[hdr_ncts:]),
                       ("short", raw[:len(raw) - 4096]), ("long", raw + b"\0" * 16)):
        bad = tmp_path / ("bad_%s.hydia" % name)
        bad.write_bytes(bytes(blob))
        with pytest.raises(im.HydiaError):
            c2.db_load(bad)
        assert c2.db_stats() == stats, name
        assert np.array_equal(im.DiagonalSender(c2, n).indexScenario(q2).export(), want), name
    del q, q2
    for c in (cc, c2, c3):
        c.close()


def test_shard_group_full_size_2p20(im):
    """BASELINE config 5's database (2^20 vectors, 64 blocks, 142.5 GiB resident) through the sharded sender in config 5's own shape,
    EIGHT shards of 8 blocks (all on the one GPU; loop A's rotations shared out over the shards and exchanged): index batch (global block order), decrypted global indices and the membership ciphertext equal the single-context run's."""
    n = 1 << 20
    planted = [0, 12345, n // 2 + 3, n - 1]
    rng = np.random.default_rng(2020)
    db = rng.integers(-99, 100, size=(n, 512), dtype=np.int8).astype(np.float64)
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    prm = im.default_params()
    cc = im.Context(prm, 0)
    cc.set_matvec("hoisted")  # one split on both sides: "auto" would give the 8-block shards 128 babies and the 64-block context 512
    cc.keygen(31)
    a = db.copy()
    im.DiagonalEnroller(cc, n).serializeDB(a, seed=8)
    del a
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    q = receiver.encryptQuery(query, seed=2, nonce=9)
    idx = sender.indexScenario(q)
    assert receiver.decryptIndex(idx) == planted
    want_idx, want_mem, want_q = idx.export(), sender.membershipScenario(q).export(), q.export()
    del q, idx
    cc.close()
    grp = im.ShardGroup([0] * 8, prm)  # config 5's own shape: 8 shards of 8 blocks (here all on the one GPU; keys resident once)
    grp.ctx0.set_matvec("hoisted")
    grp.keygen(31)
    im.ShardedDiagonalEnroller(grp, n).serializeDB(db.copy(), seed=8)
    assert grp.shard_range(0) == (0, n // 8) and grp.shard_range(7) == (7 * (n // 8), n // 8)
    gr, gs = im.DiagonalReceiver(grp.ctx0, n), im.ShardedDiagonalSender(grp, n)
    gq = gr.encryptQuery(query, seed=2, nonce=9)
    assert np.array_equal(gq.export(), want_q)
    gidx = gs.indexScenario(gq)
    assert len(gidx) == 64 and np.array_equal(gidx.export(), want_idx)
    assert gr.decryptIndex(gidx) == planted
    gmem = gs.membershipScenario(gq)
    assert np.array_equal(gmem.export(), want_mem) and gr.decryptMembership(gmem) is True
    del gidx, gmem
    # the same database under the DEFAULT policy: 8 blocks per shard -> 128 babies + 4 giant steps per block (pre-rotated diagonals).
    # Other ciphertexts, the same answers
    grp.ctx0.set_matvec("auto")
    im.ShardedDiagonalEnroller(grp, n).serializeDB(db, seed=8)
    del db
    assert grp.shard_ctx(3).db_kind() == 6 and grp.shard_ctx(3).db_babies() == grp.ctx0.auto_babies(8) == 128
    gidx = gs.indexScenario(gq)
    assert len(gidx) == 64 and gr.decryptIndex(gidx) == planted
    assert gr.decryptMembership(gs.membershipScenario(gq)) is True
    del gq, gidx
    grp.close()


def test_more_shards_or_ranks_than_gpus_fail_with_a_clear_error(im):
    """Failure handling that needs no second GPU to test (round-3 review): a shard list naming a device this node does not have is a
    DEVICE error from hydia_group_create / hydia_ctx_create (nothing is created); DistDiagonalSender over RCCL with more ranks than
    GPUs says so at construction instead of failing deep inside the first collective."""
    import torch
    ndev = torch.cuda.device_count()
    with pytest.raises(im.HydiaError) as e:
        im.ShardGroup([0, ndev], im.default_params(log_n=11, vector_dim=64))
    assert e.value.code == -3 and "only %d HIP device" % ndev in str(e.value)
    with pytest.raises(im.HydiaError) as e:
        im.Context(im.default_params(log_n=11, vector_dim=64), ndev + 3)
    assert e.value.code == -3
    with pytest.raises(im.HydiaError) as e:
        im.Context(im.default_params(log_n=11, vector_dim=64), -1)
    assert e.value.code == -1

    class NcclLike:  # only what the constructor asks before it refuses
        @staticmethod
        def get_backend():
            return "nccl"

    cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0)
    try:
        with pytest.raises(RuntimeError) as e:
            im.DistDiagonalSender(cc, 5000, NcclLike, 0, ndev + 1, staging="device")
        assert "one GPU per rank" in str(e.value) and "%d GPU" % ndev in str(e.value)
    finally:
        cc.close()

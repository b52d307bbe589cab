"""The sharded DiagonalSender: one encrypted database cut by 16384-vector row-blocks over the GPUs of a node.

The reference walks the blocks in a plain serial loop (/root/reference/src/sender/sender_diag.cpp:28-30); blocks share
nothing but the rotated queries, so shard r owns the contiguous block range `shard_blocks(G, R, r)`, holds the keys itself
(same seed -> identical keys) and runs loop A + its own mat-vec + comparator.  Two drivers of the same steps:

  ShardGroup / ShardedDiagonalSender   R contexts inside ONE process (libhydia's hydia_group_*: one host thread per shard,
                                       peer copies between GPUs; shards may share a GPU) — what ./ImageMatching uses
  DistDiagonalEnroller / DistDiagonalSender   one PROCESS per GPU over torch.distributed — backend "nccl" (= RCCL over xGMI)
                                       with device buffers, or "gloo" with host staging (CPU tests, one-GPU rehearsals) —
                                       what bench.py --gpus N uses

indexScenario     = per-shard indexScenario, results gathered to rank/shard 0 in GLOBAL block order, so the receiver's
                    j + i*batchSize (src/receiver/receiver_hers.cpp:46-49) is the database index
membershipScenario= per-shard EvalAddMany (sender_diag.cpp:46) -> integer SUM of the partial ciphertexts across shards (at most
                    16 residues below 2^60 fit 64 bits) -> one `mod q` -> EvalSum (:47) on rank/shard 0
Both give the ciphertexts a single context holding the whole database would give, bit for bit (tests/test_gpu_sharding.py).
torch is used for the collectives only.
"""
import ctypes as C

import numpy as np

from . import hydia as _h


def shard_blocks(total_blocks, world, rank):
    """Contiguous block range [lo, hi) of `rank`: the first total_blocks % world ranks take one extra block
    (libhydia's hydia_shard_blocks — the rule the in-process group uses; host-only, needs no GPU)."""
    lo, hi = C.c_size_t(), C.c_size_t()
    _h.load_library().hydia_shard_blocks(total_blocks, world, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_vectors(n_total, slots, world, rank):
    """Vector range [lo, hi) of `rank` when the database of n_total vectors is cut on block (slots) boundaries."""
    G = -(-n_total // slots)
    blo, bhi = shard_blocks(G, world, rank)
    return min(blo * slots, n_total), min(bhi * slots, n_total)


def group_babies(cc, n_total, world):
    """ONE split of the mat-vec for a sharded database (include/hydia.h, hydia_set_matvec): the context's policy applied to the
    LARGEST shard — every rank computes the same answer from (n_total, world) and its own (identically configured) context."""
    G = -(-n_total // cc.slots)
    return cc.auto_babies(max(hi - lo for lo, hi in (shard_blocks(G, world, r) for r in range(world))))


# ------------------------------------------------------------------ one process, R contexts
class ShardGroup:
    """R contexts of one process, one per entry of `devices` (a GPU may appear several times).  `ctx0` is where queries are
    encrypted / imported and results decrypted."""

    def __init__(self, devices, params=None):
        self.L = _h.load_library()
        self.params = params or _h.default_params()
        devs = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        _h._chk(self.L.hydia_group_create(C.byref(self.params), devs, len(devices), C.byref(g)))
        self.g, self.world = g, len(devices)
        self.ctx0 = _h.Context._borrowed(self.L, self.params, self.L.hydia_group_ctx(self.g, 0))

    def shard_ctx(self, r):
        return _h.Context._borrowed(self.L, self.params, self.L.hydia_group_ctx(self.g, r))

    def keygen(self, seed=None):
        _h._chk(self.L.hydia_group_keygen(self.g, _h._p(_h._seed(seed))))

    def set_rotation_split(self, on):
        """loop A shared out over the active shards and exchanged (default) or recomputed by every shard"""
        _h._chk(self.L.hydia_group_set_rotation_split(self.g, 1 if on else 0))

    def shard_range(self, r):
        a, b = C.c_size_t(), C.c_size_t()
        _h._chk(self.L.hydia_group_shard_range(self.g, r, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.g:
            self.ctx0.h = None
            self.L.hydia_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedDiagonalEnroller:
    """DiagonalEnroller over a ShardGroup: each shard encrypts the rows of its own blocks (same nonces as unsharded)."""

    def __init__(self, group, num_vectors):
        self.group, self.numVectors = group, num_vectors

    def serializeDB(self, database, seed=None):
        assert database.dtype == np.float64 and database.flags.c_contiguous
        assert database.shape == (self.numVectors, self.group.ctx0.dim)
        _h._chk(self.group.L.hydia_group_db_enroll(self.group.g, _h._p(database), self.numVectors, _h._p(_h._seed(seed))))


class ShardedDiagonalSender:
    """include/sender_diag.h:5-28 over a ShardGroup; query and results live in group.ctx0."""

    def __init__(self, group, num_vectors):
        self.group, self.numVectors = group, num_vectors

    def _call(self, fn, q):
        h = C.c_void_p()
        _h._chk(fn(self.group.g, q.h, C.byref(h)))
        return _h.Ciphertext(self.group.ctx0, h)

    def computeSimilarity(self, query_cipher):
        return self._call(self.group.L.hydia_group_compute_similarity, query_cipher)

    def indexScenario(self, query_cipher):
        return self._call(self.group.L.hydia_group_index_scenario, query_cipher)

    def membershipScenario(self, query_cipher):
        return self._call(self.group.L.hydia_group_membership_scenario, query_cipher)


# ------------------------------------------------------------------ one process per GPU (torch.distributed)
class DistDiagonalEnroller:
    """Rank `rank` of `world` enrols its own rows of a database of n_total vectors: `rows` are the rank's vectors
    (shard_vectors(n_total, slots, world, rank)), encrypted with the nonces of the unsharded enrolment."""

    def __init__(self, cc, n_total, rank, world):
        self.cc, self.n_total, self.rank, self.world = cc, n_total, rank, world
        self.first, self.last = shard_vectors(n_total, cc.slots, world, rank)
        self.babies = group_babies(cc, n_total, world)

    def serializeDB(self, rows, seed=None):
        n_local = self.last - self.first
        assert rows.shape == (n_local, self.cc.dim)
        if n_local:
            _h.DiagonalEnroller(self.cc, n_local).serializeDB(rows, seed=seed, first_block=self.first // self.cc.slots,
                                                              matvec="hoisted" if self.babies >= self.cc.dim else self.babies)


class DistDiagonalSender:
    """DiagonalSender whose database is sharded over the ranks of a torch.distributed group.

    cc      the rank's own context (keys loaded; its shard enrolled by DistDiagonalEnroller)
    dist    torch.distributed (initialised); staging "device" = tensors in HBM + device pointers (nccl/RCCL),
            "host" = numpy/torch CPU tensors (gloo)
    Every rank calls the scenario methods; rank 0 passes the query ciphertext (other ranks pass None) and receives the result
    (other ranks get None).  make_sender(cc, n_local) builds the rank-local sender (default DiagonalSender)."""

    def __init__(self, cc, n_total, dist, rank, world, staging="device", make_sender=None, rotation_split=True):
        import torch
        self.torch = torch
        self.cc, self.n_total, self.dist, self.rank, self.world, self.staging = cc, n_total, dist, rank, world, staging
        if world > 16:  # the membership reduction adds residues below 2^60 as 64-bit integers: exact for at most 16 terms
            raise ValueError("DistDiagonalSender supports at most 16 ranks (integer sum of partial membership ciphertexts)")
        # RCCL wants one GPU per rank (it refuses two ranks on one device, and then only deep inside the first collective): say so here
        backend = getattr(dist, "get_backend", lambda: None)() if world > 1 else None
        if staging == "device" and backend == "nccl" and torch.cuda.device_count() < world:
            raise RuntimeError("DistDiagonalSender over RCCL needs one GPU per rank: world size %d but %d GPU(s) visible on this node "
                               "(use fewer ranks, or staging='host' over gloo to rehearse on one GPU)" % (world, torch.cuda.device_count()))
        self.G = -(-n_total // cc.slots)
        self.ranges = [shard_blocks(self.G, world, r) for r in range(world)]
        self.lo, self.hi = self.ranges[rank]
        self.max_blocks = max(hi - lo for lo, hi in self.ranges)
        first, last = shard_vectors(n_total, cc.slots, world, rank)
        self.local = (make_sender or _h.DiagonalSender)(cc, last - first) if self.hi > self.lo else None
        self._bufs = {}
        # timing (bench.py's instrumented steps, outside its timed region): a dict switches it on — every phase of a scenario call is
        # then fenced (library stream + torch's stream) on both sides and its host time added under its name, calls counted in "calls"
        self.timing = None
        # Loop A: option A = every rank computes all 511 rotations itself; option B (rotation_split) = rank k of the K
        # ranks that hold blocks computes the contiguous range shard_blocks(dim, K, k) and the ranges are all-gathered (3 GiB in all
        # over xGMI), so the node does loop A's work once instead of once per GPU.  Same ciphertexts either way.
        self.active = [r for r, (lo, hi) in enumerate(self.ranges) if hi > lo]
        K = len(self.active)
        # The form of the mat-vec is read off the RESIDENT databases (kind 5 hoisted / kind 6 pre-rotated with B babies), not re-derived
        # from the policy: a database that was loaded from a file, enrolled with an explicit split or filled per rank need not be what
        # the policy would pick today.  The ranks that hold blocks must agree (one split serves every shard of a database); the
        # baby-step / giant-step form needs B - 1 rotations per query — nothing worth sharing out, and the *Rotated entry points take
        # hoisted databases only — so option B is refused there.
        self._want_split = bool(rotation_split) and world > 1 and K > 1
        self.bsgs, self.babies, self._form = False, None, None
        self.rotation_split = self._want_split
        self._agree_form()
        self.rot_ranges = {r: shard_blocks(cc.dim, K, k) for k, r in enumerate(self.active)}
        self.rot_even = K > 0 and cc.dim % K == 0 and K == world  # every rank holds blocks and the ranges are equal: in-place all_gather

    def _local_form(self):
        cc = self.cc
        if self.local is None or not hasattr(cc, "db_kind"):
            return None
        return (int(cc.db_kind()), int(cc.db_babies()))

    def _agree_form(self):
        """(collective) every rank's resident (kind, babies); the block-holding ranks must hold ONE form.  Run at construction and
        again — on every rank, from _local — by the first scenario call after ANY rank's database changed (_forms_changed)."""
        mine = self._local_form()
        forms = [mine]
        if self.world > 1 and hasattr(self.cc, "db_kind"):
            forms = []
            for r in range(self.world):  # (broadcast_object_list is all the sender asks of `dist` elsewhere: stand-ins need no more)
                box = [mine if r == self.rank else None]
                self.dist.broadcast_object_list(box, src=r)
                forms.append(box[0])
        self._form = mine
        held = sorted({f for f in forms if f is not None and f[0] != 0})
        if len(held) > 1:
            raise ValueError("DistDiagonalSender: the ranks hold databases of different mat-vec forms (kind, babies) = %s; enrol every "
                             "shard with the group-wide split (sharding.group_babies)" % (held,))
        if held and held[0][0] not in (5, 6):
            raise ValueError("DistDiagonalSender needs a diagonal (approach 5) database, found kind %d" % held[0][0])
        if held:
            self.bsgs, self.babies = held[0][0] == 6, held[0][1]
        elif hasattr(self.cc, "auto_babies"):  # nothing enrolled yet: what the policy will pick (checked again at the first call)
            self.babies = group_babies(self.cc, self.n_total, self.world)
            self.bsgs = self.babies < self.cc.dim
        self.rotation_split = self.rotation_split and self._want_split and not self.bsgs

    def refresh_form(self):
        """(collective: every rank) re-read the mat-vec form of the resident databases after a re-enrolment / load / re-declaration.
        The scenario calls do this themselves when any rank's form changed; calling it explicitly moves the cost out of the first query."""
        self.rotation_split = self._want_split
        self._agree_form()

    # ---- per-phase timing of a scenario call (see __init__)
    COMM_PHASES = ("query_broadcast", "form_check", "rotations_all_gather", "result_gather")
    COMPUTE_PHASES = ("loop_a_share", "local_matvec_comparator")

    def _fence(self):
        if hasattr(self.cc, "sync"):
            self.cc.sync()
        if self.staging == "device":
            self.torch.cuda.synchronize()

    class _Phase:
        def __init__(self, snd, name):
            self.snd, self.name = snd, name

        def __enter__(self):
            if self.snd.timing is not None:
                import time
                self.snd._fence()
                self.t0 = time.perf_counter()

        def __exit__(self, *exc):
            if self.snd.timing is not None and exc[0] is None:
                import time
                self.snd._fence()
                self.snd.timing[self.name] = self.snd.timing.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3
            return False

    def _phase(self, name):
        return DistDiagonalSender._Phase(self, name)

    # ---- buffers: int64 tensors that mirror [count][poly][limb][N] residues
    def _buf(self, key, n):
        b = self._bufs.get(key)
        if b is None or b.numel() != n:
            b = self.torch.zeros(n, dtype=self.torch.int64, device="cuda" if self.staging == "device" else "cpu")
            if self.staging == "device":
                # the zero fill runs on torch's stream, which may still be waiting for an earlier collective; the library writes
                # into the buffer from its OWN stream right after this — without the wait the fill can land on top of the data
                self.torch.cuda.current_stream().synchronize()
            self._bufs[key] = b
        return b

    def _fill(self, buf, ct):
        """copy a ciphertext batch into the head of `buf`"""
        if self.staging == "device":
            ct.copy_to_device(buf.data_ptr())
        else:
            a = ct.export().reshape(-1).view(np.int64)
            buf[:a.size] = self.torch.from_numpy(a)

    def _to_ct(self, t, count, npoly, nl, scale):
        if self.staging == "device":
            t = t.contiguous()
            self.torch.cuda.current_stream().synchronize()  # the collective / cat that produced `t` ran on torch's stream
            return self.cc.ct_from_device(t.data_ptr(), count, npoly, nl, scale)
        return self.cc.import_ct(t.numpy().view(np.uint64).reshape(count, npoly, nl, self.cc.N), scale)

    def _bcast_query(self, q):
        n = 2 * self.cc.nQ * self.cc.N
        buf = self._buf("q", n)
        if self.rank == 0:
            c, p, l, s = q.shape()
            assert (c, p, l) == (1, 2, self.cc.nQ), "query must be one fresh ciphertext"
            self._fill(buf, q)
        if "q_scale" not in self._bufs:  # agreed once: encryptQuery always encodes at the context's Delta
            meta = [q.shape()[3] if self.rank == 0 else None]
            if self.world > 1:
                self.dist.broadcast_object_list(meta, src=0)
            self._bufs["q_scale"] = meta[0]
        elif self.rank == 0 and q.shape()[3] != self._bufs["q_scale"]:
            raise ValueError("query scale changed between calls")
        if self.world > 1:
            self.dist.broadcast(buf, src=0)
        return q if self.rank == 0 else self._to_ct(buf, 1, 2, self.cc.nQ, self._bufs["q_scale"])

    def _meta(self, key, ct):
        """(npoly, nl, scale) of a result batch, agreed once per result kind (rank 0 always owns blocks)"""
        if key not in self._bufs:
            m = [ct.shape()[1:] if self.rank == 0 else None]
            if self.world > 1:
                self.dist.broadcast_object_list(m, src=0)
            self._bufs[key] = m[0]
        return self._bufs[key]

    def _gather_blocks(self, key, res):
        """per-rank [blocks][npoly][nl][N] -> on rank 0 one batch of G ciphertexts in global block order"""
        npoly, nl, scale = self._meta(key + "_meta", res)
        per = npoly * nl * self.cc.N
        send = self._buf(key + "_send", self.max_blocks * per)  # uneven shards: padded to the largest
        if res is not None:
            self._fill(send, res)
        if self.world == 1:
            return res
        recv = [self._buf(key + "_recv%d" % r, send.numel()) for r in range(self.world)] if self.rank == 0 else None
        self.dist.gather(send, recv, dst=0)
        if self.rank != 0:
            return None
        parts = [recv[r][:(hi - lo) * per] for r, (lo, hi) in enumerate(self.ranges) if hi > lo]
        return self._to_ct(self.torch.cat(parts), self.G, npoly, nl, scale)

    def _gathered_rotations(self, q):
        """option B: this rank's range of loop A into its slice of the full [dim][2][nQ][N] buffer, all_gather of the slices, and a
        ciphertext handle over the buffer (no copy).  Ranks without blocks take part in the collective with an empty range."""
        cc, dim = self.cc, self.cc.dim
        per = 2 * cc.nQ * cc.N
        full = self._buf("rot", dim * per)
        lo, hi = self.rot_ranges.get(self.rank, (0, 0))
        scale = self._bufs["q_scale"]
        if self.staging == "device":
            with self._phase("loop_a_share"):
                if hi > lo:
                    self.local.rotateQueryRangeInto(q, lo, hi - lo, full.data_ptr() + lo * per * 8)
                cc.sync()  # the slice is written on the library's stream; the collective runs on torch's
            with self._phase("rotations_all_gather"):
                self._all_gather_rotations_device(full, lo, hi, per)
            return cc.ct_view_device(full.data_ptr(), dim, 2, cc.nQ, scale, keepalive=full) if self.local is not None else None
        # host staging (gloo): the same steps through host tensors
        cnt = max([h - l for l, h in self.rot_ranges.values()] + [1])
        send = self._buf("rot_send", cnt * per)
        with self._phase("loop_a_share"):
            if hi > lo:
                a = self.local.rotateQueryRange(q, lo, hi - lo).export().reshape(-1).view(np.int64)
                send[:a.size] = self.torch.from_numpy(a)
        with self._phase("rotations_all_gather"):
            recv = [self._buf("rot_recv%d" % r, cnt * per) for r in range(self.world)]
            self.dist.all_gather(recv, send)
            if self.local is not None:
                for r, (l, h) in self.rot_ranges.items():
                    if h > l:
                        full[l * per:h * per] = recv[r][:(h - l) * per]
        if self.local is None:
            return None
        return cc.import_ct(full.numpy().view(np.uint64).reshape(dim, 2, cc.nQ, cc.N), scale)

    def _all_gather_rotations_device(self, full, lo, hi, per):
        if self.rot_even:
            self.dist.all_gather_into_tensor(full, full[lo * per:hi * per])
        else:
            cnt = max(h - l for l, h in self.rot_ranges.values())
            send = self._buf("rot_send", cnt * per)
            if hi > lo:
                send[:(hi - lo) * per].copy_(full[lo * per:hi * per])
            recv = [self._buf("rot_recv%d" % r, cnt * per) for r in range(self.world)]
            self.dist.all_gather(recv, send)
            for r, (l, h) in self.rot_ranges.items():
                if r != self.rank and h > l:
                    full[l * per:h * per].copy_(recv[r][:(h - l) * per])
        self.torch.cuda.current_stream().synchronize()  # the library reads `full` from its own stream

    def _forms_changed(self):
        """(collective: every rank, every scenario call) has ANY rank's resident database changed form since the ranks last agreed?
        One word, max-reduced: a rank that holds no block cannot see that the others re-enrolled, and a rank that raised on its own
        while the others entered the next collective would leave them hanging until the backend's timeout."""
        mine = 1 if self._local_form() != self._form else 0
        if self.world == 1:
            return bool(mine)
        t = self._buf("chg", 1)
        t.fill_(mine)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def _local(self, fn_name, q):
        with self._phase("form_check"):
            changed = self._forms_changed()
        if changed:
            # enrolled after the sender was built, re-enrolled, loaded or re-declared: every rank learns of it in the same call and the
            # ranks agree again (a collective, entered by all of them here).  Disagreeing forms raise ValueError on EVERY rank.
            self.refresh_form()
        if self.rotation_split and self.bsgs:
            raise ValueError("rotation_split (loop A shared out over the ranks) needs hoisted databases; the resident ones are pre-rotated "
                             "for %s babies" % self.babies)
        if self.rotation_split and not self.bsgs and len(self.active) > 1:
            rot = self._gathered_rotations(q)
            with self._phase("local_matvec_comparator"):
                return getattr(self.local, fn_name + "Rotated")(rot) if self.local is not None else None
        with self._phase("local_matvec_comparator"):  # (loop A recomputed by this rank is part of it)
            return getattr(self.local, fn_name)(q) if self.local is not None else None

    def _scenario(self, key, fn_name, query_cipher):
        if self.timing is not None:
            self.timing["calls"] = self.timing.get("calls", 0) + 1
        with self._phase("query_broadcast"):
            q = self._bcast_query(query_cipher)
        res = self._local(fn_name, q)
        with self._phase("result_gather"):
            return self._gather_blocks(key, res)

    def computeSimilarity(self, query_cipher):
        return self._scenario("sim", "computeSimilarity", query_cipher)

    def indexScenario(self, query_cipher):
        return self._scenario("idx", "indexScenario", query_cipher)

    def membershipScenario(self, query_cipher):
        q = self._bcast_query(query_cipher)
        idx = self._local("indexScenario", q)
        npoly, nl, scale = self._meta("idx_meta", idx)
        part = self._buf("mem", npoly * nl * self.cc.N)
        if idx is not None:
            self._fill(part, self.cc.add_many(idx))   # EvalAddManyInPlace over this rank's blocks
        else:
            part.zero_()
        if self.world > 1:
            self.dist.reduce(part, dst=0, op=self.dist.ReduceOp.SUM)   # plain int64 sums: < 16 * 2^60
        if self.rank != 0:
            return None
        tot = self._to_ct(part, 1, npoly, nl, scale)
        self.cc.ct_mod_reduce(tot)
        return self.cc.eval_sum(tot)

// image_matching_amd/csrc/group.cpp — the sharded DiagonalSender: one encrypted database cut by 16384-vector row-blocks over
// R contexts (one per GPU of a node; several shards may also share a GPU), driven from one process.
//
// The reference walks the blocks in a plain serial loop (/root/reference/src/sender/sender_diag.cpp:28-30) and nothing but
// the shared rotated queries connects them, so shard r owns the contiguous block range hydia_shard_blocks gives it, computes
// loop A locally (keys are replicated per GPU — SURVEY 8e option A: nothing is exchanged on the critical path) and runs an
// independent mat-vec + comparator on its own host thread and streams.  What crosses GPUs:
//   query            one 6 MiB ciphertext, shard 0 -> every shard                                   (peer copy)
//   indexScenario    0.5 MiB per block back to shard 0, placed in GLOBAL block order, so the receiver's
//                    j + i*batchSize (src/receiver/receiver_hers.cpp:46-49) is already the database index
//   membership       (sender_diag.cpp:46-47) per shard EvalAddMany over its own blocks; the R partial sums are added as plain
//                    64-bit integers (a group has at most 16 shards: 16 residues below 2^60 fit 64 bits) and reduced mod q once, then EvalSum on shard 0 —
//                    bit-identical to the unsharded EvalAddMany + EvalSum
// The multi-process form of the same steps (one rank per GPU, RCCL gather / all-reduce) is image_matching_amd/sharding.py; both
// are built from the same entry points (hydia_db_enroll_shard, hydia_index_scenario, hydia_add_many, hydia_ct_add_raw,
// hydia_ct_mod_reduce, hydia_eval_sum).
#include <algorithm>
#include <cstdio>
#include <exception>
#include <thread>

#include "capi_internal.h"

using namespace hydia;

struct hydia_group {
    std::vector<hydia_ctx *> shard;
    std::vector<size_t> blk_lo, blk_hi;  // block range of each shard of the resident database
    size_t n_vectors = 0, n_blocks = 0;
    bool rot_split = true;  // loop A's rotations shared out over the active shards and exchanged (hydia_group_set_rotation_split)
    bool bsgs = false;      // the resident database is pre-rotated for fewer than vector_dim babies: loop A is short, nothing is shared out
};

namespace {

void shard_blocks(size_t total, uint32_t world, uint32_t rank, size_t *lo, size_t *hi) {
    const size_t base = total / world, extra = total % world;
    *lo = (size_t)rank * base + std::min<size_t>(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

// f(r) for every shard in `which` on its own host thread (HIP's current device is per thread); the first exception is
// re-thrown in the caller after all threads have joined
template <class F>
void on_shards(hydia_group *g, const std::vector<uint32_t> &which, F f) {
    std::vector<std::exception_ptr> err(which.size());
    std::vector<std::thread> th;
    for (size_t k = 0; k < which.size(); k++)
        th.emplace_back([&, k] {
            try {
                use_device(g->shard[which[k]]);
                f(which[k]);
            } catch (...) {
                err[k] = std::current_exception();
            }
        });
    for (auto &t : th) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}
std::vector<uint32_t> active(const hydia_group *g) {
    std::vector<uint32_t> a;
    for (uint32_t r = 0; r < g->shard.size(); r++)
        if (g->blk_hi[r] > g->blk_lo[r]) a.push_back(r);
    return a;
}
// Enqueued on the DESTINATION context's stream, so everything that context launches afterwards is ordered behind the copy (a
// plain device-to-device hipMemcpy is asynchronous to the host and runs on the null stream, which the contexts' non-blocking
// streams do not wait for).  The source must be complete (its context synchronised) and stay alive until dst.sync().
void copy_between(Context &dst, u64 *d, Context &src, const u64 *s, size_t bytes) {
    (void)hipSetDevice(dst.device);
    if (dst.device == src.device) HIP_CHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, dst.stream));
    else HIP_CHECK(hipMemcpyPeerAsync(d, dst.device, s, src.device, bytes, dst.stream));
}
// the query enters on shard 0; every other active shard gets its own copy
std::vector<Ct> broadcast_query(hydia_group *g, const std::vector<uint32_t> &act, const Ct &q) {
    Context &c0 = g->shard[0]->cx;
    c0.sync_all();
    std::vector<Ct> qs(g->shard.size());
    for (uint32_t r : act) {
        Context &cr = g->shard[r]->cx;
        if (r == 0) {
            qs[r] = q.alias(q.nl);
            continue;
        }
        (void)hipSetDevice(cr.device);
        cr.sync_all();
        qs[r] = Ct(&cr, q.X, q.npoly, q.nl, q.scale);
        copy_between(cr, qs[r].d, c0, q.d, q.bytes());
        cr.sync();  // the caller may free or overwrite the query as soon as the scenario call returns
    }
    return qs;
}
void check_query(hydia_group *g, const hydia_ct *q) {
    if (!g || !q) throw std::runtime_error("hydia: null argument");
    if (q->owner != g->shard[0]) throw std::runtime_error("hydia: the query must live in the group's shard 0 context (hydia_group_ctx(g, 0))");
    if (g->n_blocks == 0) throw StateError("hydia: no database resident in the group");
    if (!q->c.compact()) throw std::runtime_error("hydia: query handle must be compact");
}
// per-shard results [blocks of r][2][nl][N] -> one batch on shard 0 in global block order
Ct gather_blocks(hydia_group *g, const std::vector<uint32_t> &act, std::vector<Ct> &res) {
    Context &c0 = g->shard[0]->cx;
    const Ct &f = res[act[0]];
    (void)hipSetDevice(c0.device);
    Ct out(&c0, (int)g->n_blocks, f.npoly, f.nl, f.scale);
    for (uint32_t r : act) {
        Ct &p = res[r];
        if (p.npoly != f.npoly || p.nl != f.nl || !p.compact() || p.X != (int)(g->blk_hi[r] - g->blk_lo[r]))
            throw std::runtime_error("hydia: shard result shape mismatch");
        copy_between(c0, out.d + g->blk_lo[r] * out.ct_elems(), g->shard[r]->cx, p.d, p.bytes());
    }
    (void)hipSetDevice(c0.device);
    c0.sync();  // the shard-local sources are released next
    return out;
}
Ct compact(Context &cx, Ct &&c) { return (c.view || !c.compact()) ? cx.clone(c) : std::move(c); }

// Loop A split over the active shards (SURVEY 8e option B): shard act[k] computes the contiguous rotation range
// hydia_shard_blocks(dim, K, k) into its own slice of its full [dim][2][nQ][N] buffer, then fetches every other shard's slice by
// peer copies enqueued on its OWN stream, so its mat-vec is ordered behind them.  Every shard ends up with exactly the rotations
// rotate_query gives (each one is computed by the same kernels on the same operands wherever it runs).
std::vector<Ct> split_rotations(hydia_group *g, const std::vector<uint32_t> &act, const std::vector<Ct> &qs) {
    const uint32_t K = (uint32_t)act.size();
    const int dim = g->shard[0]->cx.prm.dim;
    std::vector<Ct> rot(g->shard.size());
    std::vector<size_t> lo(K), hi(K);
    for (uint32_t k = 0; k < K; k++) shard_blocks((size_t)dim, K, k, &lo[k], &hi[k]);
    std::vector<uint32_t> idx(g->shard.size(), 0);
    for (uint32_t k = 0; k < K; k++) idx[act[k]] = k;
    on_shards(g, act, [&](uint32_t r) {
        Context &cx = g->shard[r]->cx;
        const uint32_t k = idx[r];
        rot[r] = Ct(&cx, dim, 2, qs[r].nl, qs[r].scale);
        cx.rotate_query_range(qs[r], (int)lo[k], (int)(hi[k] - lo[k]), rot[r].d + lo[k] * rot[r].ct_elems());
        cx.sync_all();  // the slice is complete before any peer reads it
    });
    on_shards(g, act, [&](uint32_t r) {
        Context &cx = g->shard[r]->cx;
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t s = act[k];
            if (s == r || hi[k] == lo[k]) continue;
            const size_t off = lo[k] * rot[r].ct_elems();
            copy_between(cx, rot[r].d + off, g->shard[s]->cx, rot[s].d + off, (hi[k] - lo[k]) * rot[r].ct_elems() * sizeof(u64));
        }
        cx.sync();  // the sources may be released once every shard has its copies
    });
    return rot;
}

template <class F, class FR>
int sharded_blocks_call(hydia_group *g, const hydia_ct *query, hydia_ct **out, F per_shard, FR per_shard_rot) {
    API_BEGIN
    REQUIRE(out, "null argument");
    check_query(g, query);
    const std::vector<uint32_t> act = active(g);
    std::vector<Ct> qs = broadcast_query(g, act, query->c);
    std::vector<Ct> res(g->shard.size());
    if (g->rot_split && !g->bsgs && act.size() > 1) {
        std::vector<Ct> rot = split_rotations(g, act, qs);
        on_shards(g, act, [&](uint32_t r) {
            Context &cx = g->shard[r]->cx;
            res[r] = compact(cx, per_shard_rot(cx, rot[r]));
            cx.sync_all();
            rot[r] = Ct();  // back to this shard's pool, under its own device
        });
    } else {
        on_shards(g, act, [&](uint32_t r) {
            Context &cx = g->shard[r]->cx;
            res[r] = compact(cx, per_shard(cx, qs[r]));
            cx.sync_all();
        });
    }
    use_device(g->shard[0]);
    Ct all = gather_blocks(g, act, res);
    for (uint32_t r : act) {  // shard-local buffers go back to their own pools under their own device
        use_device(g->shard[r]);
        res[r] = Ct();
        qs[r] = Ct();
    }
    use_device(g->shard[0]);
    *out = wrap(g->shard[0], std::move(all));
    return HYDIA_OK;
    API_END
}

}  // namespace

extern "C" {

void hydia_shard_blocks(size_t total_blocks, uint32_t world, uint32_t rank, size_t *lo, size_t *hi) {
    size_t a = 0, b = 0;
    if (world >= 1 && rank < world) shard_blocks(total_blocks, world, rank, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
}

int hydia_group_create(const hydia_params *p, const int *devices, uint32_t n_shards, hydia_group **out) {
    hydia_group *g = nullptr;
    try {
        if (!p || !devices || !out) return hydia_fail(HYDIA_ERR_ARG, "null argument");
        if (n_shards < 1 || n_shards > 16) return hydia_fail(HYDIA_ERR_ARG, "a group has 1 to 16 shards");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return hydia_fail(HYDIA_ERR_DEVICE, "hydia: no HIP device visible — libhydia has no CPU fallback");
        for (uint32_t r = 0; r < n_shards; r++) {
            if (devices[r] < 0) return hydia_fail(HYDIA_ERR_ARG, "bad device index");
            if (devices[r] >= ndev)  // a shard list written for a bigger node: nothing is created, nothing leaks
                return hydia_fail(HYDIA_ERR_DEVICE, ("hydia: shard " + std::to_string(r) + " asks for device " + std::to_string(devices[r]) + " but only " +
                                                     std::to_string(ndev) + " HIP device(s) visible"));
        }
        g = new hydia_group;
        for (uint32_t r = 0; r < n_shards; r++) g->shard.push_back(new hydia_ctx(hydia_to_params(p), devices[r]));
        g->blk_lo.assign(n_shards, 0);
        g->blk_hi.assign(n_shards, 0);
        // shards on different GPUs exchange the query and the results by peer copies
        for (uint32_t a = 0; a < n_shards; a++)
            for (uint32_t b = 0; b < n_shards; b++) {
                const int da = devices[a], db = devices[b];
                int can = 0;
                if (da == db) continue;
                hipError_t e = hipDeviceCanAccessPeer(&can, da, db);
                if (e == hipSuccess && can) {
                    (void)hipSetDevice(da);
                    e = hipDeviceEnablePeerAccess(db, 0);
                    if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
                }
                if (e != hipSuccess || !can) {  // not fatal: hipMemcpyPeerAsync then stages through the host — but say so, once per pair
                    (void)hipGetLastError();
                    fprintf(stderr, "hydia: no direct peer access from GPU %d to GPU %d (%s): shard exchanges will be staged through host memory\n",
                            da, db, e != hipSuccess ? hipGetErrorString(e) : "not supported");
                }
            }
        *out = g;
        return HYDIA_OK;
    } catch (const DeviceError &e) {
        hydia_group_destroy(g);
        return hydia_fail(HYDIA_ERR_DEVICE, e.what());
    } catch (const std::exception &e) {
        hydia_group_destroy(g);
        return hydia_fail(HYDIA_ERR_ARG, e.what());
    }
}
void hydia_group_destroy(hydia_group *g) {
    if (!g) return;
    for (size_t r = g->shard.size(); r-- > 0;) hydia_ctx_destroy(g->shard[r]);  // borrowers of shared keys go before their owner
    delete g;
}
uint32_t hydia_group_size(const hydia_group *g) { return g ? (uint32_t)g->shard.size() : 0; }
hydia_ctx *hydia_group_ctx(hydia_group *g, uint32_t shard) { return g && shard < g->shard.size() ? g->shard[shard] : nullptr; }
int hydia_group_shard_range(const hydia_group *g, uint32_t shard, size_t *first_vector, size_t *n_vectors) {
    REQUIRE(g && shard < g->shard.size(), "bad argument");
    const size_t S = (size_t)g->shard[0]->cx.slots;
    const size_t lo = std::min(g->blk_lo[shard] * S, g->n_vectors), hi = std::min(g->blk_hi[shard] * S, g->n_vectors);
    if (first_vector) *first_vector = lo;
    if (n_vectors) *n_vectors = hi - lo;
    return HYDIA_OK;
}

// the same seed on every GPU gives identical keys (no key distribution); shards that share a GPU share one resident copy
int hydia_group_keygen(hydia_group *g, const uint8_t seed[32]) {
    API_BEGIN
    REQUIRE(g && seed, "null argument");
    std::vector<uint32_t> owners, borrowers;
    std::vector<int> owner_of(g->shard.size(), -1);
    for (uint32_t r = 0; r < g->shard.size(); r++) {
        for (uint32_t o : owners)
            if (g->shard[o]->cx.device == g->shard[r]->cx.device) owner_of[r] = (int)o;
        if (owner_of[r] < 0) owners.push_back(r);
        else borrowers.push_back(r);
    }
    for (uint32_t r : borrowers)
        if (!g->shard[r]->cx.keys_borrowed && (g->shard[r]->cx.relin_key.d || g->shard[r]->cx.d_sk))
            return hydia_fail(HYDIA_ERR_STATE, "hydia: a shard already holds keys of its own");
    on_shards(g, owners, [&](uint32_t r) { client_keygen(g->shard[r]->cx, seed); });
    for (uint32_t r : borrowers) {
        use_device(g->shard[r]);
        Context &cx = g->shard[r]->cx, &src = g->shard[(size_t)owner_of[r]]->cx;
        if (cx.keys_borrowed) {  // re-key: the owner's buffers were rewritten in place, only the loop-A tables need a refresh
            cx.sync_all();
            src.build_rotptrs();
            src.sync_all();
            cx.rotptrs_packed = src.rotptrs_packed;
            cx.rotptrs_premul = src.rotptrs_premul;
            cx.d_rotpack = src.d_rotpack;
            HIP_CHECK(hipMemcpyAsync((void *)cx.d_rotptrs, (const void *)src.d_rotptrs, sizeof(u64 *) * (size_t)cx.prm.dim,
                                     hipMemcpyDeviceToDevice, cx.stream));
            cx.sync();
        } else {
            cx.adopt_keys(src);
        }
    }
    return HYDIA_OK;
    API_END
}

// DiagonalEnroller::serializeDB over the group: normalises `db` in place (enroller_diag.cpp:32-35); shard r encrypts the rows
// of its own blocks with the nonces of the unsharded enrolment, so the shards together hold exactly the ciphertexts a
// single context would
int hydia_group_db_enroll(hydia_group *g, double *db, size_t n, const uint8_t seed[32]) {
    API_BEGIN
    REQUIRE(g && db && seed && n >= 1, "bad argument");
    const uint32_t R = (uint32_t)g->shard.size();
    const size_t S = (size_t)g->shard[0]->cx.slots, dim = (size_t)g->shard[0]->cx.prm.dim;
    const size_t G = (n + S - 1) / S;
    for (uint32_t r = 0; r < R; r++) shard_blocks(G, R, r, &g->blk_lo[r], &g->blk_hi[r]);
    g->n_vectors = n;
    g->n_blocks = G;
    std::vector<uint32_t> all(R);
    for (uint32_t r = 0; r < R; r++) all[r] = r;
    // ONE form of the mat-vec for the whole database (shard 0's policy; auto looks at the LARGEST shard), so the shards' results
    // are the ciphertexts a single context of that form computes
    size_t max_blocks = 0;
    for (uint32_t r = 0; r < R; r++) max_blocks = std::max(max_blocks, g->blk_hi[r] - g->blk_lo[r]);
    const int babies = g->shard[0]->cx.babies_for(max_blocks);
    g->bsgs = babies < (int)dim;
    on_shards(g, all, [&](uint32_t r) {
        Context &cx = g->shard[r]->cx;
        const size_t first = g->blk_lo[r] * S, last = std::min(g->blk_hi[r] * S, n);
        if (last <= first) {  // more shards than blocks: this one holds nothing
            cx.db_resize(0, 0, -1);
            cx.db_kind = 0;
            return;
        }
        const size_t nl = last - first;
        const size_t per = S / dim, nblk = (nl + dim - 1) / dim;
        cx.db_resize(nl, ((nblk + per - 1) / per) * dim, babies);
        client_enroll(cx, db + first * dim, nl, seed, g->blk_lo[r], babies);
        cx.db_kind = babies < (int)dim ? 6 : 5;
        cx.db_babies = babies;
    });
    return HYDIA_OK;
    API_END
}

int hydia_group_set_rotation_split(hydia_group *g, int on) {
    REQUIRE(g, "null argument");
    g->rot_split = on != 0;
    return HYDIA_OK;
}
int hydia_group_compute_similarity(hydia_group *g, const hydia_ct *query, hydia_ct **out) {
    return sharded_blocks_call(g, query, out, [](Context &cx, const Ct &q) { return cx.similarity(q); },
                               [](Context &cx, const Ct &rot) { return cx.similarity_rot(rot); });
}
int hydia_group_index_scenario(hydia_group *g, const hydia_ct *query, hydia_ct **out) {
    return sharded_blocks_call(g, query, out, [](Context &cx, const Ct &q) { return cx.index_scenario(q); },
                               [](Context &cx, const Ct &rot) { return cx.index_scenario_rot(rot); });
}
int hydia_group_membership_scenario(hydia_group *g, const hydia_ct *query, hydia_ct **out) {
    API_BEGIN
    REQUIRE(out, "null argument");
    check_query(g, query);
    const std::vector<uint32_t> act = active(g);
    std::vector<Ct> qs = broadcast_query(g, act, query->c);
    std::vector<Ct> part(g->shard.size());
    std::vector<Ct> rot;
    const bool split = g->rot_split && !g->bsgs && act.size() > 1;
    if (split) rot = split_rotations(g, act, qs);
    on_shards(g, act, [&](uint32_t r) {
        Context &cx = g->shard[r]->cx;
        Ct idx = split ? cx.index_scenario_rot(rot[r]) : cx.index_scenario(qs[r]);
        part[r] = cx.add_many(idx);  // EvalAddManyInPlace over this shard's blocks
        cx.sync_all();
        if (split) rot[r] = Ct();
    });
    Context &c0 = g->shard[0]->cx;
    use_device(g->shard[0]);
    Ct acc;
    {
        const Ct &f = part[act[0]];
        acc = Ct(&c0, 1, f.npoly, f.nl, f.scale);
        u64 *tmp = c0.pool.get(acc.bytes());
        bool first = true;
        for (uint32_t r : act) {  // copy and integer add are both ordered on shard 0's stream: tmp is reused safely
            copy_between(c0, first ? acc.d : tmp, g->shard[r]->cx, part[r].d, acc.bytes());
            if (!first) c0.add_raw_inplace(acc, tmp);
            first = false;
        }
        c0.sync();
        c0.pool.put(tmp);
    }
    c0.mod_reduce_inplace(acc);
    Ct res = c0.eval_sum(acc);
    for (uint32_t r : act) {
        use_device(g->shard[r]);
        part[r] = Ct();
        qs[r] = Ct();
    }
    use_device(g->shard[0]);
    *out = wrap(g->shard[0], std::move(res));
    return HYDIA_OK;
    API_END
}

}  // extern "C"

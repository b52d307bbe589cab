// image_matching_amd/csrc/capi.cpp — the extern "C" boundary of libhydia.so (include/hydia.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <sys/random.h>

#include "capi_internal.h"

using namespace hydia;

static thread_local std::string g_err;
int hydia_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define fail hydia_fail
// the nonce is a 40-bit field of the sampler stream id (client_kernels.h HY_STREAM): a larger value would bleed into the domain byte
#define HYDIA_NONCE_LIMIT (1ull << 40)

Params hydia_to_params(const hydia_params *p) {
    Params r;
    r.logN = (int)p->log_n;
    r.mult_depth = (int)p->mult_depth;
    r.scale_bits = (int)p->scale_bits;
    r.first_bits = (int)p->first_mod_bits;
    r.dnum = (int)p->dnum;
    r.dim = (int)p->vector_dim;
    return r;
}
static void fill_info(const HostParams &h, hydia_info *o) {
    o->log_n = h.prm.logN; o->n = h.N; o->slots = h.slots; o->n_q = h.nQ; o->n_p = h.nP; o->dnum = h.prm.dnum;
    o->alpha = h.alpha; o->vector_dim = h.prm.dim; o->delta = h.delta;
}
extern "C" {

const char *hydia_last_error(void) { return g_err.c_str(); }
const char *hydia_version(void) { return "hydia-mi355x 0.1 (gfx950)"; }

/* 32 bytes from the operating system's entropy pool (getrandom(2), /dev/urandom as fall-back): what every role object
 * seeds its encryption randomness from unless the caller supplies a seed of its own */
int hydia_random_seed(uint8_t out[32]) {
    if (!out) return fail(HYDIA_ERR_ARG, "null argument");
    size_t got = 0;
    while (got < 32) {
        ssize_t r = getrandom(out + got, 32 - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got < 32) {
        FILE *f = fopen("/dev/urandom", "rb");
        if (f) {
            got += fread(out + got, 1, 32 - got, f);
            fclose(f);
        }
    }
    return got == 32 ? HYDIA_OK : fail(HYDIA_ERR_INTERNAL, "hydia: no entropy source (getrandom and /dev/urandom failed)");
}
void hydia_default_params(hydia_params *o) {
    o->log_n = 15;
    o->mult_depth = (uint32_t)hydia_compute_required_depth(5);
    o->scale_bits = 45;
    o->first_mod_bits = 60;
    o->dnum = 3;
    o->vector_dim = 512;
}
/* src/openFHE_wrapper.cpp:6-44 with COMP_DEPTH 10, ALPHA_DEPTH 2 (include/config.h:14,18) */
size_t hydia_compute_required_depth(size_t approach) {
    const size_t COMP_DEPTH = 10, ALPHA_DEPTH = 2;
    switch (approach) {
        case 1: return 1 + 2 + COMP_DEPTH;
        case 2: return 1 + 2 + ALPHA_DEPTH + 3 + COMP_DEPTH;
        case 3: return 1 + 1 + COMP_DEPTH;
        case 4: return 1 + COMP_DEPTH;
        case 5: return 1 + COMP_DEPTH;
        default: return 0;
    }
}
int hydia_params_describe(const hydia_params *p, hydia_info *info, uint64_t *moduli, uint64_t *roots) {
    API_BEGIN
    REQUIRE(p, "null params");
    HostParams h(hydia_to_params(p));
    if (info) fill_info(h, info);
    for (int m = 0; m < h.nT; m++) {
        if (moduli) moduli[m] = h.q[m];
        if (roots) roots[m] = h.psi[m];
    }
    return HYDIA_OK;
    API_END
}
int hydia_ctx_create(const hydia_params *p, int device, hydia_ctx **out) {
    API_BEGIN
    REQUIRE(p && out, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(HYDIA_ERR_DEVICE, "hydia: no HIP device visible — libhydia has no CPU fallback");
    REQUIRE(device >= 0, "bad device index");
    if (device >= ndev)  // a device this machine does not have (a shard list written for a bigger node): a device error, not a typo
        return fail(HYDIA_ERR_DEVICE, "hydia: device index " + std::to_string(device) + " but only " + std::to_string(ndev) + " HIP device(s) visible");
    *out = new hydia_ctx(hydia_to_params(p), device);
    return HYDIA_OK;
    API_END
}
int hydia_ctx_create_custom(const hydia_params *p, const uint64_t *moduli, const uint64_t *roots, uint32_t n_q, uint32_t n_p,
                            int device, hydia_ctx **out) {
    API_BEGIN
    REQUIRE(p && out && moduli, "null argument");
    REQUIRE(n_q >= 2 && n_p >= 1 && n_q + n_p <= HY_MAX_MODS, "bad limb counts");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(HYDIA_ERR_DEVICE, "hydia: no HIP device visible — libhydia has no CPU fallback");
    REQUIRE(device >= 0, "bad device index");
    if (device >= ndev)  // a device this machine does not have (a shard list written for a bigger node): a device error, not a typo
        return fail(HYDIA_ERR_DEVICE, "hydia: device index " + std::to_string(device) + " but only " + std::to_string(ndev) + " HIP device(s) visible");
    Params prm = hydia_to_params(p);
    prm.mult_depth = (int)n_q - 1;
    prm.custom_q.assign(moduli, moduli + n_q + n_p);
    if (roots) prm.custom_psi.assign(roots, roots + n_q + n_p);
    prm.custom_nP = (int)n_p;
    *out = new hydia_ctx(prm, device);
    return HYDIA_OK;
    API_END
}
void hydia_ctx_destroy(hydia_ctx *ctx) {
    if (!ctx) return;
    use_device(ctx);
    if (ctx->refs.fetch_sub(1) == 1) delete ctx;  // otherwise the last hydia_ct_free does it
}
int hydia_get_info(const hydia_ctx *ctx, hydia_info *out) {
    REQUIRE(ctx && out, "null argument");
    fill_info(ctx->cx, out);
    return HYDIA_OK;
}
int hydia_get_moduli(const hydia_ctx *ctx, uint64_t *moduli, uint64_t *roots) {
    REQUIRE(ctx, "null ctx");
    for (int m = 0; m < ctx->cx.nT; m++) {
        if (moduli) moduli[m] = ctx->cx.q[m];
        if (roots) roots[m] = ctx->cx.psi[m];
    }
    return HYDIA_OK;
}
int hydia_sync(hydia_ctx *ctx) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx, "null ctx");
    ctx->cx.sync();
    return HYDIA_OK;
    API_END
}
int hydia_memory_stats(hydia_ctx *ctx, uint64_t *live, uint64_t *cached, uint64_t *peak) {
    REQUIRE(ctx, "null ctx");
    if (live) *live = ctx->cx.pool.bytes_live;
    if (cached) *cached = ctx->cx.pool.bytes_cached;
    if (peak) *peak = ctx->cx.pool.peak;
    return HYDIA_OK;
}

// ------------------------------------------------------------------ keys
int hydia_keygen(hydia_ctx *ctx, const uint8_t seed[32]) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && seed, "null argument");
    client_keygen(ctx->cx, seed);
    return HYDIA_OK;
    API_END
}
int hydia_import_eval_key(hydia_ctx *ctx, int rot, const uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data && rot >= 0 && rot < ctx->cx.slots, "bad argument");
    ctx->cx.load_eval_key(rot, (const u64 *)data);
    return HYDIA_OK;
    API_END
}
int hydia_export_eval_key(hydia_ctx *ctx, int rot, uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    Context &cx = ctx->cx;
    const u64 *src = nullptr;
    if (rot == 0) src = cx.relin_key.d;
    else if (cx.rot_keys.count(rot)) src = cx.rot_keys[rot].d;
    if (!src) return fail(HYDIA_ERR_STATE, "hydia: evaluation key not loaded");
    cx.sync();
    HIP_CHECK(hipMemcpy(data, src, (size_t)cx.prm.dnum * 2 * cx.nT * cx.N * sizeof(u64), hipMemcpyDeviceToHost));
    return HYDIA_OK;
    API_END
}
int hydia_fill_eval_keys_random(hydia_ctx *ctx, uint64_t seed) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx, "null ctx");
    Context &cx = ctx->cx;
    std::vector<int> rots;
    rots.push_back(0);
    for (int i = 1; i < cx.prm.dim; i++) rots.push_back(i);
    for (int i = cx.prm.dim; i < cx.slots; i <<= 1) rots.push_back(i);
    for (int r : rots) {
        u64 *d = cx.eval_key_storage(r);
        // [dnum][2][nT][N]: limb slot index modulo nT selects the modulus
        hk::fill_uniform_hash(cx.stream, cx.d_mod, cx.N, d, (size_t)cx.prm.dnum * 2 * cx.nT, cx.nT, seed + 7919ull * r);
    }
    cx.sync();
    return HYDIA_OK;
    API_END
}
int hydia_has_eval_key(hydia_ctx *ctx, int rot) {
    if (!ctx) return 0;
    if (rot == 0) return ctx->cx.relin_key.d != nullptr;
    auto it = ctx->cx.rot_keys.find(rot);
    return it != ctx->cx.rot_keys.end() && it->second.d != nullptr;
}
static int import_buf(Context &cx, u64 **slot, const uint64_t *data, size_t elems) {
    if (cx.keys_borrowed) throw StateError("hydia: this context borrows its keys from another context (re-key the owner)");
    if (!*slot) HIP_CHECK(hipMalloc((void **)slot, elems * sizeof(u64)));
    HIP_CHECK(hipMemcpy(*slot, data, elems * sizeof(u64), hipMemcpyHostToDevice));
    return HYDIA_OK;
}
int hydia_import_public_key(hydia_ctx *ctx, const uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    return import_buf(ctx->cx, &ctx->cx.d_pk, data, (size_t)2 * ctx->cx.nQ * ctx->cx.N);
    API_END
}
int hydia_import_secret_key(hydia_ctx *ctx, const uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    return import_buf(ctx->cx, &ctx->cx.d_sk, data, (size_t)ctx->cx.nT * ctx->cx.N);
    API_END
}
int hydia_export_public_key(hydia_ctx *ctx, uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    if (!ctx->cx.d_pk) return fail(HYDIA_ERR_STATE, "hydia: public key not loaded");
    ctx->cx.sync();
    HIP_CHECK(hipMemcpy(data, ctx->cx.d_pk, (size_t)2 * ctx->cx.nQ * ctx->cx.N * sizeof(u64), hipMemcpyDeviceToHost));
    return HYDIA_OK;
    API_END
}
int hydia_export_secret_key(hydia_ctx *ctx, uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    if (!ctx->cx.d_sk) return fail(HYDIA_ERR_STATE, "hydia: secret key not loaded");
    ctx->cx.sync();
    HIP_CHECK(hipMemcpy(data, ctx->cx.d_sk, (size_t)ctx->cx.nT * ctx->cx.N * sizeof(u64), hipMemcpyDeviceToHost));
    return HYDIA_OK;
    API_END
}

// ------------------------------------------------------------------ ciphertext handles
int hydia_ct_import(hydia_ctx *ctx, const uint64_t *data, uint32_t count, uint32_t n_polys, uint32_t n_limbs, double scale,
                    hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data && out, "null argument");
    REQUIRE(count >= 1 && (n_polys == 2 || n_polys == 3) && n_limbs >= 1 && (int)n_limbs <= ctx->cx.nQ, "bad ciphertext shape");
    Ct c(&ctx->cx, (int)count, (int)n_polys, (int)n_limbs, scale);
    ctx->cx.sync();
    HIP_CHECK(hipMemcpy(c.d, data, c.bytes(), hipMemcpyHostToDevice));
    *out = wrap(ctx, std::move(c));
    return HYDIA_OK;
    API_END
}
int hydia_ct_export(hydia_ctx *ctx, const hydia_ct *ct, uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct && data, "null argument");
    ctx->cx.sync();
    HIP_CHECK(hipMemcpy(data, ct->c.d, ct->c.bytes(), hipMemcpyDeviceToHost));
    return HYDIA_OK;
    API_END
}
int hydia_ct_shape(const hydia_ct *ct, uint32_t *count, uint32_t *n_polys, uint32_t *n_limbs, double *scale) {
    REQUIRE(ct, "null ct");
    if (count) *count = ct->c.X;
    if (n_polys) *n_polys = ct->c.npoly;
    if (n_limbs) *n_limbs = ct->c.nl;
    if (scale) *scale = ct->c.scale;
    return HYDIA_OK;
}
int hydia_ct_device_ptr(const hydia_ct *ct, void **ptr, size_t *bytes) {
    REQUIRE(ct && ptr, "null argument");
    *ptr = ct->c.d;
    if (bytes) *bytes = ct->c.bytes();
    return HYDIA_OK;
}
int hydia_ct_copy_to_device(hydia_ctx *ctx, const hydia_ct *ct, void *dev_dst) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct && dev_dst, "null argument");
    HIP_CHECK(hipMemcpyAsync(dev_dst, ct->c.d, ct->c.bytes(), hipMemcpyDeviceToDevice, ctx->cx.stream));
    ctx->cx.sync();
    return HYDIA_OK;
    API_END
}
int hydia_ct_from_device(hydia_ctx *ctx, const void *dev_ptr, uint32_t count, uint32_t n_polys, uint32_t n_limbs,
                         double scale, hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && dev_ptr && out, "null argument");
    REQUIRE(count >= 1 && (n_polys == 2 || n_polys == 3) && n_limbs >= 1 && (int)n_limbs <= ctx->cx.nQ, "bad ciphertext shape");
    Ct c(&ctx->cx, (int)count, (int)n_polys, (int)n_limbs, scale);
    HIP_CHECK(hipMemcpyAsync(c.d, dev_ptr, c.bytes(), hipMemcpyDeviceToDevice, ctx->cx.stream));
    ctx->cx.sync();  // the caller's buffer is free again, and later work on the context's stream is ordered behind the copy
    *out = wrap(ctx, std::move(c));
    return HYDIA_OK;
    API_END
}
// a handle over ciphertexts that stay in the CALLER's device memory (no copy): e.g. the all-gathered rotations of a rotation-split
// loop A.  The memory must stay valid and unchanged while the handle or any operation enqueued on it is alive.
int hydia_ct_view_device(hydia_ctx *ctx, void *dev_ptr, uint32_t count, uint32_t n_polys, uint32_t n_limbs, double scale, hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && dev_ptr && out, "null argument");
    REQUIRE(count >= 1 && (n_polys == 2 || n_polys == 3) && n_limbs >= 1 && (int)n_limbs <= ctx->cx.nQ, "bad ciphertext shape");
    hydia_ct *h = new hydia_ct;
    h->c.ctx = &ctx->cx;
    h->c.d = static_cast<u64 *>(dev_ptr);
    h->c.X = (int)count;
    h->c.npoly = (int)n_polys;
    h->c.nl = h->c.lstride = (int)n_limbs;
    h->c.scale = scale;
    h->c.view = true;
    h->owner = ctx;
    ctx->refs.fetch_add(1);
    *out = h;
    return HYDIA_OK;
    API_END
}
void hydia_ct_free(hydia_ct *ct) {
    if (!ct) return;
    hydia_ctx *owner = ct->owner;
    use_device(owner);  // the buffer goes back to the pool (or the context goes away) under the context's own GPU
    delete ct;
    if (owner && owner->refs.fetch_sub(1) == 1) delete owner;
}

// ------------------------------------------------------------------ receiver
int hydia_encrypt_query(hydia_ctx *ctx, const double *query, const uint8_t seed[32], uint64_t nonce, hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && query && seed && out, "null argument");
    REQUIRE(nonce < HYDIA_NONCE_LIMIT, "nonce must be below 2^40");
    *out = wrap(ctx, client_encrypt_query(ctx->cx, query, seed, nonce));
    return HYDIA_OK;
    API_END
}
int hydia_encrypt(hydia_ctx *ctx, const double *slots, uint32_t count, const uint8_t seed[32], uint64_t nonce0,
                  hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && slots && seed && out && count >= 1, "bad argument");
    REQUIRE(nonce0 < HYDIA_NONCE_LIMIT && nonce0 + count <= HYDIA_NONCE_LIMIT, "nonce must be below 2^40");
    *out = wrap(ctx, client_encrypt(ctx->cx, slots, (int)count, seed, nonce0));
    return HYDIA_OK;
    API_END
}
int hydia_decrypt(hydia_ctx *ctx, const hydia_ct *ct, double *out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct && out, "null argument");
    client_decrypt(ctx->cx, ct->c, out);
    return HYDIA_OK;
    API_END
}
/* receiver_hers.cpp:26-35 */
int hydia_decrypt_membership(hydia_ctx *ctx, const hydia_ct *ct, int *result) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct && result, "null argument");
    std::vector<double> v((size_t)ct->c.X * ctx->cx.slots);
    client_decrypt(ctx->cx, ct->c, v.data());
    *result = v[0] >= 1.0 ? 1 : 0;
    return HYDIA_OK;
    API_END
}
/* receiver_hers.cpp:37-54 */
int hydia_decrypt_index(hydia_ctx *ctx, const hydia_ct *cts, size_t *out, size_t cap, size_t *n_out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && cts && n_out, "null argument");
    const size_t S = ctx->cx.slots;
    std::vector<double> v((size_t)cts->c.X * S);
    client_decrypt(ctx->cx, cts->c, v.data());
    size_t cnt = 0;
    for (size_t i = 0; i < (size_t)cts->c.X; i++)
        for (size_t j = 0; j < S; j++)
            if (v[i * S + j] >= 1.0) {
                if (out && cnt < cap) out[cnt] = j + i * S;
                cnt++;
            }
    *n_out = cnt;
    return HYDIA_OK;
    API_END
}

// ------------------------------------------------------------------ enroller / database
size_t hydia_db_num_cts(const hydia_ctx *ctx, size_t n) {
    if (!ctx) return 0;
    const size_t dim = ctx->cx.prm.dim, per = ctx->cx.slots / dim;
    const size_t nblk = (n + dim - 1) / dim;
    return ((nblk + per - 1) / per) * dim;
}
int hydia_db_alloc(hydia_ctx *ctx, size_t n_vectors) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && n_vectors >= 1, "bad argument");
    ctx->cx.db_resize(n_vectors, hydia_db_num_cts(ctx, n_vectors), ctx->cx.prm.dim);
    ctx->cx.db_kind = 5;
    ctx->cx.db_babies = ctx->cx.prm.dim;
    return HYDIA_OK;
    API_END
}
int hydia_db_import_ct(hydia_ctx *ctx, size_t t, const uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    Context &cx = ctx->cx;
    if (!cx.d_db) return fail(HYDIA_ERR_STATE, "hydia: no database resident (call hydia_db_alloc)");
    REQUIRE(t < cx.db_cts, "ciphertext index out of range");
    const size_t e = (size_t)2 * cx.nQ * cx.N;
    u64 *tmp = cx.pool.get(e * sizeof(u64));
    cx.sync();
    HIP_CHECK(hipMemcpy(tmp, data, e * sizeof(u64), hipMemcpyHostToDevice));
    cx.db_store(t, tmp, 1);
    cx.sync();
    cx.pool.put(tmp);
    if (cx.db_kind == 0) {
        cx.db_kind = 5;
        cx.db_babies = cx.prm.dim;
    }
    return HYDIA_OK;
    API_END
}
int hydia_db_export_ct(hydia_ctx *ctx, size_t t, uint64_t *data) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data, "null argument");
    Context &cx = ctx->cx;
    if (!cx.d_db) return fail(HYDIA_ERR_STATE, "hydia: no database resident");
    REQUIRE(t < cx.db_cts, "ciphertext index out of range");
    const size_t e = (size_t)2 * cx.nQ * cx.N;
    u64 *tmp = cx.pool.get(e * sizeof(u64));
    cx.db_fetch(t, tmp, 1);
    cx.sync();
    HIP_CHECK(hipMemcpy(data, tmp, e * sizeof(u64), hipMemcpyDeviceToHost));
    cx.pool.put(tmp);
    return HYDIA_OK;
    API_END
}
int hydia_db_fill_random(hydia_ctx *ctx, size_t n_vectors, uint64_t seed) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && n_vectors >= 1, "bad argument");
    Context &cx = ctx->cx;
    const size_t cts = hydia_db_num_cts(ctx, n_vectors);
    const int babies = cx.babies_for((n_vectors + (size_t)cx.slots - 1) / (size_t)cx.slots);  // random residues: the cost model of the split auto picks
    cx.db_resize(n_vectors, cts, babies);
    const size_t chunk = (size_t)cx.prm.dim, e = (size_t)2 * cx.nQ * cx.N;
    u64 *tmp = cx.pool.get(chunk * e * sizeof(u64));
    for (size_t t0 = 0; t0 < cts; t0 += chunk) {
        const size_t cnt = std::min(chunk, cts - t0);
        hk::fill_uniform_hash(cx.stream, cx.d_mod, cx.N, tmp, cnt * 2 * cx.nQ, cx.nQ, seed + 0x9E37ull * t0);
        cx.db_store(t0, tmp, (int)cnt);
    }
    cx.sync();
    cx.pool.put(tmp);
    cx.db_babies = babies;
    cx.db_kind = cx.db_babies < cx.prm.dim ? 6 : 5;
    return HYDIA_OK;
    API_END
}
// matvec: 0 = the context's own policy (hydia_set_matvec; auto looks at the blocks THIS context holds), 1 hoisted, otherwise the baby
// count — a sharded enrolment passes the group-wide decision so that every shard of one database uses the same split
int hydia_db_enroll_shard_ex(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32], size_t first_block, int matvec) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && db && seed && n >= 1 && matvec >= 0, "bad argument");
    Context &cx = ctx->cx;
    const size_t G = (n + (size_t)cx.slots - 1) / (size_t)cx.slots;
    const int B = cx.babies_for(G, matvec);
    cx.db_kind = 0;
    cx.db_resize(n, hydia_db_num_cts(ctx, n), B);
    client_enroll(cx, db, n, seed, first_block, B);
    cx.db_kind = B < cx.prm.dim ? 6 : 5;
    cx.db_babies = B;
    return HYDIA_OK;
    API_END
}
int hydia_db_enroll(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32]) { return hydia_db_enroll_shard_ex(ctx, db, n, seed, 0, 0); }
int hydia_db_enroll_shard(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32], size_t first_block) {
    return hydia_db_enroll_shard_ex(ctx, db, n, seed, first_block, 0);
}
int hydia_set_matvec(hydia_ctx *ctx, int mode) {
    API_BEGIN
    REQUIRE(ctx && mode >= 0, "mat-vec mode is 0 (auto), 1 (hoisted) or a baby count");
    if (mode > 1) (void)ctx->cx.babies_for(1, mode);  // validates: a power of two dividing vector_dim
    ctx->cx.matvec_mode = mode;
    return HYDIA_OK;
    API_END
}
/* form of an IMPORTED database: babies == vector_dim (hoisted: what hydia_db_alloc + hydia_db_import_ct assume, the reference
 * enroller's ciphertexts) or the baby count its diagonals were pre-rotated for */
int hydia_db_set_babies(hydia_ctx *ctx, int babies) {
    API_BEGIN
    REQUIRE(ctx, "null argument");
    Context &cx = ctx->cx;
    if (!cx.d_db || cx.db_cts == 0 || (cx.db_kind != 5 && cx.db_kind != 6)) return fail(HYDIA_ERR_STATE, "hydia: no diagonal database resident");
    // a DECLARED form: vector_dim (hoisted) or a power of two >= 2 dividing it — never 0 / 1, which mean "auto" / "hoisted" only in
    // hydia_set_matvec's policy and would silently mark an imported hoisted database as pre-rotated
    const int dim = cx.prm.dim;
    REQUIRE(babies == dim || (babies >= 2 && babies < dim && (babies & (babies - 1)) == 0 && dim % babies == 0),
            "the declared baby count must be vector_dim or a power of two >= 2 dividing it");
    const int B = babies;
    use_device(ctx);
    cx.db_relayout(B);  // no-op unless the resident order was chosen for another form (hydia_db_alloc assumes the hoisted one)
    cx.db_babies = B;
    cx.db_kind = cx.db_babies < cx.prm.dim ? 6 : 5;
    return HYDIA_OK;
    API_END
}
int hydia_get_matvec(const hydia_ctx *ctx) { return ctx ? ctx->cx.matvec_mode : -1; }
int hydia_db_kind(const hydia_ctx *ctx) { return ctx ? ctx->cx.db_kind : 0; }
int hydia_db_babies(const hydia_ctx *ctx) { return ctx && (ctx->cx.db_kind == 5 || ctx->cx.db_kind == 6) ? ctx->cx.db_babies : 0; }
int hydia_auto_babies(const hydia_ctx *ctx, size_t blocks) { return ctx ? ctx->cx.babies_for(blocks) : 0; }
int hydia_hers_db_enroll(hydia_ctx *ctx, double *db, size_t n, const uint8_t seed[32]) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && db && seed && n >= 1, "bad argument");
    Context &cx = ctx->cx;
    const size_t G = (n + cx.slots - 1) / cx.slots;  // enroller_hers.cpp:59-60
    cx.db_resize(n, G * cx.prm.dim, -1);
    client_hers_enroll(cx, db, n, seed);
    cx.db_kind = 4;
    return HYDIA_OK;
    API_END
}
int hydia_hers_encrypt_query(hydia_ctx *ctx, const double *query, const uint8_t seed[32], uint64_t nonce0, hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && query && seed && out, "null argument");
    REQUIRE(nonce0 < HYDIA_NONCE_LIMIT && nonce0 + ctx->cx.prm.dim <= HYDIA_NONCE_LIMIT, "nonce must be below 2^40");
    *out = wrap(ctx, client_hers_encrypt_query(ctx->cx, query, seed, nonce0));
    return HYDIA_OK;
    API_END
}
int hydia_db_save(hydia_ctx *ctx, const char *path) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && path, "null argument");
    ctx->cx.db_save(path);
    return HYDIA_OK;
    API_END
}
int hydia_db_load(hydia_ctx *ctx, const char *path) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && path, "null argument");
    ctx->cx.db_load(path);
    return HYDIA_OK;
    API_END
}
int hydia_db_stats(const hydia_ctx *ctx, size_t *n_vectors, size_t *n_cts, size_t *bytes) {
    REQUIRE(ctx, "null ctx");
    if (n_vectors) *n_vectors = ctx->cx.db_vectors;
    if (n_cts) *n_cts = ctx->cx.db_cts;
    if (bytes) *bytes = ctx->cx.db_cts * ctx->cx.db_layout().ct_bytes;
    return HYDIA_OK;
}
int hydia_db_group(const hydia_ctx *ctx) { return ctx && ctx->cx.d_db ? ctx->cx.db_lay.seq : 0; }
int hydia_db_residue_bits(const hydia_ctx *ctx) {
    if (!ctx || !ctx->cx.d_db) return 0;
    const DbLayout &L = ctx->cx.db_lay;
    return !L.packed ? 64 : (L.seq && L.bits46) ? 46 : 48;
}

// ------------------------------------------------------------------ sender
#define SENDER_CALL(expr)                                 \
    API_BEGIN                                             \
    use_device(ctx);                                      \
    REQUIRE(ctx && query && out, "null argument");        \
    *out = wrap(ctx, expr);                                  \
    return HYDIA_OK;                                      \
    API_END
int hydia_rotate_query(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.rotate_query(query->c)) }
int hydia_compute_similarity(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.similarity(query->c)) }
int hydia_rotate_query_range_into(hydia_ctx *ctx, const hydia_ct *query, uint32_t first, uint32_t count, void *dev_dst) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && query && dev_dst, "null argument");
    REQUIRE(first <= (uint32_t)ctx->cx.prm.dim && count <= (uint32_t)ctx->cx.prm.dim - first, "rotation range outside 0 .. vector_dim");
    ctx->cx.rotate_query_range(query->c, (int)first, (int)count, static_cast<u64 *>(dev_dst));
    return HYDIA_OK;
    API_END
}
int hydia_rotate_query_range(hydia_ctx *ctx, const hydia_ct *query, uint32_t first, uint32_t count, hydia_ct **out) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && query && out && count >= 1, "bad argument");
    // validated on the unsigned values BEFORE anything is allocated (first + count must not wrap)
    REQUIRE(first <= (uint32_t)ctx->cx.prm.dim && count <= (uint32_t)ctx->cx.prm.dim - first, "rotation range outside 0 .. vector_dim");
    Ct r(&ctx->cx, (int)count, 2, query->c.nl, query->c.scale);
    ctx->cx.rotate_query_range(query->c, (int)first, (int)count, r.d);
    *out = wrap(ctx, std::move(r));
    return HYDIA_OK;
    API_END
}
int hydia_compute_similarity_rotated(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.similarity_rot(query->c)) }
int hydia_index_scenario_rotated(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.index_scenario_rot(query->c)) }
int hydia_index_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.index_scenario(query->c)) }
int hydia_membership_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.membership_scenario(query->c)) }
int hydia_chebyshev_compare(hydia_ctx *ctx, const hydia_ct *query, double delta, size_t sign_depth, hydia_ct **out) {
    SENDER_CALL(ctx->cx.chebyshev_compare(query->c, delta, (int)sign_depth))
}
int hydia_sum_and_evalsum(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.sum_and_evalsum(query->c)) }
int hydia_add_many(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.add_many(query->c)) }
int hydia_eval_sum(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.eval_sum(query->c)) }
int hydia_ct_add_raw(hydia_ctx *ctx, hydia_ct *acc, const void *dev_src, int src_device) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && acc && dev_src, "null argument");
    Context &cx = ctx->cx;
    u64 *tmp = cx.pool.get(acc->c.bytes());
    // on the context's own stream: a plain device-to-device hipMemcpy would run on the null stream, unordered with it
    if (src_device < 0 || src_device == cx.device)
        HIP_CHECK(hipMemcpyAsync(tmp, dev_src, acc->c.bytes(), hipMemcpyDeviceToDevice, cx.stream));
    else
        HIP_CHECK(hipMemcpyPeerAsync(tmp, cx.device, dev_src, src_device, acc->c.bytes(), cx.stream));
    cx.add_raw_inplace(acc->c, tmp);
    cx.sync();
    cx.pool.put(tmp);
    return HYDIA_OK;
    API_END
}
int hydia_ct_mod_reduce(hydia_ctx *ctx, hydia_ct *ct) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct, "null argument");
    ctx->cx.mod_reduce_inplace(ct->c);
    return HYDIA_OK;
    API_END
}
int hydia_hers_compute_similarity(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.hers_similarity(query->c)) }
int hydia_hers_index_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.hers_index_scenario(query->c)) }
int hydia_hers_membership_scenario(hydia_ctx *ctx, const hydia_ct *query, hydia_ct **out) { SENDER_CALL(ctx->cx.hers_membership_scenario(query->c)) }

// ------------------------------------------------------------------ primitives
int hydia_ntt(hydia_ctx *ctx, uint64_t *data, uint32_t count, uint32_t m, int inverse) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && data && count >= 1 && (int)m < ctx->cx.nT, "bad argument");
    Context &cx = ctx->cx;
    const size_t bytes = (size_t)count * cx.N * sizeof(u64);
    u64 *d = cx.pool.get(bytes);
    cx.sync();
    HIP_CHECK(hipMemcpy(d, data, bytes, hipMemcpyHostToDevice));
    LimbSel s = cx.sel_range((int)m, (int)m + 1);
    if (inverse) cx.ntt_inv(d, d, cx.N, cx.N, (int)count, s, cx.scale_ninv(s));
    else cx.ntt_fwd(d, cx.N, (int)count, s);
    cx.sync();
    HIP_CHECK(hipMemcpy(data, d, bytes, hipMemcpyDeviceToHost));
    cx.pool.put(d);
    return HYDIA_OK;
    API_END
}
int hydia_eval_rotate(hydia_ctx *ctx, const hydia_ct *query, int rot, hydia_ct **out) { SENDER_CALL(ctx->cx.rotate(query->c, rot)) }
int hydia_eval_mult(hydia_ctx *ctx, const hydia_ct *query, const hydia_ct *b, hydia_ct **out) {
    if (!b) return fail(HYDIA_ERR_ARG, "null argument");
    SENDER_CALL(ctx->cx.mult(query->c, b->c))
}
int hydia_eval_mult_no_relin(hydia_ctx *ctx, const hydia_ct *query, const hydia_ct *b, hydia_ct **out) {
    if (!b) return fail(HYDIA_ERR_ARG, "null argument");
    SENDER_CALL(ctx->cx.mult_norelin(query->c, b->c))
}
int hydia_relinearize(hydia_ctx *ctx, hydia_ct *ct) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct, "null argument");
    ctx->cx.relinearize(ct->c);
    return HYDIA_OK;
    API_END
}
int hydia_rescale(hydia_ctx *ctx, hydia_ct *ct) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct, "null argument");
    ctx->cx.rescale(ct->c);
    return HYDIA_OK;
    API_END
}
int hydia_eval_add(hydia_ctx *ctx, hydia_ct *a, const hydia_ct *b) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && a && b, "null argument");
    ctx->cx.add_inplace(a->c, b->c);
    return HYDIA_OK;
    API_END
}
int hydia_level_reduce(hydia_ctx *ctx, hydia_ct *ct, uint32_t n_limbs) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ct && n_limbs >= 1, "bad argument");
    if ((int)n_limbs < ct->c.nl) {
        Ct v = ct->c.alias((int)n_limbs);
        ct->c = ctx->cx.clone(v);
    }
    return HYDIA_OK;
    API_END
}

// ------------------------------------------------------------------ measurement
int hydia_byte_ledger(int enable, char *out, size_t cap, size_t *needed) {
    const size_t n = (out || needed) ? hk::ledger_dump(out, cap) : 0;
    if (needed) *needed = n;
    if (enable >= 0) hk::ledger_enable(enable != 0);  // (re)starts or stops the recording; -1 = just read
    return HYDIA_OK;
}
/* transform `polys` polynomials x limbs [first_mod, first_mod + n_mods) of pooled scratch memory `iters` times (in place; the
 * data is whatever the pool holds — cost is data independent) and report the HIP-event time per iteration */
int hydia_bench_ntt(hydia_ctx *ctx, uint32_t polys, uint32_t first_mod, uint32_t n_mods, int inverse, uint32_t iters, double *ms_per_iter) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && ms_per_iter && polys >= 1 && n_mods >= 1 && iters >= 1 && (int)(first_mod + n_mods) <= ctx->cx.nT, "bad argument");
    Context &cx = ctx->cx;
    const size_t outer = (size_t)n_mods * cx.N;
    u64 *buf = cx.pool.get((size_t)polys * outer * sizeof(u64));
    HIP_CHECK(hipMemsetAsync(buf, 0, (size_t)polys * outer * sizeof(u64), cx.stream));
    const LimbSel s = cx.sel_range((int)first_mod, (int)(first_mod + n_mods));
    auto once = [&] {
        if (inverse) cx.ntt_inv(buf, buf, outer, outer, (int)polys, s, cx.scale_ninv(s));
        else cx.ntt_fwd(buf, outer, (int)polys, s);
    };
    once();
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, cx.stream));
    for (uint32_t i = 0; i < iters; i++) once();
    HIP_CHECK(hipEventRecord(b, cx.stream));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    cx.pool.put(buf);
    *ms_per_iter = ms / iters;
    return HYDIA_OK;
    API_END
}
int hydia_kernel_time(hydia_ctx *ctx, const char *name, double *total_ms, uint64_t *launches) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx && name, "null argument");
    ctx->cx.timer_collect();
    auto it = ctx->cx.timers.find(name);
    if (total_ms) *total_ms = it == ctx->cx.timers.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == ctx->cx.timers.end() ? 0 : (uint64_t)it->second.launches;
    return HYDIA_OK;
    API_END
}
int hydia_kernel_time_reset(hydia_ctx *ctx) {
    API_BEGIN
    use_device(ctx);
    REQUIRE(ctx, "null ctx");
    ctx->cx.timer_collect();
    ctx->cx.timers.clear();
    return HYDIA_OK;
    API_END
}

}  // extern "C"

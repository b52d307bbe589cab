// image_matching_amd/csrc/context.cpp — CKKS-RNS context for the HyDia path on MI355X.
//
// Parameter derivation replaces GenCryptoContext(HEStd_128_classic, depth 11, ScalingModSize 45, FIXEDMANUAL) of
// /root/reference/src/main.cpp:169-179: N = 2^15, Q = one 60-bit + eleven ~45-bit NTT primes, hybrid key switching
// with dnum = 3 (alpha = 4 limbs per digit) and four 60-bit special primes P.  The prime/root selection rule is this
// framework's own deterministic specification (DESIGN.md §"RNS parameters"); tests check it against the oracle.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <sys/stat.h>

#include "hydia_core.h"

namespace hydia {

// ------------------------------------------------------------------ host number theory
u64 mulmod_u64(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
u64 powmod_u64(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod_u64(r, a, q);
        a = mulmod_u64(a, a, q);
    }
    return r;
}
u64 invmod_u64(u64 a, u64 q) { return powmod_u64(a % q, q - 2, q); }

static bool miller_rabin(u64 n) {
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull})
        if (n % p == 0) return n == p;
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) d >>= 1, r++;
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powmod_u64(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < r && witness; i++) {
            x = mulmod_u64(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}
// NTT primes are = 1 mod 2N.  next_ntt_prime: smallest such prime >= lo; prev_ntt_prime: largest such prime < hi.
static u64 next_ntt_prime(u64 lo, u64 M) {
    u64 c = lo + (M + 1 - lo % M) % M;
    while (!miller_rabin(c)) c += M;
    return c;
}
static u64 prev_ntt_prime(u64 hi, u64 M) {
    u64 c = hi - 1;
    c -= (c % M + M - 1) % M;
    while (!miller_rabin(c)) c -= M;
    return c;
}
static u64 find_psi(u64 q, u64 M) {
    for (u64 x = 2;; x++) {
        u64 r = powmod_u64(x, (q - 1) / M, q);
        if (powmod_u64(r, M / 2, q) == q - 1) return r;  // order exactly 2N
    }
}
static unsigned bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++, x >>= 1) r = (r << 1) | (x & 1);
    return r;
}
static u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

static ModC make_modc(u64 q, int N) {
    ModC m{};
    m.q = q;
    int k = 64 - __builtin_clzll(q);
    m.ks = k - 2;
    m.mu = (u64)((((u128)1) << (k + 62)) / q);
    m.r64 = (u64)((((u128)1) << 64) / q);
    u128 hi = (((u128)1) << 64) / q, rem = (((u128)1) << 64) % q;
    u128 full = (hi << 64) + ((rem << 64) / q);
    m.r0 = (u64)full;
    m.r1 = (u64)(full >> 64);
    m.ninv = invmod_u64((u64)N, q);
    m.ninv_sh = shoup(m.ninv, q);
    return m;
}

u64 double_to_mod(double v, u64 q) {
    bool neg = v < 0;
    double a = std::fabs(v);
    u64 r;
    if (a < 9223372036854775808.0) {
        r = (u64)std::llrint(a) % q;
    } else {
        int e;
        double m = std::frexp(a, &e);
        u64 mant = (u64)std::ldexp(m, 53);
        r = mulmod_u64(mant % q, powmod_u64(2, (u64)(e - 53), q), q);
    }
    return neg ? (r ? q - r : 0) : r;
}

// ------------------------------------------------------------------ pool
Pool::~Pool() { trim(); }
u64 *Pool::get(size_t bytes) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    bytes = (bytes + 255) & ~(size_t)255;
    auto &fl = free_[cur];
    auto it = fl.lower_bound(bytes);
    u64 *p;
    if (it != fl.end() && it->first <= bytes + bytes / 4 + 4096) {
        p = it->second;
        bytes_cached -= it->first;
        bytes = it->first;
        fl.erase(it);
    } else {
        hipError_t e = hipMalloc((void **)&p, bytes);
        if (e != hipSuccess) {
            trim();
            HIP_CHECK(hipMalloc((void **)&p, bytes));
        }
    }
    size_[p] = {bytes, cur};
    bytes_live += bytes;
    peak = std::max(peak, bytes_live + bytes_cached);
    return p;
}
void Pool::put(u64 *p) {
    if (!p) return;
    std::lock_guard<std::recursive_mutex> lk(mu_);
    auto it = size_.find(p);
    if (it == size_.end()) return;
    free_[it->second.second].emplace(it->second.first, p);  // back to the lane that owns it
    bytes_live -= it->second.first;
    bytes_cached += it->second.first;
    size_.erase(it);
}
void Pool::trim() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    for (auto &fl : free_)
        for (auto &kv : fl.second) (void)hipFree(kv.second);
    free_.clear();
    bytes_cached = 0;
}

// ------------------------------------------------------------------ Ct
Ct::Ct(Context *c, int X_, int npoly_, int nl_, double scale_)
    : ctx(c), X(X_), npoly(npoly_), nl(nl_), lstride(nl_), scale(scale_) {
    d = c->pool.get(bytes());
}
Ct &Ct::operator=(Ct &&o) noexcept {
    if (this != &o) {
        if (d && !view && ctx) ctx->pool.put(d);
        ctx = o.ctx; d = o.d; X = o.X; npoly = o.npoly; nl = o.nl; lstride = o.lstride; scale = o.scale; view = o.view;
        o.d = nullptr;
    }
    return *this;
}
Ct::~Ct() {
    if (d && !view && ctx) ctx->pool.put(d);
}
size_t Ct::poly_elems() const { return (size_t)lstride * ctx->N; }
Ct Ct::alias(int nl_) const {
    Ct v;
    v.ctx = ctx; v.d = d; v.X = X; v.npoly = npoly; v.nl = nl_ < nl ? nl_ : nl; v.lstride = lstride; v.scale = scale;
    v.view = true;
    return v;
}

// ------------------------------------------------------------------ context
HostParams::HostParams(const Params &p) : prm(p) {
    if (p.logN < 11 || p.logN > 16) throw std::runtime_error("hydia: log_n must be in [11,16]");
    if (p.dnum < 1 || p.mult_depth < 1) throw std::runtime_error("hydia: bad dnum / depth");
    N = 1 << p.logN;
    slots = N / 2;
    nQ = p.mult_depth + 1;
    alpha = (nQ + p.dnum - 1) / p.dnum;
    if (alpha > HY_MAX_DIGIT) throw std::runtime_error("hydia: digit too wide");
    if (p.dim < 2 || (p.dim & (p.dim - 1)) || p.dim > slots) throw std::runtime_error("hydia: dim must be a power of two, 2 <= dim <= N/2");
    delta = std::ldexp(1.0, p.scale_bits);
    const u64 M = 2ull * N;

    if (!p.custom_q.empty()) {
        // an externally generated chain (OpenFHE's, read through GetElementParams()): validate what the kernels rely on
        nP = p.custom_nP;
        nT = nQ + nP;
        if (nP < 1 || (int)p.custom_q.size() != nT || nT > HY_MAX_MODS) throw std::runtime_error("hydia: custom moduli: need n_q + n_p primes");
        if (!p.custom_psi.empty() && (int)p.custom_psi.size() != nT) throw std::runtime_error("hydia: custom roots: need n_q + n_p values");
        q = p.custom_q;
        for (int m = 0; m < nT; m++) {
            if (q[m] % M != 1 || (q[m] >> 60) || !miller_rabin(q[m]))
                throw std::runtime_error("hydia: custom modulus " + std::to_string(m) + " must be a prime < 2^60 that is 1 mod 2N");
            for (int i = 0; i < m; i++)
                if (q[i] == q[m]) throw std::runtime_error("hydia: custom moduli must be distinct");
        }
    } else {
    q.assign(nQ, 0);
    // scaling primes, assigned from the last limb down, alternating above / below 2^scale_bits
    u64 up = next_ntt_prime(1ull << p.scale_bits, M), dn = up;
    for (int i = 0; i < nQ - 1; i++) {
        int j = nQ - 1 - i;
        if (i == 0) q[j] = up;
        else if (i & 1) q[j] = dn = prev_ntt_prime(dn, M);
        else q[j] = up = next_ntt_prime(up + 1, M);
    }
    u64 cur = prev_ntt_prime(1ull << p.first_bits, M);
    if (p.first_bits == p.scale_bits)
        while (std::find(q.begin() + 1, q.end(), cur) != q.end()) cur = prev_ntt_prime(cur, M);
    q[0] = cur;
    int digit_bits = 0;
    for (int j = 0; j < alpha && j < nQ; j++) digit_bits += (j == 0 ? p.first_bits : p.scale_bits + 1);
    nP = std::max(1, (digit_bits + 59) / 60);
    nT = nQ + nP;
    if (nT > HY_MAX_MODS) throw std::runtime_error("hydia: too many limbs");
    u64 pc = (p.first_bits == 60) ? q[0] : (1ull << 60);
    for (int k = 0; k < nP; k++) q.push_back(pc = prev_ntt_prime(pc, M));
    }

    psi.resize(nT);
    mod.resize(nT);
    for (int m = 0; m < nT; m++) {
        psi[m] = p.custom_psi.empty() ? find_psi(q[m], M) : p.custom_psi[m];
        if (powmod_u64(psi[m], M / 2, q[m]) != q[m] - 1) throw std::runtime_error("hydia: root " + std::to_string(m) + " is not a primitive 2N-th root of unity");
        mod[m] = make_modc(q[m], N);
    }
    P_mod_q.resize(nQ);
    Pinv_mod_q.resize(nQ);
    for (int j = 0; j < nQ; j++) {
        u64 pr = 1;
        for (int k = 0; k < nP; k++) pr = mulmod_u64(pr, q[nQ + k] % q[j], q[j]);
        P_mod_q[j] = pr;
        Pinv_mod_q[j] = invmod_u64(pr, q[j]);
    }
    Phat_inv.resize(nP);
    Phat_mod_q.assign(nP, std::vector<u64>(nQ));
    for (int k = 0; k < nP; k++) {
        u64 pk = q[nQ + k], pr = 1;
        for (int i = 0; i < nP; i++)
            if (i != k) pr = mulmod_u64(pr, q[nQ + i] % pk, pk);
        Phat_inv[k] = invmod_u64(pr, pk);
        for (int j = 0; j < nQ; j++) {
            u64 v = 1;
            for (int i = 0; i < nP; i++)
                if (i != k) v = mulmod_u64(v, q[nQ + i] % q[j], q[j]);
            Phat_mod_q[k][j] = v;
        }
    }
    ql_inv.assign(nQ, std::vector<u64>(nQ, 0));
    for (int l = 0; l < nQ; l++)
        for (int j = 0; j < l; j++) ql_inv[l][j] = invmod_u64(q[l] % q[j], q[j]);

}

void HostParams::twiddles(int m, std::vector<u64> &tw, std::vector<u64> &tws, std::vector<u64> &itw,
                          std::vector<u64> &itws) const {
    std::vector<u64> pw(N), ipw(N);
    u64 ipsi = invmod_u64(psi[m], q[m]);
    pw[0] = ipw[0] = 1;
    for (int i = 1; i < N; i++) {
        pw[i] = mulmod_u64(pw[i - 1], psi[m], q[m]);
        ipw[i] = mulmod_u64(ipw[i - 1], ipsi, q[m]);
    }
    tw.resize(N); tws.resize(N); itw.resize(N); itws.resize(N);
    for (int k = 0; k < N; k++) {
        unsigned r = bitrev((unsigned)k, prm.logN);
        tw[k] = pw[r];
        tws[k] = shoup(pw[r], q[m]);
        itw[k] = ipw[r];
        itws[k] = shoup(ipw[r], q[m]);
    }
}

Context::Context(const Params &p, int dev) : HostParams(p), device(dev) {
    // ---- device side
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    lane_stream.push_back(stream);
    if (const char *e = getenv("HYDIA_LANES")) nlanes = std::max(1, std::min(8, atoi(e)));
    if (getenv("HYDIA_NO_PROD_FUSE")) prod_fuse = false;
    if (getenv("HYDIA_DB_48BIT")) db_bits46_ok = false;
    if (getenv("HYDIA_NO_CSUB_FUSE")) prod_fuse_csub = false;
    if (getenv("HYDIA_NO_KS_FUSE")) ks_fuse = false;
    if (getenv("HYDIA_NO_RESCALE_CF")) rescale_cf = false;
    for (int k = 1; k < nlanes; k++) {
        hipStream_t st;
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        lane_stream.push_back(st);
    }
    size_t tb = (size_t)nT * N * sizeof(u64);
    std::vector<u64> tw((size_t)nT * N), tws((size_t)nT * N), itw((size_t)nT * N), itws((size_t)nT * N);
    for (int m = 0; m < nT; m++) {
        std::vector<u64> a, b, c, d;
        twiddles(m, a, b, c, d);
        std::copy(a.begin(), a.end(), tw.begin() + (size_t)m * N);
        std::copy(b.begin(), b.end(), tws.begin() + (size_t)m * N);
        std::copy(c.begin(), c.end(), itw.begin() + (size_t)m * N);
        std::copy(d.begin(), d.end(), itws.begin() + (size_t)m * N);
    }
    HIP_CHECK(hipMalloc((void **)&d_mod, sizeof(ModC) * nT));
    HIP_CHECK(hipMalloc((void **)&d_tw, tb));
    HIP_CHECK(hipMalloc((void **)&d_tw_sh, tb));
    HIP_CHECK(hipMalloc((void **)&d_itw, tb));
    HIP_CHECK(hipMalloc((void **)&d_itw_sh, tb));
    HIP_CHECK(hipMemcpy(d_mod, mod.data(), sizeof(ModC) * nT, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_tw, tw.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_tw_sh, tws.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_itw, itw.data(), tb, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_itw_sh, itws.data(), tb, hipMemcpyHostToDevice));
    {   // interleaved (w, w') pairs for the register-radix kernels
        std::vector<u64> pr(2 * (size_t)nT * N), ipr(2 * (size_t)nT * N);
        for (size_t i = 0; i < (size_t)nT * N; i++) {
            pr[2 * i] = tw[i];
            pr[2 * i + 1] = tws[i];
            ipr[2 * i] = itw[i];
            ipr[2 * i + 1] = itws[i];
        }
        HIP_CHECK(hipMalloc((void **)&d_twp, 2 * tb));
        HIP_CHECK(hipMalloc((void **)&d_itwp, 2 * tb));
        HIP_CHECK(hipMemcpy(d_twp, pr.data(), 2 * tb, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_itwp, ipr.data(), 2 * tb, hipMemcpyHostToDevice));
        // FP64 path: (w, w/q) as doubles (exact for the <= 47-bit primes; never read for the 60-bit ones)
        std::vector<double> fr(2 * (size_t)nT * N), ifr(2 * (size_t)nT * N);
        for (int m = 0; m < nT; m++)
            for (int k = 0; k < N; k++) {
                const size_t i = (size_t)m * N + k;
                const double qd = (double)q[m];
                fr[2 * i] = (double)tw[i];
                fr[2 * i + 1] = (double)tw[i] / qd;
                ifr[2 * i] = (double)itw[i];
                ifr[2 * i + 1] = (double)itw[i] / qd;
            }
        HIP_CHECK(hipMalloc((void **)&d_twf, 2 * tb));
        HIP_CHECK(hipMalloc((void **)&d_itwf, 2 * tb));
        HIP_CHECK(hipMemcpy(d_twf, fr.data(), 2 * tb, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_itwf, ifr.data(), 2 * tb, hipMemcpyHostToDevice));
        // the same twiddles alone (w / q is then formed as w * (1/q) in the kernel: one multiply for half the bytes per load)
        std::vector<double> dr((size_t)nT * N), idr((size_t)nT * N);
        for (size_t i = 0; i < (size_t)nT * N; i++) {
            dr[i] = (double)tw[i];
            idr[i] = (double)itw[i];
        }
        HIP_CHECK(hipMalloc((void **)&d_twd, tb));
        HIP_CHECK(hipMalloc((void **)&d_itwd, tb));
        HIP_CHECK(hipMemcpy(d_twd, dr.data(), tb, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d_itwd, idr.data(), tb, hipMemcpyHostToDevice));
    }
    tabs = NttTables{d_tw, d_tw_sh, d_itw, d_itw_sh, d_mod, (const ulonglong2 *)d_twp, (const ulonglong2 *)d_itwp,
                     (const ulonglong2 *)d_twf, (const ulonglong2 *)d_itwf, (const double *)d_twd, (const double *)d_itwd, 0u};
    if (getenv("HYDIA_NTT_INT")) tabs.twf = tabs.itwf = nullptr;  // A/B switch: integer butterflies for every limb
    getenv_int_arith = getenv("HYDIA_NTT_INT") != nullptr;
    tabs.fp_mask = 0;
    for (int m = 0; m < nT; m++)
        if (tabs.twf != nullptr && mod[m].ks + 2 <= 47) tabs.fp_mask |= 1u << m;
    {
        const char *sw = getenv("HYDIA_NTT_1PASS");
        tabs.one_pass = !sw ? 0 : (sw[0] == 'p' ? 2 : 1);
        const char *mn = getenv("HYDIA_NTT_1PASS_MIN");
        tabs.one_pass_min = mn ? atoi(mn) : 1024;
        tabs.two_ip_launches = getenv("HYDIA_RELIN_TWO_IP_LAUNCHES") ? 1 : 0;
        tabs.no_drop_in_ip = getenv("HYDIA_NO_DROP_IN_IP") ? 1 : 0;
        tabs.int_epilogue = getenv("HYDIA_INT_EPILOGUE") ? 1 : 0;
        const char *g = getenv("HYDIA_IP_GROUP");
        tabs.ip_group = g && atoi(g) > 0 ? atoi(g) : 8;
        tabs.generic = getenv("HYDIA_NTT_GENERIC") ? 1 : 0;
        tabs.cf_wide = getenv("HYDIA_COLFUSE_WIDE") ? 1 : 0;
        tabs.p2_wg_sync = getenv("HYDIA_P2_WG_SYNC") ? 1 : 0;
        tabs.no_tw_lds = getenv("HYDIA_NO_TW_LDS") ? 1 : 0;
    }
    modup_per_digit = getenv("HYDIA_MODUP_PER_DIGIT") != nullptr;
    loop_a_separate_ip = getenv("HYDIA_LOOPA_SEPARATE_IP") != nullptr;
    loop_a_int_ip = getenv("HYDIA_LOOPA_INT_IP") != nullptr;
    loop_a_limb_fastest = getenv("HYDIA_LOOPA_LIMB_FASTEST") != nullptr;
    relin_separate_intt = getenv("HYDIA_RELIN_SEPARATE_INTT") != nullptr;
    tabs.pm_mask = 0;  // OpenFHE's 60-bit primes sit just below 2^60 (q_0 = 2^60 - 0x3ffff, ...): HYDIA_NTT_INT / HYDIA_NTT_NO_PM keep Harvey [0, 4q)
    if (!getenv("HYDIA_NTT_INT") && !getenv("HYDIA_NTT_NO_PM"))
        for (int m = 0; m < nT; m++)
            if (!((tabs.fp_mask >> m) & 1u) && mod[m].q < (1ull << 60) && (1ull << 60) - mod[m].q < (1ull << 24)) tabs.pm_mask |= 1u << m;
    if (const char *e = getenv("HYDIA_TENSOR_BPP")) tensor_bpp = atoi(e);
    if (const char *e = getenv("HYDIA_TENSOR_NW")) tensor_nw = atoi(e);
    merge_rescale = getenv("HYDIA_NO_MERGE_RESCALE") == nullptr;
    fuse_ip = getenv("HYDIA_NO_FUSE_IP") == nullptr;
    fork_products = getenv("HYDIA_NO_FORK") == nullptr;
    fuse_loop_a = getenv("HYDIA_NO_FUSE_LOOPA") == nullptr;
    colfuse = getenv("HYDIA_NO_COLFUSE") == nullptr;
    if (const char *e = getenv("HYDIA_MATVEC")) {  // auto | hoisted | bsgs | <baby count>
        if (e[0] == 'h') matvec_mode = 1;
        else if (e[0] == 'b') matvec_mode = bsgs_babies();
        else if (e[0] >= '1' && e[0] <= '9') matvec_mode = babies_for(1, atoi(e));
        else matvec_mode = 0;
    }
    rot_packed = getenv("HYDIA_KEYS_UNPACKED") == nullptr;
    db_packed = getenv("HYDIA_DB_UNPACKED") == nullptr;
    db_seq_ok = getenv("HYDIA_DB_CT_MAJOR") == nullptr;
    for (int j = 1; j < nQ; j++)
        if (q[j] >> 48) db_packed = false;
    HIP_CHECK(hipMalloc((void **)&d_rotptrs, sizeof(u64 *) * (size_t)p.dim));
    HIP_CHECK(hipMalloc((void **)&d_rotgalois, sizeof(unsigned) * (size_t)p.dim));
    HIP_CHECK(hipMalloc((void **)&d_rotginv, sizeof(unsigned) * (size_t)p.dim));

}

Context::~Context() {
    (void)hipSetDevice(device);
    for (auto st : lane_stream) (void)hipStreamSynchronize(st);
    stream = lane_stream.empty() ? stream : lane_stream[0];
    for (auto &kv : timers)
        for (auto &ev : kv.second.pending) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    pool.trim();
    for (auto &kv : modup_plans)
        if (kv.second.d_tabs) (void)hipFree(kv.second.d_tabs);
    for (auto &kv : cf_plans)
        if (kv.second.dev) (void)hipFree(kv.second.dev);
    rot_keys[0] = relin_key;
    for (auto &kv : rot_keys)
        if (!kv.second.borrowed)
            for (void *p : {(void *)kv.second.d, (void *)kv.second.d_cell, (void *)kv.second.d_gal})
                if (p) (void)hipFree(p);
    if (keys_borrowed) d_rotpack = nullptr, d_sk = nullptr, d_pk = nullptr;
    for (void *p : {(void *)d_mod, (void *)d_tw, (void *)d_tw_sh, (void *)d_itw, (void *)d_itw_sh, (void *)d_twp, (void *)d_itwp, (void *)d_twf, (void *)d_itwf, (void *)d_twd, (void *)d_itwd,
                    (void *)d_rotptrs, (void *)d_rotpack, (void *)d_giant_keys, (void *)d_giant_gal, (void *)d_giant_ginv,
                    (void *)d_rotgalois, (void *)d_rotginv, (void *)d_sk, (void *)d_pk, (void *)d_db, (void *)d_rot_group, (void *)d_ksi})
        if (p) (void)hipFree(p);
    for (auto e : lane_ev) (void)hipEventDestroy(e);
    for (auto e : par_ev)
        if (e) (void)hipEventDestroy(e);
    for (size_t k = 1; k < lane_stream.size(); k++) (void)hipStreamDestroy(lane_stream[k]);
    if (stream) (void)hipStreamDestroy(stream);
}
void Context::set_lane(int k) {
    stream = lane_stream[k];
    pool.cur = k;
}
void Context::sync_all() {
    for (auto st : lane_stream) HIP_CHECK(hipStreamSynchronize(st));
}

LimbSel Context::sel_q(int nl) const { return sel_range(0, nl); }
LimbSel Context::sel_range(int lo, int hi) const {
    LimbSel s{};
    s.n = hi - lo;
    for (int j = lo; j < hi; j++) s.mod[j - lo] = j;
    return s;
}
LimbSel Context::sel_ext(int nl) const {
    LimbSel s{};
    s.n = nl + nP;
    for (int j = 0; j < nl; j++) s.mod[j] = j;
    for (int k = 0; k < nP; k++) s.mod[nl + k] = nQ + k;
    return s;
}
ScaleSel Context::scale_ninv(const LimbSel &s) const {
    ScaleSel r{};
    for (int i = 0; i < s.n; i++) {
        r.s[i] = mod[s.mod[i]].ninv;
        r.s_sh[i] = mod[s.mod[i]].ninv_sh;
    }
    return r;
}
ScaleSel Context::scale_of(const LimbSel &s, const std::vector<u64> &v, bool times_ninv) const {
    ScaleSel r{};
    for (int i = 0; i < s.n; i++) {
        u64 qq = q[s.mod[i]];
        u64 val = v[i] % qq;
        if (times_ninv) val = mulmod_u64(val, mod[s.mod[i]].ninv, qq);
        r.s[i] = val;
        r.s_sh[i] = shoup(val, qq);
    }
    return r;
}
u64 Context::galois_elt(int rot) const {
    u64 M = 2ull * N, g = 1;
    int r = ((rot % slots) + slots) % slots;
    for (int i = 0; i < r; i++) g = (g * 5) % M;
    return g;
}

// ------------------------------------------------------------------ evaluation keys
void Context::adopt_keys(Context &src) {
    if (&src == this) return;
    if (src.device != device) throw std::runtime_error("hydia: keys can only be shared between contexts on the same GPU");
    if (src.nT != nT || src.N != N || src.prm.dnum != prm.dnum || src.prm.dim != prm.dim || src.q != q)
        throw std::runtime_error("hydia: keys can only be shared between contexts with identical parameters");
    if (relin_key.d || !rot_keys.empty() || d_sk || d_pk) throw StateError("hydia: this context already holds keys of its own");
    src.build_rotptrs();  // packed shadow of rotations 1..dim-1 (throws StateError when a rotation key is missing)
    src.sync_all();
    relin_key = src.relin_key;
    relin_key.borrowed = true;
    for (auto &kv : src.rot_keys) {
        EvalKey k = kv.second;
        k.borrowed = true;
        rot_keys[kv.first] = k;
    }
    d_sk = src.d_sk;
    d_pk = src.d_pk;
    d_rotpack = src.d_rotpack;
    rotptrs_packed = src.rotptrs_packed;
    rotptrs_premul = src.rotptrs_premul;
    keys_borrowed = true;
    HIP_CHECK(hipMemcpyAsync((void *)d_rotptrs, (const void *)src.d_rotptrs, sizeof(u64 *) * (size_t)prm.dim, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_rotgalois, src.d_rotgalois, sizeof(unsigned) * (size_t)prm.dim, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_rotginv, src.d_rotginv, sizeof(unsigned) * (size_t)prm.dim, hipMemcpyDeviceToDevice, stream));
    sync();
    rotptrs_valid = true;
}
u64 *Context::eval_key_storage(int rot) {
    if (keys_borrowed) throw StateError("hydia: this context borrows its keys from another context (re-key the owner)");
    EvalKey &k = rot == 0 ? relin_key : rot_keys[rot];
    if (!k.d) {
        const size_t bytes = (size_t)prm.dnum * 2 * nT * N * sizeof(u64);
        HIP_CHECK(hipMalloc((void **)&k.d, bytes));
        HIP_CHECK(hipMalloc((void **)&k.d_cell, sizeof(u64 *)));
        HIP_CHECK(hipMalloc((void **)&k.d_gal, 2 * sizeof(unsigned)));
        const u64 *self = k.d;
        unsigned g[2] = {1u, 1u};
        if (rot != 0) {
            g[0] = (unsigned)galois_elt(rot);
            u64 x = 1;  // inverse of an odd number modulo the power of two 2N (Newton)
            for (int i = 0; i < 6; i++) x = x * (2 - (u64)g[0] * x);
            g[1] = (unsigned)(x & (2ull * N - 1));
        }
        HIP_CHECK(hipMemcpy((void *)k.d_cell, &self, sizeof(u64 *), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(k.d_gal, g, 2 * sizeof(unsigned), hipMemcpyHostToDevice));
    }
    // every caller is about to WRITE this key (import, key generation, random fill): loop A's pointer table and its packed
    // shadow of rotations 1..dim-1 are rebuilt from the new contents before the next query
    if (rot >= 1 && rot < prm.dim) rotptrs_valid = giants_valid = false;
    return k.d;
}
void Context::load_eval_key(int rot, const u64 *host) {
    u64 *dst = eval_key_storage(rot);
    sync();
    HIP_CHECK(hipMemcpy(dst, host, (size_t)prm.dnum * 2 * nT * N * sizeof(u64), hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------ resident database
// the layout loop B wants for `cts` ciphertexts in blocks of `form`: group-sequential for more than 8 blocks, and then with 46-bit
// residues when every packed limb's modulus is below 2^46 (HYDIA_DB_48BIT keeps 6-byte residues)
DbLayout Context::db_layout_for(size_t cts, int form) const {
    const DbLayout ctm = hk::db_layout(N, nQ, db_packed ? 1 : 0);
    if (!(db_seq_ok && db_packed && form >= 2 && cts % (size_t)form == 0)) return ctm;
    bool b46 = db_bits46_ok;
    for (int j = 1; j < nQ; j++)
        if (q[j] >> 46) b46 = false;
    return hk::db_layout_seq(N, nQ, 1, form, (int)(cts / (size_t)form), tensor_bpp, tensor_nw, b46);
}
void Context::db_resize(size_t n_vectors, size_t cts, int form) {
    const DbLayout want = db_layout_for(cts, form);
    const size_t bytes = db_alloc_size(want, cts);  // (+ the tail a lane's last 16-byte load may touch: db_layout.h)
    if (d_db) sync_all();  // the layout may change under a still asynchronous query
    if (d_db && db_alloc_bytes != bytes) {
        HIP_CHECK(hipFree(d_db));
        d_db = nullptr;
    }
    if (!d_db && cts) {
        HIP_CHECK(hipMalloc((void **)&d_db, bytes));
        db_alloc_bytes = bytes;
    }
    db_cts = cts;
    db_vectors = n_vectors;
    // loop B walks "blocks" of `form` ciphertexts (the hoisted rotations per query: vector_dim, or the baby count of a pre-rotated
    // database, whose blocks are the (database block, giant step) pairs)
    db_lay = want;
}
// An imported database whose form is declared after the fact (hydia_db_set_babies): same ciphertexts, the order loop B wants for
// that form.  Needs room for a second copy while it runs — the databases the split is meant for (a few dozen blocks) have it; when
// the second buffer does not fit (a 148 GiB database on a 288 GB GPU) the call fails with a DeviceError BEFORE anything is touched:
// the resident database, its layout and its declared form stay what they were.
void Context::db_relayout(int form) {
    if (!d_db || db_cts == 0) return;
    const DbLayout want = db_layout_for(db_cts, form);
    if (want.seq == db_lay.seq && want.bd == db_lay.bd && want.seq_bpp == db_lay.seq_bpp && want.bits46 == db_lay.bits46) return;
    sync_all();
    struct Scratch {  // whatever throws below, neither buffer leaks and the resident database is still the old one
        Context *c;
        unsigned char *fresh = nullptr;
        u64 *plain = nullptr;
        ~Scratch() {
            if (plain) c->pool.put(plain);
            if (fresh) (void)hipFree(fresh);
        }
    } s{this};
    const size_t bytes = db_alloc_size(want, db_cts);
    hipError_t e = hipMalloc((void **)&s.fresh, bytes);
    if (e != hipSuccess) {
        pool.trim();  // cached evaluator temporaries may be what stands in the way
        (void)hipGetLastError();
        e = hipMalloc((void **)&s.fresh, bytes);
    }
    if (e != hipSuccess) {
        s.fresh = nullptr;
        (void)hipGetLastError();
        throw DeviceError("hydia: re-ordering the resident database for another mat-vec form needs a second buffer of " +
                          std::to_string(bytes >> 20) + " MiB, which does not fit in device memory (" + hipGetErrorString(e) +
                          "); the database is unchanged — enrol or load it with the form set beforehand (hydia_set_matvec)");
    }
    const size_t chunk = 16;
    s.plain = pool.get(chunk * 2 * nQ * N * sizeof(u64));
    for (size_t t0 = 0; t0 < db_cts; t0 += chunk) {
        const int cnt = (int)std::min(chunk, db_cts - t0);
        hk::db_unpack(stream, N, nQ, s.plain, d_db, t0, cnt, db_lay);
        hk::db_pack(stream, N, nQ, s.plain, s.fresh, t0, cnt, want);
    }
    sync();
    std::swap(d_db, s.fresh);  // the old buffer leaves with the scratch object
    db_alloc_bytes = bytes;
    db_lay = want;
}
namespace {
struct DbFileHeader {
    char magic[8];  // "HYDIADB1"
    uint32_t logN, nQ, dim, packed, kind, babies;  // babies: kind 5 / 6 — hoisted rotations the diagonals are laid out for (0 in old files = dim)
    uint64_t n_vectors, n_cts, ct_bytes;
    uint64_t moduli[HY_MAX_MODS];
};
const size_t DB_IO_CHUNK = (size_t)64 << 20;
struct FileCloser {
    FILE *f;
    ~FileCloser() {
        if (f) fclose(f);
    }
};
struct PinnedBuf {
    void *p = nullptr;
    explicit PinnedBuf(size_t n) { HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault)); }
    ~PinnedBuf() { (void)hipHostFree(p); }
};
// staging of the conversion between a group-sequential resident database and the ciphertext-major file: plain residues and a
// ciphertext-major image of DB_CONV_CTS ciphertexts
const size_t DB_CONV_CTS = 16;
struct DbConv {
    Context *cx;
    u64 *plain = nullptr;
    unsigned char *image = nullptr;
    DbConv(Context *c, const DbLayout &ctm) : cx(c) {
        plain = cx->pool.get(DB_CONV_CTS * 2 * cx->nQ * cx->N * sizeof(u64));
        image = reinterpret_cast<unsigned char *>(cx->pool.get(DB_CONV_CTS * ctm.ct_bytes));
    }
    ~DbConv() {
        cx->pool.put(plain);
        cx->pool.put(reinterpret_cast<u64 *>(image));
    }
};
}  // namespace
void Context::db_save(const char *path) {
    if (!d_db || db_cts == 0) throw StateError("hydia: no database resident");
    sync_all();
    FileCloser fc{fopen(path, "wb")};
    if (!fc.f) throw std::runtime_error(std::string("hydia: cannot open ") + path + " for writing");
    DbFileHeader h{};
    memcpy(h.magic, "HYDIADB1", 8);
    h.logN = (uint32_t)prm.logN; h.nQ = (uint32_t)nQ; h.dim = (uint32_t)prm.dim; h.packed = db_packed ? 1 : 0; h.kind = (uint32_t)db_kind;
    h.babies = (uint32_t)db_babies;
    const DbLayout ctm = hk::db_layout(N, nQ, db_packed ? 1 : 0);  // the file is ALWAYS ciphertext-major
    h.n_vectors = db_vectors; h.n_cts = db_cts; h.ct_bytes = ctm.ct_bytes;
    for (int j = 0; j < nQ; j++) h.moduli[j] = q[j];
    if (fwrite(&h, sizeof h, 1, fc.f) != 1) throw std::runtime_error("hydia: write failed (header)");
    const size_t total = db_cts * ctm.ct_bytes;
    if (!db_lay.seq) {
        PinnedBuf buf(std::min(total, DB_IO_CHUNK));
        for (size_t off = 0; off < total; off += DB_IO_CHUNK) {
            const size_t n = std::min(DB_IO_CHUNK, total - off);
            HIP_CHECK(hipMemcpy(buf.p, d_db + off, n, hipMemcpyDeviceToHost));
            if (fwrite(buf.p, 1, n, fc.f) != n) throw std::runtime_error("hydia: write failed (disk full?)");
        }
        return;
    }
    // group-sequential resident layout: DB_CONV_CTS ciphertexts at a time through plain residues into a ciphertext-major staging image
    DbConv cv(this, ctm);
    PinnedBuf buf(DB_CONV_CTS * ctm.ct_bytes);
    for (size_t t0 = 0; t0 < db_cts; t0 += DB_CONV_CTS) {
        const size_t cnt = std::min(DB_CONV_CTS, db_cts - t0);
        db_fetch(t0, cv.plain, (int)cnt);
        hk::db_pack(stream, N, nQ, cv.plain, cv.image, 0, (int)cnt, ctm);
        HIP_CHECK(hipMemcpyAsync(buf.p, cv.image, cnt * ctm.ct_bytes, hipMemcpyDeviceToHost, stream));
        sync();
        if (fwrite(buf.p, 1, cnt * ctm.ct_bytes, fc.f) != cnt * ctm.ct_bytes) throw std::runtime_error("hydia: write failed (disk full?)");
    }
}
void Context::db_load(const char *path) {
    FileCloser fc{fopen(path, "rb")};
    if (!fc.f) throw std::runtime_error(std::string("hydia: cannot open ") + path);
    DbFileHeader h{};
    if (fread(&h, sizeof h, 1, fc.f) != 1 || memcmp(h.magic, "HYDIADB1", 8) != 0) throw std::runtime_error("hydia: not a hydia database file");
    if ((int)h.logN != prm.logN || (int)h.nQ != nQ || (int)h.dim != prm.dim) throw std::runtime_error("hydia: database file was written for other parameters");
    for (int j = 0; j < nQ; j++)
        if (h.moduli[j] != q[j]) throw std::runtime_error("hydia: database file was written on another prime chain");
    const DbLayout ctm = hk::db_layout(N, nQ, db_packed ? 1 : 0);
    if ((h.packed != 0) != db_packed || h.ct_bytes != ctm.ct_bytes)
        throw std::runtime_error("hydia: database file layout (48-bit packed / 8-byte) differs from this context's");
    if (h.kind != 4 && h.kind != 5 && h.kind != 6) throw std::runtime_error("hydia: database file has an unknown packing kind");
    if (h.kind == 6 && (h.babies < 2 || h.babies >= (uint32_t)prm.dim || (h.babies & (h.babies - 1)) || prm.dim % h.babies))
        throw std::runtime_error("hydia: database file carries an invalid baby count");
    if (h.kind == 5 && h.babies != 0 && h.babies != (uint32_t)prm.dim) throw std::runtime_error("hydia: database file header is inconsistent (kind 5 with a baby count)");
    // the header is untrusted input: the ciphertext count must be the one the packing implies for n_vectors (diagonal packing:
    // ceil(ceil(n / dim) / (slots / dim)) * dim; column packing: ceil(n / slots) * dim), and the file must hold exactly that many
    // — checked BEFORE anything is allocated or the resident database is touched
    if (h.n_vectors < 1 || h.n_cts < 1) throw std::runtime_error("hydia: database file header declares an empty database");
    const uint64_t dim = (uint64_t)prm.dim, S = (uint64_t)slots;
    const uint64_t G = (h.n_vectors + S - 1) / S;
    const uint64_t want_cts = h.kind != 4 ? (((h.n_vectors + dim - 1) / dim + S / dim - 1) / (S / dim)) * dim : G * dim;
    if (h.n_vectors > (UINT64_MAX - S) || h.n_cts != want_cts)
        throw std::runtime_error("hydia: database file header is inconsistent (ciphertext count does not match the vector count)");
    {
        struct stat sb {};
        if (fstat(fileno(fc.f), &sb) != 0) throw std::runtime_error("hydia: cannot stat the database file");
        const unsigned __int128 need = (unsigned __int128)h.n_cts * h.ct_bytes + sizeof h;
        if ((unsigned __int128)sb.st_size < need) throw std::runtime_error("hydia: database file is truncated");
        if ((unsigned __int128)sb.st_size > need) throw std::runtime_error("hydia: database file has trailing bytes");
    }
    sync_all();  // an earlier, still asynchronous query may be reading the resident database this load overwrites
    db_kind = 0;
    db_resize(h.n_vectors, h.n_cts, h.kind == 4 ? -1 : (h.kind == 6 ? (int)h.babies : prm.dim));
    const size_t total = db_cts * ctm.ct_bytes;
    if (!db_lay.seq) {
        PinnedBuf buf(std::min(total, DB_IO_CHUNK));
        for (size_t off = 0; off < total; off += DB_IO_CHUNK) {
            const size_t n = std::min(DB_IO_CHUNK, total - off);
            if (fread(buf.p, 1, n, fc.f) != n) {
                db_vectors = db_cts = 0;  // nothing usable is resident
                throw std::runtime_error("hydia: database file is truncated");
            }
            HIP_CHECK(hipMemcpy(d_db + off, buf.p, n, hipMemcpyHostToDevice));
        }
    } else {
        DbConv cv(this, ctm);
        PinnedBuf buf(DB_CONV_CTS * ctm.ct_bytes);
        for (size_t t0 = 0; t0 < db_cts; t0 += DB_CONV_CTS) {
            const size_t cnt = std::min(DB_CONV_CTS, db_cts - t0);
            if (fread(buf.p, 1, cnt * ctm.ct_bytes, fc.f) != cnt * ctm.ct_bytes) {
                db_vectors = db_cts = 0;
                throw std::runtime_error("hydia: database file is truncated");
            }
            HIP_CHECK(hipMemcpyAsync(cv.image, buf.p, cnt * ctm.ct_bytes, hipMemcpyHostToDevice, stream));
            hk::db_unpack(stream, N, nQ, cv.plain, cv.image, 0, (int)cnt, ctm);
            db_store(t0, cv.plain, (int)cnt);
            sync();  // the pinned buffer is refilled next
        }
    }
    db_kind = (int)h.kind;
    db_babies = h.kind == 4 ? 0 : (h.babies ? (int)h.babies : prm.dim);
}
void Context::db_store(size_t t0, const u64 *d_plain, int X) { hk::db_pack(stream, N, nQ, d_plain, d_db, t0, X, db_lay); }
void Context::db_fetch(size_t t0, u64 *d_plain, int X) { hk::db_unpack(stream, N, nQ, d_plain, d_db, t0, X, db_lay); }

// ------------------------------------------------------------------ kernel timers
void Context::timer_begin(const char *name) {
    if (!timing) return;
    if (timers[name].pending.size() >= 2048) timer_collect();  // bound the event backlog of a caller that never reads the timers
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, stream));
    timers[name].pending.emplace_back(a, b);
}
void Context::timer_end(const char *name) {
    if (!timing) return;
    auto &t = timers[name];
    HIP_CHECK(hipEventRecord(t.pending.back().second, stream));
}
void Context::timer_collect() {
    for (auto &kv : timers) {
        for (auto &ev : kv.second.pending) {
            HIP_CHECK(hipEventSynchronize(ev.second));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
            kv.second.total_ms += ms;
            kv.second.launches++;
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        kv.second.pending.clear();
    }
}

}  // namespace hydia

// image_matching_amd/csrc/hydia_core.h — host side of libhydia: RNS context, device tables, HBM pool, batched
// ciphertext evaluator and the HyDia sender/receiver/enroller engines behind the C-ABI of include/hydia.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <exception>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"

namespace hydia {
// error classes of the C-ABI (include/hydia.h hydia_status): the boundary classifies by TYPE, never by message text
struct DeviceError : std::runtime_error {  // a HIP call failed -> HYDIA_ERR_DEVICE
    using std::runtime_error::runtime_error;
};
struct StateError : std::runtime_error {   // missing key / database / wrong level -> HYDIA_ERR_STATE
    using std::runtime_error::runtime_error;
};
}  // namespace hydia

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            throw hydia::DeviceError(std::string(#expr) + " failed: " + hipGetErrorString(_e));           \
    } while (0)

namespace hydia {

struct Params {
    int logN = 15, mult_depth = 11, scale_bits = 45, first_bits = 60, dnum = 3, dim = 512;
    // caller-supplied prime chain (hydia_ctx_create_custom): nQ ciphertext primes (q_0 first) then nP special primes;
    // empty = derive.  custom_psi (optional) = the 2N-th roots of unity to use.
    std::vector<u64> custom_q, custom_psi;
    int custom_nP = 0;
};

// Caching HBM allocator: every evaluator temporary comes from here.  Work is enqueued on one stream per LANE; a block is
// cached on the free list of the lane that allocated it and only handed out again on that lane, so reuse is ordered by the
// lane's stream without synchronisation.  `cur` selects the lane new requests are served from (Context::set_lane).
class Pool {
  public:
    ~Pool();
    u64 *get(size_t bytes);
    void put(u64 *p);
    void trim();
    size_t bytes_live = 0, bytes_cached = 0, peak = 0;
    int cur = 0;

  private:
    // handles may be freed from another thread than the one inside a library call (a garbage collector's finaliser while ctypes has
    // released the GIL): the free lists are guarded; `cur` is only changed by the thread that owns the context
    std::recursive_mutex mu_;
    std::map<int, std::multimap<size_t, u64 *>> free_;    // per lane
    std::map<u64 *, std::pair<size_t, int>> size_;         // block -> (bytes, owning lane)
};

struct Context;

// A batch of X ciphertexts of identical shape, [X][npoly][lstride][N] with the first nl limbs of each polynomial in
// use, evaluation form, resident in HBM.  lstride > nl after a level drop: dropping limbs is O(1) (a view keeps the
// allocation), every kernel that consumes ciphertexts takes the limb stride.
struct Ct {
    Context *ctx = nullptr;
    u64 *d = nullptr;
    int X = 0, npoly = 0, nl = 0, lstride = 0;
    double scale = 0;
    bool view = false;  // does not own d
    Ct() = default;
    Ct(Context *c, int X_, int npoly_, int nl_, double scale_);
    Ct(const Ct &) = delete;
    Ct &operator=(const Ct &) = delete;
    Ct(Ct &&o) noexcept { *this = std::move(o); }
    Ct &operator=(Ct &&o) noexcept;
    ~Ct();
    size_t poly_elems() const;                                    // lstride * N
    size_t ct_elems() const { return (size_t)npoly * poly_elems(); }
    size_t bytes() const { return ct_elems() * X * sizeof(u64); }  // allocated bytes
    bool compact() const { return lstride == nl; }
    Ct alias(int nl_) const;  // non-owning view with nl_ <= nl limbs in use
};

struct KernelTimer {
    double total_ms = 0;
    long launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// Host-only part: parameter derivation and conversion constants (no GPU needed; `hydia_params_describe` uses it).
struct HostParams {
    Params prm;
    int N, slots, nQ, nP, nT, alpha;
    double delta;
    std::vector<u64> q, psi;
    std::vector<ModC> mod;
    std::vector<u64> P_mod_q, Pinv_mod_q, Phat_inv;
    std::vector<std::vector<u64>> Phat_mod_q;  // [k][j]
    std::vector<std::vector<u64>> ql_inv;      // [l][j]
    explicit HostParams(const Params &p);
    void twiddles(int m, std::vector<u64> &tw, std::vector<u64> &tws, std::vector<u64> &itw, std::vector<u64> &itws) const;
};

struct Context : HostParams {
    int device;
    hipStream_t stream = nullptr;  // the CURRENT lane's stream (lane 0 unless inside a multi-lane section)
    std::vector<hipStream_t> lane_stream;  // lane 0 = the main stream
    int nlanes = 2;                        // comparator lanes (HYDIA_LANES)
    std::vector<hipEvent_t> lane_ev;
    void set_lane(int k);
    void sync_all();
    // Two INDEPENDENT pieces of evaluator work.  On the main lane of a single-lane section (one block: everything is
    // latency-bound, a launch fills a fraction of the GPU) `side` is enqueued on lane 1 and runs concurrently with `here`;
    // otherwise they simply run one after the other.  Same arithmetic per ciphertext either way.  Not nestable.
    template <class F0, class F1>
    void par2(F0 &&here, F1 &&side);
    bool side_lane_free = true;   // false inside multi-lane sections and inside par2
    hipEvent_t par_ev[2] = {nullptr, nullptr};
    Pool pool;

    // device tables
    ModC *d_mod = nullptr;
    u64 *d_tw = nullptr, *d_tw_sh = nullptr, *d_itw = nullptr, *d_itw_sh = nullptr, *d_twp = nullptr, *d_itwp = nullptr, *d_twf = nullptr, *d_itwf = nullptr, *d_twd = nullptr, *d_itwd = nullptr;
    NttTables tabs{};

    // evaluation keys resident in HBM: [dnum][2][nT][N]; each carries a one-element device cell holding its own
    // address and Galois element so single-key launches need no per-call upload
    struct EvalKey {
        u64 *d = nullptr;
        const u64 **d_cell = nullptr;
        unsigned *d_gal = nullptr;   // [0] = Galois element g, [1] = g^{-1} mod 2N
        bool borrowed = false;       // storage belongs to another context on the same GPU (adopt_keys)
    };
    EvalKey relin_key;
    std::map<int, EvalKey> rot_keys;
    const u64 **d_rotptrs = nullptr;   // device array: key pointers of rotations 1..dim-1 (hoisted loop A)
    unsigned *d_rotgalois = nullptr;   // device array: their Galois elements
    unsigned *d_rotginv = nullptr;     // device array: inverse Galois elements (scatter form of the automorphism)
    bool rotptrs_valid = false, rotptrs_packed = false, rotptrs_premul = false;
    unsigned char *d_rotpack = nullptr;  // packed shadow of rotation keys 1..dim-1 (45/46-bit limbs as 6-byte residues), loop A only
    bool rot_packed = true;              // HYDIA_KEYS_UNPACKED turns the shadow off
    // Contexts on the SAME GPU (shards of one database that share a device) can use one resident copy of the keys: this
    // context borrows every key buffer of `src` (relin, rotations, packed shadow, pk, sk); `src` must outlive it and must
    // not be re-keyed while borrowers exist.
    void adopt_keys(Context &src);
    bool keys_borrowed = false;
    void load_eval_key(int rot /* 0 = relinearisation */, const u64 *host);
    u64 *eval_key_storage(int rot);    // allocates (or returns) the HBM buffer of key `rot`
    u64 *d_sk = nullptr;               // [nT][N] (receiver side only)
    u64 *d_pk = nullptr;               // [2][nQ][N]
    unsigned *d_rot_group = nullptr;   // canonical embedding: 5^j mod 2N
    double *d_ksi = nullptr;           // (cos, sin)(2 pi k / 2N), k <= 2N

    // encrypted database resident in HBM: G*dim ciphertexts in the DbLayout of kernels.h (45/46-bit limbs stored as
    // 48-bit residues when every scaling prime is below 2^48; HYDIA_DB_UNPACKED keeps plain [2][nQ][N] u64)
    unsigned char *d_db = nullptr;
    size_t db_vectors = 0, db_cts = 0;
    int db_kind = 0;   // 0 none, 5 diagonal packing (HyDia, approach 5), 6 the same with pre-rotated diagonals (BSGS mat-vec), 4 column packing (HERS)
    int db_babies = 0; // kind 5 / 6: hoisted (baby) rotations the resident database was enrolled for; == vector_dim for kind 5
    // Which form of the diagonal mat-vec a database enrolled on this context gets (HYDIA_MATVEC=auto|hoisted|bsgs|<B>, hydia_set_matvec).
    // With i = b + B g: B - 1 hoisted rotations of the query per QUERY, dim / B relinearisations + dim / B - 1 giant rotations per BLOCK.
    //   hoisted  B = dim: the reference's own form (dim - 1 hoisted rotations, one relinearisation per block, no giant step)
    //   bsgs     B = the smallest power of two with B^2 >= dim (32 at dim 512): the classic baby-step / giant-step split
    //   <B>      any power of two dividing dim
    //   auto     B grows with the blocks this context holds (auto_babies): few blocks -> few babies, many blocks -> hoisted
    int matvec_mode = 0;  // 0 auto, 1 hoisted, otherwise the baby count
    int bsgs_babies() const {
        int B = 1;
        while (B * B < prm.dim) B <<= 1;
        return B;
    }
    // measured on MI355X (profiles/r04/matvec_sweep.txt: every split at 1 .. 64 blocks, after the giant steps moved to the fused key
    // switch): 64 babies win up to 3 blocks, 128 up to 12, 256 up to 40 (24 in round 3), all hoisted above
    int auto_babies(size_t blocks) const {
        const int base = bsgs_babies();
        static const struct { size_t limit; int mult; } rule[] = {{3, 2}, {12, 4}, {40, 8}};  // 64 / 128 / 256 babies at dim 512
        for (const auto &r : rule)
            if (blocks <= r.limit) return std::min(prm.dim, base * r.mult);
        return prm.dim;
    }
    // babies of an enrolment of `blocks` blocks; explicit > 0 overrides the context's policy (1 = hoisted)
    int babies_for(size_t blocks, int explicit_mode = 0) const {
        const int m = explicit_mode ? explicit_mode : matvec_mode;
        if (m == 0) return auto_babies(blocks);
        if (m == 1) return prm.dim;
        if (m < 2 || m > prm.dim || (m & (m - 1)) || prm.dim % m) throw std::runtime_error("hydia: the baby count must be a power of two dividing vector_dim");
        return m;
    }
    const u64 **d_giant_keys = nullptr;  // device arrays over g = 0 .. dim/B - 1: key, Galois element and inverse of rotation B g
    unsigned *d_giant_gal = nullptr, *d_giant_ginv = nullptr;
    bool giants_valid = false;
    int giants_B = 0, giants_G = 0;  // the tables hold one entry per (giant step g >= 1, database block): x = (g - 1) * G + block
    void build_giants(int G);
    bool db_packed = true;
    bool db_seq_ok = true;  // many-block hoisted databases take the group-sequential layout (HYDIA_DB_CT_MAJOR turns it off)
    bool db_bits46_ok = true;  // ... with 46-bit residues for the packed limbs (HYDIA_DB_48BIT turns that off)
    size_t db_alloc_bytes = 0;
    DbLayout db_layout_for(size_t cts, int form) const;
    DbLayout db_lay{};      // layout of the resident database (set by db_resize)
    DbLayout db_layout() const { return d_db ? db_lay : hk::db_layout(N, nQ, db_packed ? 1 : 0); }
    // (re)allocates the resident database for `cts` ciphertexts.  form = the hoisted-rotation count its diagonals will be laid out
    // for (vector_dim = the reference's form, or the baby count): loop B walks blocks of `form` ciphertexts, and more than 8 of them
    // take the group-sequential layout; -1 (HERS' column packing) stays ciphertext-major
    void db_resize(size_t n_vectors, size_t cts, int form);
    void db_relayout(int form);  // the same ciphertexts laid out for another form (a second buffer for the duration)
    // persistence of the resident database (own streaming format: header + the ciphertext-major layout verbatim, so a restart does
    // not re-enrol from plaintext; the reference keeps serial/db_diagonal/index<t>.bin, enroller_diag.cpp:158-166)
    void db_save(const char *path);
    void db_load(const char *path);
    void db_store(size_t t0, const u64 *d_plain, int X);  // [X][2][nQ][N] device residues -> ciphertexts t0..t0+X-1
    void db_fetch(size_t t0, u64 *d_plain, int X);

    std::map<std::string, KernelTimer> timers;
    bool timing = true;

    explicit Context(const Params &p, int device);
    ~Context();
    void sync() { HIP_CHECK(hipStreamSynchronize(stream)); }

    // ---- helpers
    // ModUp constants of one level: per-digit conversion tables in device memory + the (D/q_j)^{-1} factors of the inverse NTT
    struct ModUpPlan {
        ConvTab *d_tabs = nullptr;
        std::vector<u64> inv;
        int nd = 0;
    };
    std::map<int, ModUpPlan> modup_plans;
    const ModUpPlan &modup_plan(int nl);
    // column-fused conversions (colfuse.hip): the (sources -> targets) maps of a ModUp (one per digit), a ModDown, loop A's ModDown
    // and the merged ModDown + Rescale at one level, in host and device memory
    struct CfPlan {
        std::vector<ColFuse> host;
        ColFuse *dev = nullptr;
    };
    std::map<std::string, CfPlan> cf_plans;
    const CfPlan &cf_plan_modup(int nl);
    const CfPlan &cf_plan_moddown(int nl, bool premul);          // premul: constants carry P^{-1} (loop A's pre-scaled key shadow)
    const CfPlan &cf_plan_moddown_rescale(int nl, bool dbl);
    const CfPlan &cf_plan_rescale(int nl);  // Rescale alone as a column-fused map without conversion sources: u -> every remaining limb     // merged ModDown + Rescale from level nl
    const CfPlan &cf_plan_store(const std::string &key, std::vector<ColFuse> &&maps);
    bool colfuse = true;        // HYDIA_NO_COLFUSE: pass 1' / conversion / pass 1 as three kernels
    // (five special primes, all below 2^48, take the five-source instantiation of the narrow column-fused kernel: the secondary
    // "special primes on the FP64 pipe" configuration of tools/exp_fp64_special_primes.py; its ModDown conversions never take the small-launch form)
    bool cf_ok() const {
        if (!(colfuse && prm.logN == 15 && alpha <= HY_CF_SRC && nP <= HY_CF_SRC_MAX)) return false;
        if (nP > HY_CF_SRC)
            for (int k = nQ; k < nT; k++)
                if (q[k] >> 48) return false;
        return true;
    }
    bool cf_small_moddown(int XP) const { return nP <= HY_CF_SRC && hk::ntt15_colfuse_small(XP, 1); }
    LimbSel sel_q(int nl) const;           // limbs 0..nl-1
    LimbSel sel_ext(int nl) const;         // limbs 0..nl-1 then all P limbs
    LimbSel sel_range(int lo, int hi) const;
    ScaleSel scale_ninv(const LimbSel &s) const;                   // N^{-1}
    ScaleSel scale_of(const LimbSel &s, const std::vector<u64> &v, bool times_ninv) const;
    u64 galois_elt(int rot) const;

    // ---- primitive evaluator ops (all asynchronous on `stream`)
    void ntt_fwd(u64 *base, size_t outer, int X, const LimbSel &s);
    void ntt_inv(const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &s, const ScaleSel &sc);
    // ModUp: c [X][nl][N] at stride c_outer -> dig [X][nd][nE][N]
    // ps != nullptr (column-fused path only): the polynomial is d2 = a1 b1 of a fused product, formed in the load of the inverse
    // transform and also written to d2_out, compact [X][nl][N]
    void modup_digits(const u64 *c, size_t c_outer, int X, int nl, u64 *dig, bool copy_own = true, bool p1_only = false,
                      const ProdSrc *ps = nullptr, u64 *d2_out = nullptr);
    bool fuse_loop_a = true;    // loop A: Q-limb inner product inside the ModDown transform's epilogue (HYDIA_NO_FUSE_LOOPA)
    bool fork_products = true;  // comparator: independent products of a one-block query on two lanes (HYDIA_NO_FORK)
    bool getenv_int_arith = false;  // HYDIA_NTT_INT: integer arithmetic everywhere (also the coefficient-wise kernels' FP64 paths)
    bool fuse_ip = true;  // relinearisation: second NTT pass of ModUp fused with the inner product (HYDIA_NO_FUSE_IP)
    // inner product with X keys + ModDown (+ addend, + automorphism): out [X][2][nl][N]
    // keys_packed_nQ > 0: d_keys point at packed keys (hk::key_pack)
    void ks_apply(const u64 *dig, size_t dig_x_stride, int X, int nl, const u64 *const *d_keys, int same_key,
                  const u64 *addend, size_t add_x_stride, size_t add_poly_stride, int add_polys, const unsigned *d_galois,
                  const unsigned *d_ginv, int same_galois, bool dbl, u64 *out, int keys_packed_nQ = 0);
    void build_rotptrs();
    void relinearize(Ct &c, bool dbl = false);  // [X][3][nl] -> [X][2][nl]; dbl: result doubled (2ab of a Chebyshev step)
    // drop the last limb; optionally fused: result -= sub (a view at the new level), result += addc (constant, poly 0)
    void rescale(Ct &c, const Ct *sub = nullptr, const double *addc = nullptr);
    // RelinearizeInPlace followed by RescaleInPlace (sender_diag.cpp:79-80) as ONE pipeline with bit-identical results:
    // the dropped limb of the ModDown output is obtained in the coefficient domain, so ModDown's and Rescale's
    // corrections share a single forward NTT per remaining limb ((l+1) transforms per polynomial saved)
    void relin_rescale(Ct &c, bool dbl = false, const Ct *sub = nullptr, const double *addc = nullptr, bool sub_is_add = false);
    // hybrid key switch of one polynomial per ciphertext through the fused pipeline (round 4: what relin_rescale_into does without its
    // rescale half): ModUp with the digits' second pass inside the inner product, the special-prime sums straight into the inverse
    // transform, column-fused ModDown conversion, combine (+ addend, doubling, automorphism) in the last pass.  key: one key for every
    // ciphertext, or d_keys (device array) one per ciphertext.  false: not available here (generic rings, switches) — use ks_apply
    bool ks_fused_ok() const { return prm.logN == 15 && fuse_ip && !relin_separate_intt && cf_ok() && ks_fuse; }
    bool rescale_cf = true;  // HYDIA_NO_RESCALE_CF: Rescale's spread + first pass as k_ntt15_p1<false, 2> (round 3's form)
    bool ks_fuse = true;  // HYDIA_NO_KS_FUSE: relinearize / rotate / giant steps through modup_digits + ks_apply (round 3's form)
    void ks_fused(const u64 *c1, size_t c1_xs, int X, int nl, const u64 *key, const u64 *const *d_key_cell, const u64 *const *d_keys,
                  const u64 *addend, size_t add_x, size_t add_p, int add_polys, const unsigned *d_ginv, int same_g, bool dbl, u64 *out);
    // ps != nullptr: c carries shape and scale only (X, nl, scale; no data) — the degree-2 ciphertext is the product ps and is never formed
    void relin_rescale_into(const Ct &c, bool dbl, const Ct *sub, const double *addc, bool sub_is_add, u64 *out_d, const ProdSrc *ps = nullptr,
                            const ScaleSel *kap = nullptr);
    // (a b) relinearised (doubled) and rescaled (+- sub)(+ addc): EvalMultNoRelin + relin_rescale, with the tensor fused into its
    // consumers where the merged pipeline runs (prod_fusable)
    bool prod_fuse = true;  // HYDIA_NO_PROD_FUSE: k_tensor materialises every degree-2 ciphertext (round 3's form)
    bool prod_fusable(int nl) const;
    // csub != nullptr: the Chebyshev step 2ab - K csub (K = round(s_a s_b / s_c): mult_norelin_sub's arithmetic), dbl must be set
    Ct mult_relin_rescale(const Ct &a, const Ct &b, bool dbl = false, const Ct *sub = nullptr, const double *addc = nullptr, bool sub_is_add = false,
                          const Ct *csub = nullptr);
    bool prod_fuse_csub = true;  // HYDIA_NO_CSUB_FUSE: Chebyshev steps with a subtrahend keep k_tensor<true>
    bool merge_rescale = true;      // HYDIA_NO_MERGE_RESCALE: run the two steps separately (A/B)
    Ct clone(const Ct &a);          // compact copy
    void drop_to(Ct &a, int nl);    // O(1): keeps the allocation, lstride unchanged
    int tensor_bpp = 2;             // DB blocks per wave in loop B (HYDIA_TENSOR_BPP; 4 spills past 168 VGPRs)
    int tensor_nw = 4;              // max waves per workgroup in loop B (HYDIA_TENSOR_NW; 0 = up to 16)
    // round-2 fusions, each with its off switch for the parity variants (read once per context)
    bool modup_per_digit = false, loop_a_separate_ip = false, loop_a_int_ip = false, relin_separate_intt = false, loop_a_limb_fastest = false;
    void add_inplace(Ct &a, const Ct &b);
    void sub_inplace(Ct &a, const Ct &b);
    void add_const(Ct &a, double c);
    Ct mul_const(const Ct &a, double c, double const_scale);
    // sum_t coef[t] * terms[t] + c0 at common scale S (each constant encoded at S / scale(term)); all terms same X, npoly, nl
    Ct lincomb(const std::vector<const Ct *> &terms, const std::vector<double> &coef, double c0, double S);
    // K leaves over the same terms in one pass: result batch [K][X] (leaf k = ciphertexts k*X .. k*X+X-1), every leaf at limb count of the terms
    Ct lincomb_multi(const std::vector<const Ct *> &terms, const std::vector<std::vector<double>> &coef, const std::vector<double> &c0,
                     const std::vector<double> &S);
    std::vector<std::vector<u64>> lcm_stage;  // host staging of the constant tables: kept alive until the stream has consumed them
    Ct mult_norelin(const Ct &a, const Ct &b);
    Ct similarity_accumulate(const Ct &qc);
    Ct relin_compare_lanes(Ct &acc, double delta, int sign_depth);
    Ct mult_norelin_sub(const Ct &a, const Ct &b, const Ct &c);
    Ct mult(const Ct &a, const Ct &b);  // align, tensor, relin, rescale
    Ct rotate(const Ct &a, int rot);    // X = any; full key switch

    // ---- HyDia sender (src/sender/sender_diag.cpp)
    Ct rotate_query(const Ct &q);                   // -> [dim][2][nQ][N]
    // rotations first .. first+count-1 of the query (rotation 0 = the query itself) into out [count][2][nQ][N]: one rank's share of
    // loop A when the rotations are split over the GPUs of a node and all-gathered (SURVEY 8e option B)
    void rotate_query_range(const Ct &q, int first, int count, u64 *out);
    Ct similarity(const Ct &q);                     // -> [G][2][nQ-1][N]
    // the same scenarios on rotations supplied by the caller ([dim][2][nQ][N], as rotate_query returns them)
    Ct similarity_bsgs_sum(const Ct &q);            // BSGS mat-vec up to (not including) the rescale: [G][2][nQ][N]
    Ct similarity_accumulate_rot(const Ct &rot);
    Ct similarity_rot(const Ct &rot);
    Ct index_scenario_rot(const Ct &rot);
    Ct chebyshev_compare(const Ct &x, double delta, int sign_depth);
    Ct index_scenario(const Ct &q);
    Ct membership_scenario(const Ct &q);
    Ct sum_and_evalsum(const Ct &s);  // EvalAddMany over the batch + EvalSum over all slots
    Ct add_many(const Ct &s);         // EvalAddManyInPlace alone (the per-shard part of a sharded membership query)
    Ct eval_sum(const Ct &a);         // EvalSum alone
    void add_raw_inplace(Ct &a, const u64 *other /* compact, same shape, this device */);  // integer sum, no reduction
    void mod_reduce_inplace(Ct &a);   // every 64-bit value -> canonical residue of its limb
    // ---- HERS sender (approach 4, src/sender/sender_hers.cpp): q = dim query ciphertexts
    Ct hers_similarity(const Ct &q);
    Ct hers_index_scenario(const Ct &q);
    Ct hers_membership_scenario(const Ct &q);

    // timing of named kernels (HIP events on `stream`)
    void timer_begin(const char *name);
    void timer_end(const char *name);
    void timer_collect();
};

template <class F0, class F1>
void Context::par2(F0 &&here, F1 &&side) {
    if (!(side_lane_free && nlanes >= 2 && pool.cur == 0 && fork_products)) {
        side();
        here();
        return;
    }
    if (!par_ev[0]) {
        HIP_CHECK(hipEventCreateWithFlags(&par_ev[0], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&par_ev[1], hipEventDisableTiming));
    }
    side_lane_free = false;
    struct Guard {  // an exception inside either piece must not leave the context on the side lane
        Context *c;
        ~Guard() {
            c->set_lane(0);
            c->side_lane_free = true;
            if (std::uncaught_exceptions() > 0)
                for (auto st : c->lane_stream) (void)hipStreamSynchronize(st);
        }
    } guard{this};
    // blocks the side lane takes from its free list were released on the host before this point, i.e. after their last
    // consumers were enqueued: waiting for the main lane's position covers them (same discipline as relin_compare_lanes)
    HIP_CHECK(hipEventRecord(par_ev[0], stream));
    set_lane(1);
    HIP_CHECK(hipStreamWaitEvent(stream, par_ev[0], 0));
    side();
    HIP_CHECK(hipEventRecord(par_ev[1], stream));
    set_lane(0);
    here();
    HIP_CHECK(hipStreamWaitEvent(stream, par_ev[1], 0));
}

// host residue of a real constant (value already multiplied by its scale)
u64 double_to_mod(double v, u64 q);
u64 powmod_u64(u64 a, u64 e, u64 q);
u64 invmod_u64(u64 a, u64 q);
u64 mulmod_u64(u64 a, u64 b, u64 q);

}  // namespace hydia

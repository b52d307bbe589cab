// image_matching_amd/csrc/evaluator.cpp — batched CKKS evaluator and the HyDia sender engine on MI355X.
//
// Mirrors, as batched HBM-resident operations on one HIP stream, the OpenFHE calls of the reference's hot path:
//   DiagonalSender::computeSimilarity        /root/reference/src/sender/sender_diag.cpp:12-33   -> Context::similarity
//   DiagonalSender::computeSimilarityMatrix  /root/reference/src/sender/sender_diag.cpp:66-83   -> tensor kernel + relinearize + rescale
//   DiagonalSender::indexScenario / membershipScenario  sender_diag.cpp:35-63                    -> index_scenario / membership_scenario
//   OpenFHEWrapper::chebyshevCompare         /root/reference/src/openFHE_wrapper.cpp:143-185     -> chebyshev_compare
// Every operation works on a batch of X ciphertexts so the per-block tails (relinearise, rescale, comparator) of all
// DB blocks resident on this GPU run as ONE launch sequence with X-times larger grids.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "hydia_core.h"

#include <exception>

namespace hydia {

static u64 shoup_h(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

// Inherent ("algorithmic") bytes of an OPERATION as SURVEY 8d prices them — 2 N 8 B (one read + one write) per limb-transform, an
// evaluation key once per launch, operands of coefficient-wise work once — recorded in the byte ledger under "op:<name>" next to
// the per-kernel entries (which count what the launches really move: a two-pass transform twice that).  bench.py's roofline.step
// sums them over one indexScenario.  Costs nothing while the ledger is off.
static void op_bytes(const char *op, size_t N, double transforms, double other_bytes) {
    hk::ledger_add(op, transforms * 2.0 * (double)N * 8.0 + other_bytes);
}

void Context::ntt_fwd(u64 *base, size_t outer, int X, const LimbSel &s) {
    if (X <= 0 || s.n <= 0) return;
    hk::ntt_forward(stream, tabs, prm.logN, base, base, outer, outer, X, s);
}
void Context::ntt_inv(const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &s, const ScaleSel &sc) {
    if (X <= 0 || s.n <= 0) return;
    hk::ntt_inverse(stream, tabs, prm.logN, src, dst, so, dso, X, s, sc);
}

const Context::ModUpPlan &Context::modup_plan(int nl) {
    auto it = modup_plans.find(nl);
    if (it != modup_plans.end()) return it->second;
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    const LimbSel esel = sel_ext(nl);
    ModUpPlan pl;
    pl.nd = nd;
    pl.inv.resize(nl);
    std::vector<ConvTab> tabs(nd);
    for (int d = 0; d < nd; d++) {
        const int lo = d * alpha, hi = std::min(lo + alpha, nl), sz = hi - lo;
        ConvTab &tab = tabs[d];
        tab = ConvTab{};
        tab.ns = sz;
        tab.nt = nE;
        tab.skip_lo = lo;
        tab.skip_hi = hi;
        for (int s = 0; s < sz; s++) {
            const u64 qj = q[lo + s];
            u64 pr = 1;
            for (int i = lo; i < hi; i++)
                if (i != lo + s) pr = mulmod_u64(pr, q[i] % qj, qj);
            pl.inv[lo + s] = invmod_u64(pr, qj);
            for (int t = 0; t < nE; t++) {
                const u64 qt = q[esel.mod[t]];
                u64 f = 1;
                for (int i = lo; i < hi; i++)
                    if (i != lo + s) f = mulmod_u64(f, q[i] % qt, qt);
                tab.f[s][t] = f;
            }
        }
    }
    HIP_CHECK(hipMalloc((void **)&pl.d_tabs, sizeof(ConvTab) * nd));
    HIP_CHECK(hipMemcpy(pl.d_tabs, tabs.data(), sizeof(ConvTab) * nd, hipMemcpyHostToDevice));
    return modup_plans.emplace(nl, std::move(pl)).first->second;
}

// ---- column-fused conversion plans (colfuse.hip): the constants of the unfused kernels, regrouped per (sources -> targets) map
const Context::CfPlan &Context::cf_plan_store(const std::string &key, std::vector<ColFuse> &&maps) {
    CfPlan pl;
    pl.host = std::move(maps);
    for (ColFuse &cf : pl.host) {
        for (int t = 0; t < cf.nt; t++) {
            cf.t60[t] = (1ull << 60) % q[cf.tmod[t]];
            cf.tM[t] = mod[cf.tmod[t]];
        }
        for (int s = 0; s < cf.nk; s++) cf.sM[s] = mod[cf.smod[s]];
        if (cf.mdr) {
            cf.uM = mod[cf.umod];
            cf.lM = mod[cf.l];
        }
    }
    HIP_CHECK(hipMalloc((void **)&pl.dev, sizeof(ColFuse) * pl.host.size()));
    HIP_CHECK(hipMemcpy(pl.dev, pl.host.data(), sizeof(ColFuse) * pl.host.size(), hipMemcpyHostToDevice));
    return cf_plans.emplace(key, std::move(pl)).first->second;
}
// ModUp at level nl: map d = digit d, sources = the digit's own limbs (inverse scale N^{-1} (D/q_j)^{-1}), targets = every other
// limb of Q_l u P, written to row d*nE + t of the polynomial's digit block
const Context::CfPlan &Context::cf_plan_modup(int nl) {
    const std::string key = "mu:" + std::to_string(nl);
    auto it = cf_plans.find(key);
    if (it != cf_plans.end()) return it->second;
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    const LimbSel esel = sel_ext(nl);
    const ModUpPlan &pl = modup_plan(nl);
    std::vector<ConvTab> tabs(nd);
    HIP_CHECK(hipMemcpy(tabs.data(), pl.d_tabs, sizeof(ConvTab) * nd, hipMemcpyDeviceToHost));
    std::vector<ColFuse> maps(nd);
    for (int d = 0; d < nd; d++) {
        const int lo = d * alpha, hi = std::min(lo + alpha, nl);
        ColFuse &cf = maps[d];
        cf = ColFuse{};
        cf.nk = hi - lo;
        for (int s = 0; s < cf.nk; s++) {
            const u64 qq = q[lo + s];
            cf.smod[s] = lo + s;
            cf.srow[s] = lo + s;
            cf.ssc[s] = mulmod_u64(pl.inv[lo + s] % qq, mod[lo + s].ninv, qq);
            cf.ssc_sh[s] = shoup_h(cf.ssc[s], qq);
        }
        for (int t = 0; t < nE; t++) {
            if (t >= lo && t < hi) continue;
            if (cf.nt >= HY_CF_TGT) throw std::runtime_error("hydia: too many ModUp targets for the column-fused conversion");
            cf.tmod[cf.nt] = esel.mod[t];
            cf.trow[cf.nt] = d * nE + t;
            for (int s = 0; s < cf.nk; s++) cf.f[s][cf.nt] = tabs[d].f[s][t];
            cf.nt++;
        }
    }
    return cf_plan_store(key, std::move(maps));
}
// ModDown at level nl: sources = the special-prime limbs (rows 0..nP-1, inverse scale N^{-1} (P/p_k)^{-1}), targets = limbs 0..nl-1
const Context::CfPlan &Context::cf_plan_moddown(int nl, bool premul) {
    const std::string key = std::string(premul ? "mdp:" : "md:") + std::to_string(nl);
    auto it = cf_plans.find(key);
    if (it != cf_plans.end()) return it->second;
    if (nl > HY_CF_TGT) throw std::runtime_error("hydia: too many ModDown targets for the column-fused conversion");
    std::vector<ColFuse> maps(1);
    ColFuse &cf = maps[0];
    cf = ColFuse{};
    cf.nk = nP;
    cf.nt = nl;
    for (int k = 0; k < nP; k++) {
        const u64 qq = q[nQ + k];
        cf.smod[k] = nQ + k;
        cf.srow[k] = k;
        cf.ssc[k] = mulmod_u64(Phat_inv[k] % qq, mod[nQ + k].ninv, qq);
        cf.ssc_sh[k] = shoup_h(cf.ssc[k], qq);
        for (int j = 0; j < nl; j++) cf.f[k][j] = premul ? mulmod_u64(Phat_mod_q[k][j], Pinv_mod_q[j], q[j]) : Phat_mod_q[k][j];
    }
    for (int j = 0; j < nl; j++) {
        cf.tmod[j] = j;
        cf.trow[j] = j;
    }
    return cf_plan_store(key, std::move(maps));
}
// merged ModDown + Rescale from level nl (l = nl - 1 dropped): sources = row 0 the dropped limb u (N^{-1}), rows 1..nP the special-prime
// limbs; constants with P^{-1} (and the doubling) folded in; targets = limbs 0..l-1
const Context::CfPlan &Context::cf_plan_moddown_rescale(int nl, bool dbl) {
    const std::string key = std::string(dbl ? "mdr2:" : "mdr:") + std::to_string(nl);
    auto it = cf_plans.find(key);
    if (it != cf_plans.end()) return it->second;
    const int l = nl - 1;
    if (l > HY_CF_TGT) throw std::runtime_error("hydia: too many targets for the column-fused conversion");
    std::vector<ColFuse> maps(1);
    ColFuse &cf = maps[0];
    cf = ColFuse{};
    cf.nk = nP;
    cf.nt = l;
    cf.mdr = 1;
    cf.l = l;
    cf.umod = l;
    cf.urow = 0;
    cf.usc = mod[l].ninv;
    cf.usc_sh = mod[l].ninv_sh;
    for (int k = 0; k < nP; k++) {
        const u64 qq = q[nQ + k];
        cf.smod[k] = nQ + k;
        cf.srow[k] = 1 + k;
        cf.ssc[k] = mulmod_u64(Phat_inv[k] % qq, mod[nQ + k].ninv, qq);
        cf.ssc_sh[k] = shoup_h(cf.ssc[k], qq);
        for (int j = 0; j < nl; j++) {
            u64 f = mulmod_u64(Phat_mod_q[k][j], Pinv_mod_q[j], q[j]);
            if (dbl) f = (f + f) % q[j];
            if (j < l) cf.f[k][j] = f;
            else cf.fl[k] = f;
        }
    }
    for (int j = 0; j < l; j++) {
        cf.tmod[j] = j;
        cf.trow[j] = j;
    }
    return cf_plan_store(key, std::move(maps));
}

// Rescale from level nl (l = nl - 1 dropped) as a column-fused map: no conversion sources, the dropped limb u (row 0, N^{-1}) is
// centred and spread to limbs 0..l-1 — pass 1' of u, the spread and pass 1 of every target in one launch
const Context::CfPlan &Context::cf_plan_rescale(int nl) {
    const std::string key = "rs:" + std::to_string(nl);
    auto it = cf_plans.find(key);
    if (it != cf_plans.end()) return it->second;
    const int l = nl - 1;
    if (l > HY_CF_TGT) throw std::runtime_error("hydia: too many targets for the column-fused conversion");
    std::vector<ColFuse> maps(1);
    ColFuse &cf = maps[0];
    cf = ColFuse{};
    cf.nk = 0;
    cf.nt = l;
    cf.mdr = 1;
    cf.l = l;
    cf.umod = l;
    cf.urow = 0;
    cf.usc = mod[l].ninv;
    cf.usc_sh = mod[l].ninv_sh;
    for (int j = 0; j < l; j++) {
        cf.tmod[j] = j;
        cf.trow[j] = j;
    }
    return cf_plan_store(key, std::move(maps));
}

// ModUp of hybrid key switching: for each digit d (limbs [d*alpha, min((d+1)alpha, nl))) the digit's residues are
// extended to every other limb of Q_l u P by fast base conversion.  The (D/q_j)^{-1} factors ride on the inverse
// NTT's N^{-1} scaling, so the conversion kernel is a pure lazy multiply-accumulate.  One inverse transform and one conversion
// launch serve all digits; the forward transforms run per digit (the digit's own limbs are skipped) unless the launch is small
// (a query's fixed-cost tail), where one launch over every row of every digit beats nd launches that each leave most CUs idle —
// the own rows then hold transformed garbage nobody reads (copy_own overwrites them, the inner product reads the input itself).
void Context::modup_digits(const u64 *c, size_t c_outer, int X, int nl, u64 *dig, bool copy_own, bool p1_only, const ProdSrc *ps, u64 *d2_out) {
    if (ps && !cf_ok()) throw std::runtime_error("hydia: fused product outside the column-fused pipeline");
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    const size_t dig_x = (size_t)nd * nE * N;
    const LimbSel esel = sel_ext(nl), qsel = sel_q(nl);
    const ModUpPlan &pl = modup_plan(nl);
    u64 *y = pool.get((size_t)X * nl * N * sizeof(u64));
    if (cf_ok()) {
        // inverse pass 2' of every limb, then ONE column-fused launch: pass 1' of a digit's limbs, conversion, pass 1 of its targets
        const CfPlan &cp = cf_plan_modup(nl);
        const bool small = hk::ntt15_colfuse_small(X, nd);  // few workgroups: the inverse transform as its own (wider) launch
        if (ps) {  // d2 = a1 b1 in the load of inverse pass 2' (+ kept in d2_out); small launches: pass 1' as its own launch
            hk::ntt15_inverse_p2_prod(stream, tabs, *ps, y, (size_t)nl * N, X, qsel, d2_out);
            if (small) hk::ntt15_inverse_p1(stream, tabs, y, (size_t)nl * N, X, qsel, scale_of(qsel, pl.inv, true));
        } else if (small) {
            ntt_inv(c, y, c_outer, (size_t)nl * N, X, qsel, scale_of(qsel, pl.inv, true));
        } else {
            hk::ntt15_inverse_p2(stream, tabs, c, y, c_outer, (size_t)nl * N, X, qsel);
        }
        hk::ntt15_colfuse(stream, tabs, y, (size_t)nl * N, dig, dig_x, X, cp.dev, cp.host.data(), nd, small);
        pool.put(y);
        if (!p1_only) {
            if ((size_t)X * nd * nE < 128) {  // small launch: one second pass over every row (the own rows hold garbage nobody reads)
                hk::ntt15_forward_p2(stream, tabs, dig, (size_t)nE * N, X * nd, esel);
            } else {
                for (int d = 0; d < nd; d++) {
                    const int lo = d * alpha, hi = std::min(lo + alpha, nl);
                    u64 *out = dig + (size_t)d * nE * N;
                    LimbSel rest{};
                    rest.n = nE - hi;
                    for (int t = hi; t < nE; t++) rest.mod[t - hi] = esel.mod[t];
                    if (lo > 0) hk::ntt15_forward_p2(stream, tabs, out, dig_x, X, sel_range(0, lo));
                    if (rest.n > 0) hk::ntt15_forward_p2(stream, tabs, out + (size_t)hi * N, dig_x, X, rest);
                }
            }
        }
        if (copy_own)
            for (int d = 0; d < nd; d++) {
                const int lo = d * alpha, hi = std::min(lo + alpha, nl);
                hk::copy_limbs(stream, N, c + (size_t)lo * N, dig + (size_t)d * nE * N + (size_t)lo * N, c_outer, dig_x, X, hi - lo);
            }
        return;
    }
    ntt_inv(c, y, c_outer, (size_t)nl * N, X, qsel, scale_of(qsel, pl.inv, true));
    hk::base_convert_digits(stream, d_mod, N, y, (size_t)nl * N, dig, dig_x, X, pl.d_tabs, nd, nl, nE, esel);
    const bool merged = !modup_per_digit && nd > 1 && (size_t)X * nd * nE < 128;
    if (merged) {
        if (p1_only) hk::ntt15_forward_p1(stream, tabs, dig, dig, (size_t)nE * N, (size_t)nE * N, X * nd, esel);
        else ntt_fwd(dig, (size_t)nE * N, X * nd, esel);
    }
    for (int d = 0; d < nd; d++) {
        const int lo = d * alpha, hi = std::min(lo + alpha, nl), sz = hi - lo;
        u64 *out = dig + (size_t)d * nE * N;
        LimbSel rest{};
        rest.n = nE - hi;
        for (int t = hi; t < nE; t++) rest.mod[t - hi] = esel.mod[t];
        if (!merged) {
            if (p1_only) {  // the caller runs the second pass fused with the inner product
                if (lo > 0) hk::ntt15_forward_p1(stream, tabs, out, out, dig_x, dig_x, X, sel_range(0, lo));
                if (rest.n > 0) hk::ntt15_forward_p1(stream, tabs, out + (size_t)hi * N, out + (size_t)hi * N, dig_x, dig_x, X, rest);
            } else {
                if (lo > 0) ntt_fwd(out, dig_x, X, sel_range(0, lo));
                ntt_fwd(out + (size_t)hi * N, dig_x, X, rest);
            }
        }
        if (copy_own) hk::copy_limbs(stream, N, c + (size_t)lo * N, out + (size_t)lo * N, c_outer, dig_x, X, sz);
    }
    pool.put(y);
}

// <digits, key> over Q_l u P, then ModDown by P; optionally adds `addend` and applies the evaluation-form
// automorphism (EvalFastRotation's tail).  out: [X][2][nl][N].
void Context::ks_apply(const u64 *dig, size_t dig_x_stride, int X, int nl, const u64 *const *d_keys, int same_key,
                       const u64 *addend, size_t add_x_stride, size_t add_poly_stride, int add_polys,
                       const unsigned *d_galois, const unsigned *d_ginv, int same_galois, bool dbl, u64 *out, int keys_packed_nQ) {
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    const LimbSel esel = sel_ext(nl);
    const LimbSel qsel = sel_q(nl);
    const LimbSel psel = sel_range(nQ, nT);
    std::vector<u64> pinv(Pinv_mod_q.begin(), Pinv_mod_q.begin() + nl);
    ConvTab tab{};
    tab.ns = nP;
    tab.nt = nl;
    tab.skip_lo = tab.skip_hi = 0;
    for (int k = 0; k < nP; k++)
        for (int j = 0; j < nl; j++) tab.f[k][j] = Phat_mod_q[k][j];
    // Loop A (one ModUp shared by all X rotations, one key per rotation): the Q limbs of <digits, key> are formed inside the ModDown
    // transform's epilogue (NttStore mode 5), so only the special-prime limbs of the accumulator ever exist in HBM — the
    // [X][2][nl][N] part (3 of 4 GiB at X = 511) is neither written nor read back
    const bool premul = keys_packed_nQ > 0 && rotptrs_premul;  // the packed shadow's Q-limb rows already carry P^{-1}
    if (premul && !(prm.logN == 15 && fuse_loop_a && dig_x_stride == 0 && !same_key && !dbl))
        throw std::runtime_error("hydia: pre-scaled rotation keys outside the fused loop A");
    if (prm.logN == 15 && fuse_loop_a && dig_x_stride == 0 && !same_key && !dbl) {
        if (premul)
            for (int k = 0; k < nP; k++)
                for (int j = 0; j < nl; j++) tab.f[k][j] = mulmod_u64(Phat_mod_q[k][j], Pinv_mod_q[j], q[j]);
        {
            const int x0 = 0, Xc = X;
            u64 *y = pool.get((size_t)Xc * 2 * nP * N * sizeof(u64));
            const bool ip_in_intt = !loop_a_separate_ip;
            if (ip_in_intt) {
                // the special-prime limbs of <digits, key> are formed in the load of the ModDown inverse transform (never in HBM)
                LoopAIp lp{};
                lp.keys = d_keys + x0;
                lp.dig = dig;
                lp.nd = nd;
                lp.dig_rows = nE;
                lp.nT = nT;
                lp.packed_nQ = keys_packed_nQ;
                lp.key_row0 = nQ;
                lp.dig_row0 = nl;
                hk::ntt15_inverse_loop_a(stream, tabs, y, (size_t)nP * N, Xc * 2, psel, scale_of(psel, Phat_inv, true), lp, /*p1=*/!cf_ok());
            } else {
                u64 *accp = pool.get((size_t)Xc * 2 * nP * N * sizeof(u64));
                timer_begin("ks_inner_product");
                hk::inner_product(stream, d_mod, N, dig, 0, nd, d_keys + x0, 0, nT, accp, Xc, psel, nullptr, 0, 1, 0, nP, keys_packed_nQ, nE, nl);
                timer_end("ks_inner_product");
                if (cf_ok()) hk::ntt15_inverse_p2(stream, tabs, accp, y, (size_t)nP * N, (size_t)nP * N, Xc * 2, psel);
                else ntt_inv(accp, y, (size_t)nP * N, (size_t)nP * N, Xc * 2, psel, scale_of(psel, Phat_inv, true));
                pool.put(accp);
            }
            const bool cfu = cf_ok();
            u64 *conv = pool.get((size_t)Xc * 2 * nl * N * sizeof(u64));
            if (cfu) {
                const CfPlan &cp = cf_plan_moddown(nl, premul);
                hk::ntt15_colfuse(stream, tabs, y, (size_t)nP * N, conv, (size_t)nl * N, Xc * 2, cp.dev, cp.host.data(), 1);
            } else {
                hk::base_convert(stream, d_mod, N, y, (size_t)nP * N, conv, (size_t)nl * N, Xc * 2, tab, qsel);
            }
            pool.put(y);
            NttLoad ld{};
            NttStore stp{};
            stp.mode = 5;
            stp.out = out + (size_t)x0 * 2 * nl * N;
            stp.nl = nl;
            stp.mul = scale_of(qsel, pinv, false);
            stp.addend = addend ? addend + (size_t)x0 * add_x_stride : nullptr;
            stp.add_x = add_x_stride;
            stp.add_p = add_poly_stride;
            stp.add_polys = add_polys;
            stp.ginv = d_ginv ? (same_galois ? d_ginv : d_ginv + x0) : nullptr;
            stp.same_g = same_galois;
            stp.la.keys = d_keys + x0;
            stp.la.dig = dig;
            stp.la.nd = nd;
            stp.la.dig_rows = nE;
            stp.la.nT = nT;
            stp.la.packed_nQ = keys_packed_nQ;
            stp.la.premul = premul ? 1 : 0;
            stp.la.fp = loop_a_int_ip ? 0 : 1;
            stp.la.limb_fastest = loop_a_limb_fastest ? 1 : 0;
            if (cfu) hk::ntt15_forward_p2_fused(stream, tabs, conv, (size_t)nl * N, Xc * 2, qsel, stp);
            else hk::ntt15_forward_fused(stream, tabs, conv, conv, (size_t)nl * N, (size_t)nl * N, Xc * 2, qsel, ld, stp);
            pool.put(conv);
        }
        return;
    }
    u64 *acc = pool.get((size_t)X * 2 * nE * N * sizeof(u64));
    timer_begin("ks_inner_product");
    hk::inner_product(stream, d_mod, N, dig, dig_x_stride, nd, d_keys, same_key, nT, acc, X, esel, nullptr, 0, 1, 0, 0, keys_packed_nQ);
    timer_end("ks_inner_product");
    // P limbs -> coefficient form, pre-multiplied by (P/p_k)^{-1}
    u64 *y = pool.get((size_t)X * 2 * nP * N * sizeof(u64));
    u64 *conv = pool.get((size_t)X * 2 * nl * N * sizeof(u64));
    if (cf_ok()) {
        // special-prime limbs: pass 2' alone, then pass 1' + conversion + pass 1 in one column-fused launch, then pass 2 + ModDown combine
        const CfPlan &cp = cf_plan_moddown(nl, false);
        const bool small = cf_small_moddown(X * 2);
        if (small) ntt_inv(acc + (size_t)nl * N, y, (size_t)nE * N, (size_t)nP * N, X * 2, psel, scale_of(psel, Phat_inv, true));
        else hk::ntt15_inverse_p2(stream, tabs, acc + (size_t)nl * N, y, (size_t)nE * N, (size_t)nP * N, X * 2, psel);
        hk::ntt15_colfuse(stream, tabs, y, (size_t)nP * N, conv, (size_t)nl * N, X * 2, cp.dev, cp.host.data(), 1, small);
        NttStore stp{};
        stp.mode = 1;
        stp.out = out;
        stp.nl = nl;
        stp.in = acc;
        stp.in_ls = nE;
        stp.mul = scale_of(qsel, pinv, false);
        stp.addend = addend;
        stp.add_x = add_x_stride;
        stp.add_p = add_poly_stride;
        stp.add_polys = add_polys;
        stp.dbl = dbl ? 1 : 0;
        stp.ginv = d_ginv;
        stp.same_g = same_galois;
        hk::ntt15_forward_p2_fused(stream, tabs, conv, (size_t)nl * N, X * 2, qsel, stp);
        pool.put(conv);
        pool.put(y);
        pool.put(acc);
        return;
    }
    ntt_inv(acc + (size_t)nl * N, y, (size_t)nE * N, (size_t)nP * N, X * 2, psel, scale_of(psel, Phat_inv, true));
    if (prm.logN == 15) {
        // ModDown combine (+ addend, doubling, automorphism scatter) fused into the NTT's second pass; the P -> Q base
        // conversion runs as its own all-targets kernel
        NttLoad ld{};
        hk::base_convert(stream, d_mod, N, y, (size_t)nP * N, conv, (size_t)nl * N, X * 2, tab, qsel);
        NttStore stp{};
        stp.mode = 1;
        stp.out = out;
        stp.nl = nl;
        stp.in = acc;
        stp.in_ls = nE;
        stp.mul = scale_of(qsel, pinv, false);
        stp.addend = addend;
        stp.add_x = add_x_stride;
        stp.add_p = add_poly_stride;
        stp.add_polys = add_polys;
        stp.dbl = dbl ? 1 : 0;
        stp.ginv = d_ginv;
        stp.same_g = same_galois;
        hk::ntt15_forward_fused(stream, tabs, conv, conv, (size_t)nl * N, (size_t)nl * N, X * 2, qsel, ld, stp);
    } else {
        hk::base_convert(stream, d_mod, N, y, (size_t)nP * N, conv, (size_t)nl * N, X * 2, tab, qsel);
        ntt_fwd(conv, (size_t)nl * N, X * 2, qsel);
        hk::moddown_combine(stream, d_mod, prm.logN, acc, nE, conv, addend, add_x_stride, add_poly_stride, add_polys, out, X, nl,
                            scale_of(qsel, pinv, false), d_galois, same_galois);
        if (dbl) hk::add(stream, d_mod, N, out, out, out, X * 2, qsel, nl, nl, nl);
    }
    pool.put(conv);
    pool.put(y);
    pool.put(acc);
}

void Context::build_rotptrs() {
    if (rotptrs_valid) return;
    std::vector<const u64 *> ptrs(prm.dim, nullptr);
    std::vector<unsigned> gal(prm.dim, 1u), ginv(prm.dim, 1u);
    // loop A streams every rotation key exactly once per query: give it a packed shadow (6-byte residues for the < 2^48 limbs)
    bool pack = rot_packed && prm.dim > 1;
    for (int j = 1; j < nQ; j++)
        if (q[j] >> 48) pack = false;
    const size_t kb = hk::key_packed_bytes(N, nQ, nT, prm.dnum);
    sync_all();  // a re-key must not repack the shadow under a query still in flight
    if (d_rotpack && !pack) {
        (void)hipFree(d_rotpack);
        d_rotpack = nullptr;
    }
    if (pack && !d_rotpack && hipMalloc((void **)&d_rotpack, kb * (size_t)(prm.dim - 1)) != hipSuccess) {
        d_rotpack = nullptr;  // not enough HBM for the shadow: loop A reads the plain keys
        pack = false;
    }
    // with the fused loop A the shadow's Q-limb rows carry P^{-1} (see k_key_pack): only that path reads them
    const bool premul = pack && fuse_loop_a && prm.logN == 15;
    std::vector<u64> pinv_all(Pinv_mod_q.begin(), Pinv_mod_q.begin() + nQ);
    const ScaleSel pinv_sel = scale_of(sel_q(nQ), pinv_all, false);
    for (int i = 1; i < prm.dim; i++) {
        auto it = rot_keys.find(i);
        if (it == rot_keys.end()) throw StateError("hydia: rotation key " + std::to_string(i) + " not loaded");
        if (pack) {
            hk::key_pack(stream, d_mod, N, nQ, nT, prm.dnum, it->second.d, d_rotpack + kb * (size_t)(i - 1), premul ? &pinv_sel : nullptr);
            ptrs[i] = reinterpret_cast<const u64 *>(d_rotpack + kb * (size_t)(i - 1));
        } else {
            ptrs[i] = it->second.d;
        }
        gal[i] = (unsigned)galois_elt(i);
        u64 x = 1;
        for (int it = 0; it < 6; it++) x = x * (2 - (u64)gal[i] * x);
        ginv[i] = (unsigned)(x & (2ull * N - 1));
    }
    rotptrs_packed = pack;
    rotptrs_premul = premul;
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipMemcpy(d_rotginv, ginv.data(), sizeof(unsigned) * prm.dim, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy((void *)d_rotptrs, ptrs.data(), sizeof(u64 *) * prm.dim, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_rotgalois, gal.data(), sizeof(unsigned) * prm.dim, hipMemcpyHostToDevice));
    rotptrs_valid = true;
}

void Context::ks_fused(const u64 *c1, size_t c1_xs, int X, int nl, const u64 *key, const u64 *const *d_key_cell, const u64 *const *d_keys,
                       const u64 *addend, size_t add_x, size_t add_p, int add_polys, const unsigned *d_ginv, int same_g, bool dbl, u64 *out) {
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha, XP = 2 * X;
    const LimbSel qsel = sel_q(nl), psel = sel_range(nQ, nT);
    u64 *dig = pool.get((size_t)X * nd * nE * N * sizeof(u64));
    modup_digits(c1, c1_xs, X, nl, dig, /*copy_own=*/false, /*p1_only=*/true);
    u64 *acc = pool.get((size_t)XP * nE * N * sizeof(u64));
    u64 *y = pool.get((size_t)XP * nP * N * sizeof(u64));  // raw image of the special-prime sums' inverse pass 2'
    timer_begin("ks_inner_product");
    hk::ntt15_p2_inner_product(stream, tabs, d_mod, dig, (size_t)nd * nE * N, nd, X, nl, nP, nT, alpha, d_keys ? d_keys : d_key_cell, key, c1, c1_xs, acc,
                               y, (size_t)nP * N, 0, nullptr, d_keys != nullptr);
    timer_end("ks_inner_product");
    pool.put(dig);
    const CfPlan &cp = cf_plan_moddown(nl, false);
    const bool small = cf_small_moddown(XP);
    if (small) hk::ntt15_inverse_p1(stream, tabs, y, (size_t)nP * N, XP, psel, scale_of(psel, Phat_inv, true));
    u64 *conv = pool.get((size_t)XP * nl * N * sizeof(u64));
    hk::ntt15_colfuse(stream, tabs, y, (size_t)nP * N, conv, (size_t)nl * N, XP, cp.dev, cp.host.data(), 1, small);
    pool.put(y);
    std::vector<u64> pinv(Pinv_mod_q.begin(), Pinv_mod_q.begin() + nl);
    NttStore stp{};
    stp.mode = 1;
    stp.out = out;
    stp.nl = nl;
    stp.in = acc;
    stp.in_ls = nE;
    stp.mul = scale_of(qsel, pinv, false);
    stp.addend = addend;
    stp.add_x = add_x;
    stp.add_p = add_p;
    stp.add_polys = add_polys;
    stp.dbl = dbl ? 1 : 0;
    stp.ginv = d_ginv;
    stp.same_g = same_g;
    hk::ntt15_forward_p2_fused(stream, tabs, conv, (size_t)nl * N, XP, qsel, stp);
    pool.put(conv);
    pool.put(acc);
}

// RelinearizeInPlace (sender_diag.cpp:79)
void Context::relinearize(Ct &c, bool dbl) {
    if (c.npoly != 3) return;
    if (!relin_key.d) throw StateError("hydia: relinearisation key not loaded");
    const int nl = c.nl, nE = nl + nP, nd = (nl + alpha - 1) / alpha, X = c.X;
    op_bytes("op:relinearize", N, (double)X * (nd * nE + 2 * nP + 2 * nl), (double)nd * 2 * nE * N * 8);
    if (ks_fused_ok()) {
        Ct out(this, X, 2, nl, c.scale);
        ks_fused(c.d + 2 * c.poly_elems(), c.ct_elems(), X, nl, relin_key.d, relin_key.d_cell, nullptr, c.d, c.ct_elems(), c.poly_elems(), 2, nullptr, 0, dbl,
                 out.d);
        c = std::move(out);
        return;
    }
    u64 *dig = pool.get((size_t)X * nd * nE * N * sizeof(u64));
    modup_digits(c.d + 2 * c.poly_elems(), c.ct_elems(), X, nl, dig);
    Ct out(this, X, 2, nl, c.scale);
    ks_apply(dig, (size_t)nd * nE * N, X, nl, relin_key.d_cell, 1, c.d, c.ct_elems(), c.poly_elems(), 2, nullptr, nullptr, 0, dbl,
             out.d);
    pool.put(dig);
    c = std::move(out);
}
// RescaleInPlace (sender_diag.cpp:80): divide by the last prime with rounding to nearest
void Context::rescale(Ct &c, const Ct *sub, const double *addc) {
    const int nl = c.nl, l = nl - 1, XP = c.X * c.npoly;
    if (nl < 2) throw std::runtime_error("hydia: rescale with one limb left");
    if (sub && (sub->X != c.X || sub->npoly != c.npoly || sub->nl < l)) throw std::runtime_error("hydia: rescale sub operand shape");
    op_bytes("op:rescale", N, (double)XP * nl, 0);
    u64 *t = pool.get((size_t)XP * N * sizeof(u64));
    const LimbSel last = sel_range(l, l + 1);
    u64 *tmp = pool.get((size_t)XP * l * N * sizeof(u64));
    const LimbSel qsel = sel_q(l);
    Ct out(this, c.X, c.npoly, l, c.scale / (double)q[l]);
    std::vector<u64> qi(ql_inv[l].begin(), ql_inv[l].begin() + l);
    // (round 4) batches that fill the chip: inverse pass 2' of the dropped limb, then ONE column-fused launch (pass 1', centring,
    // spread, pass 1 of every remaining limb; u is read once instead of once per limb), then the combine in pass 2
    const bool rcf = rescale_cf && cf_ok() && l <= HY_LC_LIMBS && l <= HY_CF_TGT && (q[l] >> 50) == 0 && XP >= 256;  // (measured: 2^20, 512 polynomials per launch, -0.5 ms per query; 2^17, 64 per launch, +0.1 ms)
    if (rcf) {
        hk::ntt15_inverse_p2(stream, tabs, c.d + (size_t)l * N, t, c.poly_elems(), (size_t)N, XP, last);
        const CfPlan &cp = cf_plan_rescale(nl);
        hk::ntt15_colfuse(stream, tabs, t, (size_t)N, tmp, (size_t)l * N, XP, cp.dev, cp.host.data(), 1, false);
        NttStore stp{};
        stp.mode = 2;
        stp.out = out.d;
        stp.nl = l;
        stp.in = c.d;
        stp.in_ls = c.lstride;
        stp.mul = scale_of(qsel, qi, false);
        stp.sub = sub ? sub->d : nullptr;
        stp.sub_ls = sub ? sub->lstride : 0;
        stp.has_addc = addc ? 1 : 0;
        stp.npoly = c.npoly;
        if (addc)
            for (int j = 0; j < l; j++) stp.addc[j] = double_to_mod(*addc * out.scale, q[j]);
        hk::ntt15_forward_p2_fused(stream, tabs, tmp, (size_t)l * N, XP, qsel, stp);
        pool.put(tmp);
        pool.put(t);
        c = std::move(out);
        return;
    }
    ntt_inv(c.d + (size_t)l * N, t, c.poly_elems(), (size_t)N, XP, last, scale_ninv(last));
    if (prm.logN == 15 && l <= HY_LC_LIMBS) {
        // spread fused into the NTT's first pass, combine (+ the caller's subtraction / constant) into its second
        NttLoad ld{};
        ld.mode = 2;
        ld.y = t;
        ld.y_outer = (size_t)N;
        ld.l = l;
        NttStore stp{};
        stp.mode = 2;
        stp.out = out.d;
        stp.nl = l;
        stp.in = c.d;
        stp.in_ls = c.lstride;
        stp.mul = scale_of(qsel, qi, false);
        stp.sub = sub ? sub->d : nullptr;
        stp.sub_ls = sub ? sub->lstride : 0;
        stp.has_addc = addc ? 1 : 0;
        stp.npoly = c.npoly;
        if (addc)
            for (int j = 0; j < l; j++) stp.addc[j] = double_to_mod(*addc * out.scale, q[j]);
        hk::ntt15_forward_fused(stream, tabs, nullptr, tmp, 0, (size_t)l * N, XP, qsel, ld, stp);
    } else {
        hk::rescale_spread(stream, d_mod, N, t, tmp, XP, l);
        ntt_fwd(tmp, (size_t)l * N, XP, qsel);
        hk::rescale_combine(stream, d_mod, N, c.d, tmp, out.d, XP, l, scale_of(qsel, qi, false), c.lstride);
        if (sub) {
            Ct sv = sub->alias(l);
            sub_inplace(out, sv);
        }
        if (addc) add_const(out, *addc);
    }
    pool.put(tmp);
    pool.put(t);
    c = std::move(out);
}
void Context::relin_rescale(Ct &c, bool dbl, const Ct *sub, const double *addc, bool sub_is_add) {
    if (c.npoly != 3) throw std::runtime_error("hydia: relin_rescale needs a 3-component ciphertext");
    const int nl = c.nl, l = nl - 1;
    if (prm.logN != 15 || !merge_rescale || l < 1 || l > HY_LC_LIMBS) {  // generic rings: the two steps in sequence
        relinearize(c, dbl);
        rescale(c, sub_is_add ? nullptr : sub, addc);
        if (sub && sub_is_add) {
            Ct sv = sub->alias(l);
            add_inplace(c, sv);
        }
        return;
    }
    if (!relin_key.d) throw StateError("hydia: relinearisation key not loaded");
    if (sub && (sub->X != c.X || sub->npoly != 2 || sub->nl < l)) throw std::runtime_error("hydia: rescale sub operand shape");
    Ct out(this, c.X, 2, l, c.scale / (double)q[l]);
    relin_rescale_into(c, dbl, sub, addc, sub_is_add, out.d);
    c = std::move(out);
}
// the merged pipeline: c [X][3][nl][N] -> out_d [X][2][nl - 1][N]
void Context::relin_rescale_into(const Ct &c, bool dbl, const Ct *sub, const double *addc, bool sub_is_add, u64 *out_d, const ProdSrc *ps,
                                 const ScaleSel *kap) {
    const int nl = c.nl, l = nl - 1;
    const int nE = nl + nP, nd = (nl + alpha - 1) / alpha, X = c.X, XP = X * 2;
    // relinearise (ModUp nd nE, ModDown 2 (nP + nl)) + rescale (2 nl) transforms per ciphertext: 80 + 24 at nl = 12
    op_bytes("op:relin_rescale", N, (double)X * (nd * nE + 2 * nP + 4 * nl), (double)nd * 2 * nE * N * 8);
    u64 *dig = pool.get((size_t)X * nd * nE * N * sizeof(u64));
    // fused product: d2 is formed in the inverse transform's load and kept (compact) for the inner product's own-digit rows
    u64 *d2buf = ps ? pool.get((size_t)X * nl * N * sizeof(u64)) : nullptr;
    const u64 *c2 = ps ? d2buf : c.d + 2 * c.poly_elems();
    const size_t c2_xs = ps ? (size_t)nl * N : c.ct_elems();
    // the inner product reads a digit's own limbs from c2, and (fuse_ip) consumes the ModUp transforms' second pass directly
    const bool fip = fuse_ip;
    modup_digits(c2, c2_xs, X, nl, dig, /*copy_own=*/false, /*p1_only=*/fip, ps, d2buf);
    const LimbSel esel = sel_ext(nl);
    u64 *acc = pool.get((size_t)XP * nE * N * sizeof(u64));
    // yu [XP][1 + nP][N]: row 0 = the dropped limb u, rows 1.. = the special-prime limbs, on their way to the coefficient domain
    const size_t yu_outer = (size_t)(1 + nP) * N;
    u64 *yu = pool.get((size_t)XP * yu_outer * sizeof(u64));
    // with the fused inner product the special-prime sums never reach acc: the kernel runs the first pass of their inverse
    // transform itself and leaves its raw image in yu (HYDIA_RELIN_SEPARATE_INTT: through acc, as before)
    const bool tail_in_ip = !relin_separate_intt;
    const bool fused_tail = fip && tail_in_ip && prm.logN == 15;
    const LimbSel qsel_full = sel_q(nl);
    std::vector<u64> pinv(Pinv_mod_q.begin(), Pinv_mod_q.begin() + nl);
    const ScaleSel pinv_sel = scale_of(qsel_full, pinv, false);
    // (round 4) the dropped limb's sums take the same tail inside the merged kernel: (sum P^{-1} + d_l)(x2), inverse pass 2, row 0 of yu
    const DropLimb drop{l, dbl ? 1 : 0, pinv_sel.s[l], pinv_sel.s_sh[l], c.d, ps ? 0 : c.ct_elems(), ps ? 0 : c.poly_elems(), ps};
    bool drop_done = false;
    timer_begin("ks_inner_product");
    if (fip)
        drop_done = hk::ntt15_p2_inner_product(stream, tabs, d_mod, dig, (size_t)nd * nE * N, nd, X, nl, nP, nT, alpha, relin_key.d_cell, relin_key.d,
                                               c2, c2_xs, acc, fused_tail ? yu : nullptr, yu_outer, 1, fused_tail ? &drop : nullptr);
    else
        hk::inner_product(stream, d_mod, N, dig, (size_t)nd * nE * N, nd, relin_key.d_cell, 1, nT, acc, X, esel, c2, c2_xs, alpha, nl);
    timer_end("ks_inner_product");
    pool.put(dig);
    if (d2buf) pool.put(d2buf);
    if (ps && !(drop_done && fused_tail)) throw std::runtime_error("hydia: fused product outside the merged pipeline (prod_fusable out of step)");
    // limb l of the would-be ModDown output (+ d_l, doubled) replaces row l of the accumulator (nothing else reads that row), so that
    // ONE inverse transform takes rows l .. nE-1 — the dropped limb and the special-prime limbs (pre-multiplied by (P/p_k)^{-1}) —
    // to the coefficient domain: yu [XP][1 + nP][N], row 0 = u
    if (!fused_tail)
        hk::moddown_last_limb(stream, d_mod, N, acc, nE, c.d, c.ct_elems(), c.poly_elems(), acc + (size_t)l * N, (size_t)nE * N, XP, l,
                              pinv_sel.s[l], pinv_sel.s_sh[l], dbl ? 1 : 0);
    LimbSel tail{};
    tail.n = 1 + nP;
    tail.mod[0] = l;
    std::vector<u64> tail_scale(1 + nP, 1);
    for (int k = 0; k < nP; k++) {
        tail.mod[1 + k] = nQ + k;
        tail_scale[1 + k] = Phat_inv[k];
    }
    const bool cfu = cf_ok() && (q[l] >> 50) == 0;  // the fused kernel carries the dropped limb's centred residue as a double
    const bool cf_pre = cfu && cf_small_moddown(XP);  // few workgroups: pass 1' as its own (wider) launch
    if (fused_tail) {
        if (!drop_done)
            hk::ntt15_inverse_p2_last_limb(stream, tabs, acc + (size_t)l * N, yu, (size_t)nE * N, yu_outer, XP, l, pinv_sel.s[l], pinv_sel.s_sh[l],
                                           c.d, c.ct_elems(), c.poly_elems(), dbl ? 1 : 0);
        if (!cfu || cf_pre) hk::ntt15_inverse_p1(stream, tabs, yu, yu_outer, XP, tail, scale_of(tail, tail_scale, true));
    } else if (cfu && !cf_pre) {
        hk::ntt15_inverse_p2(stream, tabs, acc + (size_t)l * N, yu, (size_t)nE * N, yu_outer, XP, tail);
    } else {
        ntt_inv(acc + (size_t)l * N, yu, (size_t)nE * N, yu_outer, XP, tail, scale_of(tail, tail_scale, true));
    }
    const u64 *u = yu, *y = yu + N;
    // coefficient-domain correction of every remaining limb, then ONE forward NTT per limb with the merged epilogue
    ConvTab tab{};
    tab.ns = nP;
    tab.nt = nl;
    for (int k = 0; k < nP; k++)
        for (int j = 0; j < nl; j++) {  // P^{-1} and the doubling folded into the conversion constants
            u64 f = mulmod_u64(Phat_mod_q[k][j], Pinv_mod_q[j], q[j]);
            tab.f[k][j] = dbl ? (f + f) % q[j] : f;
        }
    u64 *w = pool.get((size_t)XP * l * N * sizeof(u64));
    if (cfu) {  // pass 1' of u and the special-prime limbs, the correction of every remaining limb and its pass 1: one launch
        const CfPlan &cp = cf_plan_moddown_rescale(nl, dbl);
        hk::ntt15_colfuse(stream, tabs, yu, yu_outer, w, (size_t)l * N, XP, cp.dev, cp.host.data(), 1, cf_pre);
    } else {
        hk::moddown_rescale_conv(stream, d_mod, N, y, yu_outer, u, yu_outer, w, XP, l, nP, tab);
    }
    const LimbSel qsel = sel_q(l);
    const double out_scale = c.scale / (double)q[l];
    std::vector<u64> qi(ql_inv[l].begin(), ql_inv[l].begin() + l);
    NttLoad ld{};
    NttStore stp{};
    stp.mode = 3;
    stp.out = out_d;
    stp.nl = l;
    stp.in = acc;
    stp.in_ls = nE;
    stp.mul = pinv_sel;
    stp.mul2 = scale_of(qsel, qi, false);
    stp.addend = c.d;
    stp.add_x = ps ? 0 : c.ct_elems();
    stp.add_p = ps ? 0 : c.poly_elems();
    stp.add_polys = 2;
    if (ps) {  // d0, d1 of the product are formed in the epilogue
        stp.has_prod = 1;
        stp.prod = *ps;
        if (ps->c) stp.kap = *kap;
    }
    stp.dbl = dbl ? 1 : 0;
    stp.sub = sub ? sub->d : nullptr;
    stp.sub_ls = sub ? sub->lstride : 0;
    stp.sub_add = sub_is_add ? 1 : 0;
    stp.has_addc = addc ? 1 : 0;
    stp.npoly = 2;
    if (addc)
        for (int j = 0; j < l; j++) stp.addc[j] = double_to_mod(*addc * out_scale, q[j]);
    if (cfu) hk::ntt15_forward_p2_fused(stream, tabs, w, (size_t)l * N, XP, qsel, stp);
    else hk::ntt15_forward_fused(stream, tabs, w, w, (size_t)l * N, (size_t)l * N, XP, qsel, ld, stp);
    pool.put(w);
    pool.put(yu);
    pool.put(acc);
}
Ct Context::clone(const Ct &a) {
    Ct o(this, a.X, a.npoly, a.nl, a.scale);
    hk::copy_limbs(stream, N, a.d, o.d, a.poly_elems(), o.poly_elems(), a.X * a.npoly, a.nl);
    return o;
}
void Context::drop_to(Ct &a, int nl) {
    if (nl < a.nl) a.nl = nl;
}
static void check_same(const Ct &a, const Ct &b, const char *what) {
    if (a.X != b.X || a.npoly != b.npoly || a.nl != b.nl)
        throw std::runtime_error(std::string("hydia: shape mismatch in ") + what);
}
void Context::add_inplace(Ct &a, const Ct &b) {
    check_same(a, b, "add");
    op_bytes("op:add", N, 0, 3.0 * a.X * a.npoly * a.nl * N * 8);
    hk::add(stream, d_mod, N, a.d, b.d, a.d, a.X * a.npoly, sel_q(a.nl), a.lstride, b.lstride, a.lstride);
}
void Context::sub_inplace(Ct &a, const Ct &b) {
    check_same(a, b, "sub");
    op_bytes("op:add", N, 0, 3.0 * a.X * a.npoly * a.nl * N * 8);
    hk::sub(stream, d_mod, N, a.d, b.d, a.d, a.X * a.npoly, sel_q(a.nl), a.lstride, b.lstride, a.lstride);
}
// EvalAddInPlace(ct, double) (openFHE_wrapper.cpp:182)
void Context::add_const(Ct &a, double c) {
    const LimbSel s = sel_q(a.nl);
    ScaleSel sc{};
    for (int j = 0; j < a.nl; j++) sc.s[j] = double_to_mod(c * a.scale, q[j]);
    hk::add_scalar(stream, d_mod, N, a.d, a.ct_elems(), a.X, s, sc);
}
Ct Context::mul_const(const Ct &a, double c, double const_scale) {
    const LimbSel s = sel_q(a.nl);
    ScaleSel sc{};
    for (int j = 0; j < a.nl; j++) {
        sc.s[j] = double_to_mod(c * const_scale, q[j]);
        sc.s_sh[j] = shoup_h(sc.s[j], q[j]);
    }
    Ct o(this, a.X, a.npoly, a.nl, a.scale * const_scale);
    hk::mul_scalar(stream, d_mod, N, a.d, o.d, a.X * a.npoly, s, sc, a.lstride, o.lstride);
    return o;
}
Ct Context::lincomb(const std::vector<const Ct *> &terms, const std::vector<double> &coef, double c0, double S) {
    if (terms.empty() || terms.size() > HY_LC_TERMS || terms.size() != coef.size()) throw std::runtime_error("hydia: bad lincomb");
    const Ct &f = *terms[0];
    if (f.nl > HY_LC_LIMBS) throw std::runtime_error("hydia: lincomb limb count");
    LinComb lc{};
    lc.nterms = (int)terms.size();
    for (int t = 0; t < lc.nterms; t++) {
        const Ct &a = *terms[t];
        if (a.X != f.X || a.npoly != f.npoly || a.nl != f.nl) throw std::runtime_error("hydia: lincomb shape mismatch");
        lc.src[t] = a.d;
        lc.ls[t] = a.lstride;
        for (int j = 0; j < f.nl; j++) {
            lc.c[t][j] = double_to_mod(coef[t] * (S / a.scale), q[j]);
            lc.cs[t][j] = shoup_h(lc.c[t][j], q[j]);
        }
    }
    for (int j = 0; j < f.nl; j++) lc.c0[j] = double_to_mod(c0 * S, q[j]);
    op_bytes("op:lincomb", N, 0, (double)(lc.nterms + 1) * f.X * f.npoly * f.nl * N * 8);
    Ct o(this, f.X, f.npoly, f.nl, S);
    hk::lincomb(stream, d_mod, N, lc, o.d, f.X, f.npoly, f.nl);
    return o;
}
Ct Context::lincomb_multi(const std::vector<const Ct *> &terms, const std::vector<std::vector<double>> &coef,
                          const std::vector<double> &c0, const std::vector<double> &S) {
    const int K = (int)coef.size(), nt = (int)terms.size();
    if (nt < 1 || nt > HY_LC_TERMS || K < 1 || (int)c0.size() != K || (int)S.size() != K) throw std::runtime_error("hydia: bad lincomb_multi");
    const Ct &f = *terms[0];
    if (f.nl > HY_LC_LIMBS) throw std::runtime_error("hydia: lincomb limb count");
    LinCombMulti lc{};
    lc.nterms = nt;
    lc.K = K;
    lc.fp = (tabs.twf != nullptr || prm.logN != 15) && !getenv_int_arith ? 1 : 0;
    if (lcm_stage.size() >= 64) {  // uploads are asynchronous: recycle the staging buffers only behind a stream fence
        sync_all();
        lcm_stage.clear();
    }
    lcm_stage.emplace_back((size_t)K * HY_LCM_BLOCK, 0);
    std::vector<u64> &lcm_host = lcm_stage.back();
    for (int t = 0; t < nt; t++) {
        const Ct &a = *terms[t];
        if (a.X != f.X || a.npoly != f.npoly || a.nl != f.nl) throw std::runtime_error("hydia: lincomb shape mismatch");
        lc.src[t] = a.d;
        lc.ls[t] = a.lstride;
        for (int k = 0; k < K; k++)
            for (int j = 0; j < f.nl; j++)
                lcm_host[(size_t)k * HY_LCM_BLOCK + t * HY_LC_LIMBS + j] = double_to_mod(coef[k][t] * (S[k] / a.scale), q[j]);
    }
    for (int k = 0; k < K; k++)
        for (int j = 0; j < f.nl; j++) lcm_host[(size_t)k * HY_LCM_BLOCK + HY_LC_TERMS * HY_LC_LIMBS + j] = double_to_mod(c0[k] * S[k], q[j]);
    u64 *tab = pool.get(lcm_host.size() * sizeof(u64));
    HIP_CHECK(hipMemcpyAsync(tab, lcm_host.data(), lcm_host.size() * sizeof(u64), hipMemcpyHostToDevice, stream));
    lc.tab = tab;
    op_bytes("op:lincomb", N, 0, (double)(nt + K) * f.X * f.npoly * f.nl * N * 8);
    Ct o(this, f.X * K, f.npoly, f.nl, S[0]);
    hk::lincomb_multi(stream, d_mod, N, lc, o.d, f.X, f.npoly, f.nl);
    pool.put(tab);
    return o;
}
// EvalMultNoRelin (sender_diag.cpp:93)
Ct Context::mult_norelin(const Ct &a, const Ct &b) {
    if (a.X != b.X || a.nl != b.nl || a.npoly != 2 || b.npoly != 2) throw std::runtime_error("hydia: mult shape mismatch");
    op_bytes("op:mult_norelin", N, 0, 7.0 * a.X * a.nl * N * 8);
    Ct o(this, a.X, 3, a.nl, a.scale * b.scale);
    hk::tensor(stream, d_mod, N, a.d, b.d, o.d, a.X, a.nl, a.lstride, b.lstride);
    return o;
}
// EvalMultNoRelin with c (2 components, same limbs) leaving at the product's scale: d0,d1 -= (K/2 mod q_j) * c0,c1 where
// K = round(s_a*s_b/s_c); the caller's doubling relinearisation turns that into 2ab - K*c
Ct Context::mult_norelin_sub(const Ct &a, const Ct &b, const Ct &c) {
    if (a.X != b.X || a.nl != b.nl || a.npoly != 2 || b.npoly != 2 || c.X != a.X || c.npoly != 2 || c.nl != a.nl)
        throw std::runtime_error("hydia: mult-sub shape mismatch");
    op_bytes("op:mult_norelin", N, 0, 9.0 * a.X * a.nl * N * 8);
    Ct o(this, a.X, 3, a.nl, a.scale * b.scale);
    const u64 K = (u64)std::llround(o.scale / c.scale);
    ScaleSel kap{};
    for (int j = 0; j < a.nl; j++) {
        const u64 inv2 = (q[j] + 1) >> 1;
        kap.s[j] = mulmod_u64(K % q[j], inv2, q[j]);
        kap.s_sh[j] = shoup_h(kap.s[j], q[j]);
    }
    hk::tensor(stream, d_mod, N, a.d, b.d, o.d, a.X, a.nl, a.lstride, b.lstride, c.d, c.lstride, &kap);
    return o;
}
// the conditions under which relin_rescale_into runs its merged pipeline end to end (every consumer of d0, d1, d2 can form them)
bool Context::prod_fusable(int nl) const {
    const int l = nl - 1, nd = (nl + alpha - 1) / alpha;
    return prod_fuse && prm.logN == 15 && merge_rescale && fuse_ip && !relin_separate_intt && cf_ok() && l >= 1 && l <= HY_LC_LIMBS && nd >= 2 &&
           nd <= 4 && !tabs.two_ip_launches && !tabs.no_drop_in_ip && (q[l] >> 50) == 0 && relin_key.d != nullptr;
}
Ct Context::mult_relin_rescale(const Ct &a, const Ct &b, bool dbl, const Ct *sub, const double *addc, bool sub_is_add, const Ct *csub) {
    if (a.X != b.X || a.nl != b.nl || a.npoly != 2 || b.npoly != 2) throw std::runtime_error("hydia: mult shape mismatch");
    if (csub && (csub->X != a.X || csub->npoly != 2 || csub->nl != a.nl || !dbl)) throw std::runtime_error("hydia: mult-sub shape mismatch");
    // (no epilogue handles a product's subtrahend K c AND a `sub` operand at once — mode 10 would drop kap c while the dropped limb's tail
    // kept it; no caller asks for both: cheb_step passes csub, the Paterson-Stockmeyer nodes pass sub)
    if (csub && sub) throw std::logic_error("hydia: mult_relin_rescale takes a product subtrahend (csub) or a sub operand, not both");
    const int nl = a.nl, l = nl - 1, nd = (nl + alpha - 1) / alpha;
    // (the tail's subtrahend path exists in the three-and-more-digit kernels only: the steps with a subtrahend run at >= 9 limbs)
    if (!prod_fusable(nl) || (csub && (!prod_fuse_csub || nd < 3)) || (sub && (sub->X != a.X || sub->npoly != 2 || sub->nl < l))) {
        Ct o = csub ? mult_norelin_sub(a, b, *csub) : mult_norelin(a, b);
        relin_rescale(o, dbl, sub, addc, sub_is_add);
        return o;
    }
    op_bytes("op:mult_norelin", N, 0, (csub ? 9.0 : 7.0) * a.X * a.nl * N * 8);  // the inherent bytes of the product stay what they were
    Ct shape;  // shape and scale of the degree-2 ciphertext that is never formed
    shape.ctx = this;
    shape.X = a.X;
    shape.npoly = 3;
    shape.nl = shape.lstride = nl;
    shape.scale = a.scale * b.scale;
    shape.view = true;
    ProdSrc ps{a.d, b.d, a.ct_elems(), a.poly_elems(), b.ct_elems(), b.poly_elems(), nullptr, 0, 0, 0, 0};
    ScaleSel kap{};
    if (csub) {  // d0, d1 -= (K/2 mod q_j) c0, c1 ahead of the doubling relinearisation
        const u64 K = (u64)std::llround(shape.scale / csub->scale);
        for (int j = 0; j < nl; j++) {
            const u64 inv2 = (q[j] + 1) >> 1;
            kap.s[j] = mulmod_u64(K % q[j], inv2, q[j]);
            kap.s_sh[j] = shoup_h(kap.s[j], q[j]);
        }
        ps.c = csub->d;
        ps.c_x = csub->ct_elems();
        ps.c_p = csub->poly_elems();
        ps.kap_l = kap.s[l];
        ps.kap_l_sh = kap.s_sh[l];
    }
    Ct out(this, a.X, 2, l, shape.scale / (double)q[l]);
    relin_rescale_into(shape, dbl, sub, addc, sub_is_add, out.d, &ps, &kap);
    return out;
}
Ct Context::mult(const Ct &a, const Ct &b) {
    const int nl = std::min(a.nl, b.nl);
    Ct x = a.alias(nl), y = b.alias(nl);
    return mult_relin_rescale(x, y);
}
// EvalRotate on every ciphertext of the batch (EvalSum's step, sender_diag.cpp:47)
Ct Context::rotate(const Ct &a, int rot) {
    auto it = rot_keys.find(rot);
    if (it == rot_keys.end()) throw StateError("hydia: rotation key " + std::to_string(rot) + " not loaded");
    const int nl = a.nl, nE = nl + nP, nd = (nl + alpha - 1) / alpha, X = a.X;
    op_bytes("op:rotate", N, (double)X * (nd * nE + 2 * nP + 2 * nl), (double)nd * 2 * nE * N * 8 + 4.0 * X * nl * N * 8);
    if (ks_fused_ok()) {
        Ct out(this, X, 2, nl, a.scale);
        ks_fused(a.d + a.poly_elems(), a.ct_elems(), X, nl, it->second.d, it->second.d_cell, nullptr, a.d, a.ct_elems(), a.poly_elems(), 1, it->second.d_gal + 1, 1,
                 false, out.d);
        return out;
    }
    u64 *dig = pool.get((size_t)X * nd * nE * N * sizeof(u64));
    modup_digits(a.d + a.poly_elems(), a.ct_elems(), X, nl, dig);
    Ct out(this, X, 2, nl, a.scale);
    ks_apply(dig, (size_t)nd * nE * N, X, nl, it->second.d_cell, 1, a.d, a.ct_elems(), a.poly_elems(), 1, it->second.d_gal,
             it->second.d_gal + 1, 1, false, out.d);
    pool.put(dig);
    return out;
}

// ------------------------------------------------------------------ HyDia sender
// loop A (sender_diag.cpp:20-26): ONE ModUp of the query's c1, then dim-1 hoisted rotations as one batched
// inner-product + ModDown + automorphism sequence.  Output: rot[0] = q, rot[i] = Rot_i(q).
Ct Context::rotate_query(const Ct &qc) {
    if (qc.X != 1 || qc.npoly != 2 || !qc.compact()) throw std::runtime_error("hydia: query must be one 2-component ciphertext");
    Ct rot(this, prm.dim, 2, qc.nl, qc.scale);
    rotate_query_range(qc, 0, prm.dim, rot.d);
    return rot;
}
// A contiguous range of the hoisted rotations (the whole loop A for first = 0, count = dim).  Every range repeats the ModUp of c1
// (48 transforms against 32 per rotation), so R ranks that each take dim / R rotations do loop A's work once between them.
void Context::rotate_query_range(const Ct &qc, int first, int count, u64 *out) {
    if (qc.X != 1 || qc.npoly != 2 || !qc.compact()) throw std::runtime_error("hydia: query must be one 2-component ciphertext");
    const int dim = prm.dim;
    if (first < 0 || count < 0 || first > dim || count > dim - first) throw std::runtime_error("hydia: rotation range outside 0 .. vector_dim");
    if (count == 0) return;
    const int nl = qc.nl, nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    const size_t ce = qc.ct_elems();
    if (first == 0) HIP_CHECK(hipMemcpyAsync(out, qc.d, qc.bytes(), hipMemcpyDeviceToDevice, stream));
    const int r0 = std::max(first, 1), nr = first + count - r0;
    if (nr <= 0) return;
    build_rotptrs();
    // loop A as the review prices it: the shared ModUp, every rotation key in once (as resident: packed shadow or plain), every
    // rotated ciphertext out once; the per-rotation ModDown transforms are NOT charged (13.9 GB at 511 rotations)
    op_bytes("op:loop_a", N, (double)nd * nE,
             (double)nr * (rotptrs_packed ? (double)hk::key_packed_bytes(N, nQ, nT, prm.dnum) : (double)nd * 2 * nE * N * 8) + (double)nr * 2 * nl * N * 8);
    u64 *dig = pool.get((size_t)nd * nE * N * sizeof(u64));
    modup_digits(qc.d + (size_t)nl * N, 0, 1, nl, dig);
    ks_apply(dig, 0, nr, nl, d_rotptrs + r0, 0, qc.d, 0, qc.poly_elems(), 1, d_rotgalois + r0, d_rotginv + r0, 0, false,
             out + (size_t)(r0 - first) * ce, rotptrs_packed ? nQ : 0);
    pool.put(dig);
}
// computeSimilarity (sender_diag.cpp:12-33): all G blocks of the resident DB in one tensor-accumulate launch
Ct Context::similarity(const Ct &qc) {
    if (db_kind == 6) {
        Ct sum = similarity_bsgs_sum(qc);
        rescale(sum);
        return sum;
    }
    Ct acc = similarity_accumulate(qc);
    relin_rescale(acc);
    return acc;
}

// ------------------------------------------------------------------ baby-step / giant-step form of the mat-vec (north_star)
// With i = b + B g (B babies, NG = dim / B giants) and Rot_i = Rot_{Bg} o Rot_b:
//     sum_i Rot_i(q) . db_i  =  sum_g Rot_{Bg}( sum_b Rot_b(q) . Rot_{-Bg}(db_{b + Bg}) )
// and the enroller has rotated diagonal i by -B (i div B) in the clear (database kind 6; same ciphertext order, so the inner sums
// are loop B's own kernel on NG G "blocks" of B diagonals).  Per query B - 1 hoisted rotations instead of dim - 1; per block NG
// relinearisations and NG - 1 ordinary rotations (rotation keys B, 2B, ..: already part of the key set) instead of one
// relinearisation — the better trade while a GPU holds few blocks; B grows with the blocks (Context::auto_babies) up to B = dim,
// which is the reference's form.  The reference itself hoists every
// rotation (sender_diag.cpp:22-26); SURVEY "fact 2" allows this form as long as decrypted scores stay within 1e-4, and the oracle
// restates it (oracle/path.c hyo_compute_similarity_bsgs) so the ciphertexts are still checked bit for bit.
void Context::build_giants(int G) {
    const int B = db_babies, NG = (prm.dim + B - 1) / B, X = (NG - 1) * G;
    if (giants_valid && giants_B == B && giants_G == G) return;
    if (d_giant_keys) {  // another split or block count: the tables change size
        sync_all();
        for (void *p : {(void *)d_giant_keys, (void *)d_giant_gal, (void *)d_giant_ginv}) (void)hipFree(p);
        d_giant_keys = nullptr;
        d_giant_gal = d_giant_ginv = nullptr;
    }
    if (X <= 0) {
        giants_valid = true;
        giants_B = B;
        giants_G = G;
        return;
    }
    std::vector<const u64 *> ptrs(X, nullptr);
    std::vector<unsigned> gal(X, 1u), ginv(X, 1u);
    for (int g = 1; g < NG; g++) {
        auto it = rot_keys.find(g * B);
        if (it == rot_keys.end()) throw StateError("hydia: rotation key " + std::to_string(g * B) + " not loaded");
        const unsigned ge = (unsigned)galois_elt(g * B);
        u64 x = 1;
        for (int k = 0; k < 6; k++) x = x * (2 - (u64)ge * x);
        for (int m = 0; m < G; m++) {  // giant-major batch: the G blocks of one giant step share its key (adjacent in the launch)
            ptrs[(size_t)(g - 1) * G + m] = it->second.d;
            gal[(size_t)(g - 1) * G + m] = ge;
            ginv[(size_t)(g - 1) * G + m] = (unsigned)(x & (2ull * N - 1));
        }
    }
    HIP_CHECK(hipMalloc((void **)&d_giant_keys, sizeof(u64 *) * X));
    HIP_CHECK(hipMalloc((void **)&d_giant_gal, sizeof(unsigned) * X));
    HIP_CHECK(hipMalloc((void **)&d_giant_ginv, sizeof(unsigned) * X));
    sync_all();
    HIP_CHECK(hipMemcpy((void *)d_giant_keys, ptrs.data(), sizeof(u64 *) * X, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_giant_gal, gal.data(), sizeof(unsigned) * X, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_giant_ginv, ginv.data(), sizeof(unsigned) * X, hipMemcpyHostToDevice));
    giants_valid = true;
    giants_B = B;
    giants_G = G;
}
Ct Context::similarity_bsgs_sum(const Ct &qc) {
    if (!d_db || db_cts == 0 || db_kind != 6) throw StateError("hydia: no database resident (pre-rotated diagonal packing)");
    if (qc.nl != nQ) throw StateError("hydia: query must be a fresh (level 0) ciphertext");
    const int dim = prm.dim, B = db_babies, NG = (dim + B - 1) / B;
    if (B < 1 || dim % B) throw StateError("hydia: the resident database carries no valid baby count");
    const int G = (int)(db_cts / dim), nl = nQ, nE = nl + nP, nd = (nl + alpha - 1) / alpha;
    build_giants(G);
    // babies: rotations 0 .. B-1 of the query (loop A on B - 1 keys)
    Ct rot(this, B, 2, nl, qc.scale);
    rotate_query_range(qc, 0, B, rot.d);
    // inner sums: ciphertext t = (block*NG + g)*B + b is "diagonal b of block block*NG + g"; the accumulators come out giant-major
    // (slot g*G + block), so every later step is ONE batch over all database blocks
    Ct acc(this, G * NG, 3, nl, qc.scale * delta);
    op_bytes("op:loop_b", N, 0, (double)db_cts * (double)db_layout().ct_bytes + ((double)B * 2 + (double)G * NG * 3) * nl * N * 8);
    timer_begin("hydia_tensor");
    hk::hydia_tensor_accumulate(stream, d_mod, N, rot.d, d_db, acc.d, G * NG, B, nl, tensor_bpp, tensor_nw, db_lay, NG);
    timer_end("hydia_tensor");
    relinearize(acc);  // [NG*G][2][nl]
    // giant steps: the partial sums g >= 1 of ALL blocks go through one batched key switch, rotation key B g for slot (g, block)
    // (the automorphism rides in the ModDown epilogue); then out[block] = sum over g of slot (g, block)
    Ct out(this, G, 2, nl, acc.scale);
    if (NG > 1) {
        const int X = (NG - 1) * G;
        const size_t ce = acc.ct_elems();
        op_bytes("op:rotate", N, (double)X * (nd * nE + 2 * nP + 2 * nl), (double)(NG - 1) * nd * 2 * nE * N * 8 + 4.0 * X * nl * N * 8);
        Ct rotd(this, X, 2, nl, acc.scale);
        const u64 *c = acc.d + (size_t)G * ce;
        if (ks_fused_ok()) {
            ks_fused(c + acc.poly_elems(), ce, X, nl, nullptr, nullptr, d_giant_keys, c, ce, acc.poly_elems(), 1, d_giant_ginv, 0, false, rotd.d);
        } else {
            u64 *dig = pool.get((size_t)X * nd * nE * N * sizeof(u64));
            modup_digits(c + acc.poly_elems(), ce, X, nl, dig);
            ks_apply(dig, (size_t)nd * nE * N, X, nl, d_giant_keys, 0, c, ce, acc.poly_elems(), 1, d_giant_gal, d_giant_ginv, 0, false, rotd.d);
            pool.put(dig);
        }
        HIP_CHECK(hipMemcpyAsync(acc.d + (size_t)G * ce, rotd.d, (size_t)X * ce * sizeof(u64), hipMemcpyDeviceToDevice, stream));
    }
    op_bytes("op:add", N, 0, (double)(NG + 1) * G * 2 * nl * N * 8);
    hk::batch_sum(stream, d_mod, N, acc.d, out.d, NG, 2, nl, G, G);
    return out;
}

// ------------------------------------------------------------------ comparator
namespace {
struct Cheb {
    Context *cx;
    std::vector<Ct> T;  // T[1..8]
    std::vector<Ct> G;  // G[i] = T_{8*2^i}; G[0] is a view of T[8]
};
// 2ab - c (c == nullptr: the constant 1).  c is brought to the product's scale s_a*s_b by the integer factor
// K = round(s_a*s_b/s_c) and leaves with the tensor (d -= (K/2 mod q)*c ahead of the doubling relinearisation), so the
// subtraction joins operands of identical scale; same arithmetic as the oracle's relin, x2, -K*c, rescale.
Ct cheb_step(Context *cx, const Ct &a, const Ct &b, const Ct *c) {
    const int nl = std::min(a.nl, b.nl);
    Ct x = a.alias(nl), y = b.alias(nl);
    const double minus_one = -1.0;
    if (c) {
        Ct cv = c->alias(nl);
        return cx->mult_relin_rescale(x, y, true, nullptr, nullptr, false, &cv);
    }
    return cx->mult_relin_rescale(x, y, true, nullptr, &minus_one);
}
// a*b (relinearised, rescaled) + r with r already at the product's scale; r joins in the rescale epilogue when it has the limbs
Ct mult_add(Context *cx, const Ct &a, const Ct &b, Ct &r) {
    const int nl = std::min(a.nl, b.nl);
    Ct x = a.alias(nl), y = b.alias(nl);
    if (r.nl >= nl - 1) {
        Ct rv = r.alias(nl - 1);
        return cx->mult_relin_rescale(x, y, false, &rv, nullptr, true);
    }
    Ct o = cx->mult_relin_rescale(x, y);
    cx->drop_to(o, r.nl);
    cx->add_inplace(o, r);
    return o;
}
int leaf_nl(Cheb &ch, const double *c, int deg) {
    int nl = ch.T[1].nl;
    for (int j = 1; j <= deg; j++)
        if (c[j] != 0.0) nl = std::min(nl, ch.T[j].nl);
    return nl;
}
void cheb_split(const double *c, int deg, int g, std::vector<double> &qc, std::vector<double> &rc) {
    qc.assign(g, 0.0);
    rc.assign(c, c + g);
    qc[0] = c[g];
    for (int j = g + 1; j <= deg; j++) {
        qc[j - g] = 2.0 * c[j];
        rc[2 * g - j] -= c[j];
    }
}
int cheb_plan_nl(Cheb &ch, const double *c, int deg, int gi) {  // limbs a subtree's result will have
    while (deg > 0 && c[deg] == 0.0) deg--;
    if (deg < 8) return leaf_nl(ch, c, deg) - 1;
    const int g = 8 << gi;
    if (deg < g) return cheb_plan_nl(ch, c, deg, gi - 1);
    std::vector<double> qc, rc;
    cheb_split(c, deg, g, qc, rc);
    const int nq = cheb_plan_nl(ch, qc.data(), deg - g, gi - 1), nr = cheb_plan_nl(ch, rc.data(), g - 1, gi - 1);
    return std::min(std::min(nq, ch.G[gi].nl) - 1, nr);
}
// The Paterson-Stockmeyer tree is PLANNED on the host first (scales and limb counts do not depend on ciphertext data), so
// that all leaves over the same limbs run as ONE multi-output pass + ONE batched rescale; then the products are combined.
// sum_j c_j T_j at scale `target`: the target is pushed down the tree (quotient: target*q_l/scale(T_g), remainder: the
// product's scale) so every addition joins operands of identical scale.  A leaf encodes its constants at
// target*q_l/scale(T_j): the rescaled leaf has scale `target` exactly.
struct PNode {
    bool leaf = false;
    std::vector<double> c;  // leaf: c[0..deg]
    int deg = 0, gi = 0, nl_in = 0, nl = 0, q = -1, r = -1, group = -1, slot = -1;
    double target = 0, scale = 0;
};
int cheb_plan(Cheb &ch, std::vector<PNode> &tree, const double *c, int deg, int gi, double target) {
    Context *cx = ch.cx;
    while (deg > 0 && c[deg] == 0.0) deg--;
    if (deg < 8) {
        PNode n;
        n.leaf = true;
        n.c.assign(c, c + deg + 1);
        n.deg = deg;
        n.nl_in = leaf_nl(ch, c, deg);
        n.nl = n.nl_in - 1;
        n.target = n.scale = target;
        tree.push_back(n);
        return (int)tree.size() - 1;
    }
    const int g = 8 << gi;
    if (deg < g) return cheb_plan(ch, tree, c, deg, gi - 1, target);
    std::vector<double> qc, rc;
    cheb_split(c, deg, g, qc, rc);
    const Ct &G = ch.G[gi];
    const int nq = cheb_plan_nl(ch, qc.data(), deg - g, gi - 1);
    const int lp = std::min(nq, G.nl);
    PNode n;
    n.gi = gi;
    n.target = target;
    n.q = cheb_plan(ch, tree, qc.data(), deg - g, gi - 1, target * (double)cx->q[lp - 1] / G.scale);
    n.scale = (tree[n.q].scale * G.scale) / (double)cx->q[std::min(tree[n.q].nl, G.nl) - 1];  // what mult will report
    n.r = cheb_plan(ch, tree, rc.data(), g - 1, gi - 1, n.scale);
    n.nl = std::min(std::min(tree[n.q].nl, G.nl) - 1, tree[n.r].nl);
    tree.push_back(n);
    return (int)tree.size() - 1;
}
struct LeafGroup {
    int nl_in = 0;
    std::vector<int> nodes;
    Ct out;  // [leaf][x]
};
Ct cheb_eval(Cheb &ch, std::vector<PNode> &tree, std::vector<LeafGroup> &groups, int id, bool fork) {
    Context *cx = ch.cx;
    PNode &n = tree[id];
    if (n.leaf) {
        const Ct &big = groups[n.group].out;
        const int X = ch.T[1].X;
        Ct v = big.alias(big.nl);
        v.X = X;
        v.d = big.d + (size_t)n.slot * X * big.ct_elems();
        v.scale = n.target;
        return v;
    }
    Ct Q, R;
    if (fork) {  // quotient and remainder subtrees share nothing but the babies and giants they read
        cx->par2([&] { Q = cheb_eval(ch, tree, groups, n.q, false); }, [&] { R = cheb_eval(ch, tree, groups, n.r, false); });
    } else {
        Q = cheb_eval(ch, tree, groups, n.q, false);
        R = cheb_eval(ch, tree, groups, n.r, false);
    }
    return mult_add(cx, Q, ch.G[n.gi], R);
}
struct ChebTree {
    std::vector<PNode> nodes;
    std::vector<LeafGroup> groups;
    int root = -1;
};
// plan (host only: needs the giants' limb counts and scales, i.e. their Ct objects, not their data) + all leaves
void cheb_leaves(Cheb &ch, ChebTree &tr, const double *c, int degree, int gi, double target) {
    Context *cx = ch.cx;
    std::vector<PNode> &tree = tr.nodes;
    std::vector<LeafGroup> &groups = tr.groups;
    tr.root = cheb_plan(ch, tree, c, degree, gi, target);
    for (int i = 0; i < (int)tree.size(); i++) {
        if (!tree[i].leaf) continue;
        int g = -1;
        for (int k = 0; k < (int)groups.size(); k++)
            if (groups[k].nl_in == tree[i].nl_in) g = k;
        if (g < 0) {
            groups.emplace_back();
            g = (int)groups.size() - 1;
            groups[g].nl_in = tree[i].nl_in;
        }
        tree[i].group = g;
        tree[i].slot = (int)groups[g].nodes.size();
        groups[g].nodes.push_back(i);
    }
    for (auto &grp : groups) {
        int maxdeg = 1;  // a pure constant is 0*T_1 + c_0
        for (int id : grp.nodes) maxdeg = std::max(maxdeg, tree[id].deg);
        std::vector<Ct> views;
        for (int j = 1; j <= maxdeg; j++) views.push_back(ch.T[j].alias(grp.nl_in));
        std::vector<const Ct *> terms;
        for (auto &v : views) terms.push_back(&v);
        std::vector<std::vector<double>> coef;
        std::vector<double> c0, S;
        for (int id : grp.nodes) {
            const PNode &n = tree[id];
            std::vector<double> cj(maxdeg, 0.0);
            for (int j = 1; j <= n.deg; j++) cj[j - 1] = n.c[j];
            coef.push_back(cj);
            c0.push_back(n.c[0]);
            S.push_back(n.target * (double)cx->q[grp.nl_in - 1]);
        }
        grp.out = cx->lincomb_multi(terms, coef, c0, S);
        cx->rescale(grp.out);
    }
}
Ct cheb_combine(Cheb &ch, ChebTree &tr) {
    Ct res = cheb_eval(ch, tr.nodes, tr.groups, tr.root, true);
    if (res.view) res = ch.cx->clone(res);  // a tree that is a single leaf
    return res;
}
// interpolation of step-at-delta at the degree+1 Chebyshev nodes (what EvalChebyshevFunction derives)
std::vector<double> step_coeffs(double delta, int degree) {
    const int n = degree + 1;
    std::vector<double> f(n), c(n);
    for (int i = 0; i < n; i++) f[i] = (std::cos(M_PI * (i + 0.5) / n) >= delta) ? 1.0 : -1.0;
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += f[i] * std::cos(M_PI * j * (i + 0.5) / n);
        c[j] = s * 2.0 / n;
    }
    c[0] *= 0.5;
    return c;
}
const double F4[10] = {0.0, 315.0 / 128.0, 0.0, -420.0 / 128.0, 0.0, 378.0 / 128.0, 0.0, -180.0 / 128.0, 0.0, 35.0 / 128.0};
}  // namespace

// OpenFHEWrapper::chebyshevCompare (openFHE_wrapper.cpp:143-185) on a batch of score ciphertexts.
Ct Context::chebyshev_compare(const Ct &x, double dlt, int sign_depth) {
    if (sign_depth < 7 || sign_depth > 15) {  // :146-149
        fprintf(stderr, "Error: chebshevCompare requires a depth parameter between 7 and 15\n");
        return clone(x);
    }
    static const int DEPTH_TO_DEGREE[12] = {-1, -1, -1, 5, 13, 27, 59, 119, 247, 495, 1007, 2031};  // :153-155
    const int degree = DEPTH_TO_DEGREE[sign_depth - 4];
    std::vector<double> c = step_coeffs(dlt, degree);
    Cheb ch;
    ch.cx = this;
    ch.T.resize(9);
    ch.T[1] = x.alias(x.nl);
    const int top = std::min(degree, 8);
    // independent products run as pairs: on one lane for multi-block batches (throughput-bound), on two for a single block
    if (top >= 2) ch.T[2] = cheb_step(this, ch.T[1], ch.T[1], nullptr);
    par2([&] { if (top >= 3) ch.T[3] = cheb_step(this, ch.T[2], ch.T[1], &ch.T[1]); },
         [&] { if (top >= 4) ch.T[4] = cheb_step(this, ch.T[2], ch.T[2], nullptr); });
    par2([&] {
             if (top >= 5) ch.T[5] = cheb_step(this, ch.T[3], ch.T[2], &ch.T[1]);
             if (top >= 6) ch.T[6] = cheb_step(this, ch.T[3], ch.T[3], nullptr);
         },
         [&] {
             if (top >= 7) ch.T[7] = cheb_step(this, ch.T[4], ch.T[3], &ch.T[1]);
             if (top >= 8) ch.T[8] = cheb_step(this, ch.T[4], ch.T[4], nullptr);
         });
    int gi = -1;
    ChebTree tr;
    // the giants T_16, T_32 (a chain of squarings) next to the Paterson-Stockmeyer leaves (linear in T_1..T_7): the side piece is
    // enqueued first, so the giants' limb counts and scales are known when the tree is planned
    par2([&] { cheb_leaves(ch, tr, c.data(), degree, gi, delta); },
         [&] {
             if (degree >= 8) {
                 ch.G.push_back(ch.T[8].alias(ch.T[8].nl));
                 gi = 0;
                 while ((8 << (gi + 1)) <= degree) {
                     Ct nx = cheb_step(this, ch.G[gi], ch.G[gi], nullptr);
                     ch.G.push_back(std::move(nx));
                     gi++;
                 }
             }
         });
    Ct y = cheb_combine(ch, tr);
    tr.groups.clear();
    ch.G.clear();
    ch.T.clear();
    // f4 in depth 4: (c1 y + c3 y^3) + y^4 (c5 y + c7 y^3) + (c9 y) y^8   (openFHE_wrapper.cpp:158-169, :179)
    Ct y2 = mult(y, y), y3, y4, y8;
    par2([&] { y3 = mult(y2, y); }, [&] { y4 = mult(y2, y2); });
    const int nl = y3.nl;
    Ct yd = y.alias(nl);
    // v at scale Delta; a = v*y^4 fixes the scale the other two summands are steered to.  u is evaluated before a and b
    // and the sum a + b + u is formed in the rescale epilogues of the two products.  y^8 (one more squaring) runs beside the three
    // linear pieces; its limb count and scale follow from y^4's on the host
    const int y8_nl = y4.nl - 1;
    const double y8_scale = (y4.scale * y4.scale) / (double)q[y4.nl - 1];
    Ct v, u, w;
    double a_scale = 0;
    par2([&] {
             v = lincomb({&yd, &y3}, {F4[5], F4[7]}, 0.0, delta * (double)q[nl - 1]);
             rescale(v);
             v.scale = delta;
             a_scale = (v.scale * y4.scale) / (double)q[std::min(v.nl, y4.nl) - 1];
             u = lincomb({&yd, &y3}, {F4[1], F4[3]}, 0.0, a_scale * (double)q[nl - 1]);
             rescale(u);
             u.scale = a_scale;
             const int lb = std::min(y.nl - 1, y8_nl);
             const double wt = a_scale * (double)q[lb - 1] / y8_scale;
             w = lincomb({&y}, {F4[9]}, 0.0, wt * (double)q[y.nl - 1]);
             rescale(w);
             w.scale = wt;
         },
         [&] { y8 = mult(y4, y4); });
    if (y8.nl != y8_nl || y8.scale != y8_scale) throw std::runtime_error("hydia: comparator scale plan out of step");
    Ct au = mult_add(this, v, y4, u);
    Ct a = mult_add(this, w, y8, au);
    a.scale = a_scale;
    add_const(a, 1.0);  // :182
    return a;
}

// relinearise + rescale + compare of a degree-2 batch, split over the comparator LANES: each lane runs the whole chain on its
// share of the blocks on its own stream, so the latency- and multiplier-bound kernels of one lane fill the HBM idle time of
// another's (loop B itself stays alone on the main stream).  Same arithmetic per ciphertext whatever the split.
Ct Context::relin_compare_lanes(Ct &acc, double dlt, int sign_depth) {
    const int G = acc.X, L = std::min(nlanes, G);
    const bool relin = acc.npoly == 3;  // the BSGS mat-vec hands over relinearised sums: only the rescale is left
    if (L <= 1) {
        if (relin) relin_rescale(acc);
        else rescale(acc);
        return chebyshev_compare(acc, dlt, sign_depth);
    }
    while ((int)lane_ev.size() < nlanes + 1) {  // created once, re-recorded per call
        hipEvent_t e;
        HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        lane_ev.push_back(e);
    }
    std::vector<hipEvent_t> &ev = lane_ev;
    struct LaneGuard {  // an exception inside a lane (out of memory, ...) must not leave the context pointing at a side lane
        Context *c;
        ~LaneGuard() {
            c->set_lane(0);
            c->side_lane_free = true;
            if (std::uncaught_exceptions() > 0)  // error path: let every lane drain before buffers shared across lanes are released
                for (auto st : c->lane_stream) (void)hipStreamSynchronize(st);
        }
    } guard{this};
    side_lane_free = false;  // the lanes are taken by the block split: products inside run in sequence
    HIP_CHECK(hipEventRecord(ev[L], stream));  // acc is ready (everything enqueued on the main stream so far)
    std::vector<Ct> res(L);
    Ct out;
    for (int k = 0; k < L; k++) {
        const int g0 = (int)((long)G * k / L), g1 = (int)((long)G * (k + 1) / L);
        set_lane(k);
        if (k > 0) HIP_CHECK(hipStreamWaitEvent(stream, ev[L], 0));
        Ct part = acc.alias(acc.nl);
        part.X = g1 - g0;
        part.d = acc.d + (size_t)g0 * acc.ct_elems();
        if (relin) relin_rescale(part);
        else rescale(part);
        res[k] = chebyshev_compare(part, dlt, sign_depth);
        if (k == 0) {
            out = Ct(this, G, res[0].npoly, res[0].nl, res[0].scale);
            HIP_CHECK(hipEventRecord(ev[0], stream));  // `out` may recycle main-lane memory: other lanes write it only after this
        } else {
            HIP_CHECK(hipStreamWaitEvent(stream, ev[0], 0));
        }
        Ct rk = res[k].compact() ? res[k].alias(res[k].nl) : clone(res[k]);
        HIP_CHECK(hipMemcpyAsync(out.d + (size_t)g0 * out.ct_elems(), rk.d, rk.bytes(), hipMemcpyDeviceToDevice, stream));
        if (k > 0) HIP_CHECK(hipEventRecord(ev[k], stream));
    }
    set_lane(0);
    for (int k = 1; k < L; k++) HIP_CHECK(hipStreamWaitEvent(stream, ev[k], 0));  // main stream: results complete, acc released
    res.clear();  // each lane's buffers return to that lane's free list (Pool::put)
    return out;
}
// the degree-2 accumulators of loop B for all resident blocks (computeSimilarity without its relinearise / rescale tail)
Ct Context::similarity_accumulate(const Ct &qc) {
    if (!d_db || db_cts == 0 || db_kind != 5) throw StateError("hydia: no database resident (diagonal packing)");
    if (qc.nl != nQ) throw StateError("hydia: query must be a fresh (level 0) ciphertext");
    Ct rot = rotate_query(qc);
    return similarity_accumulate_rot(rot);
}
Ct Context::similarity_accumulate_rot(const Ct &rot) {
    if (!d_db || db_cts == 0 || db_kind != 5) throw StateError("hydia: no database resident (diagonal packing)");
    const int dim = prm.dim;
    if (rot.X != dim || rot.npoly != 2 || rot.nl != nQ || !rot.compact())
        throw StateError("hydia: rotations must be vector_dim fresh 2-component ciphertexts");
    const int G = (int)(db_cts / dim);
    Ct acc(this, G, 3, nQ, rot.scale * delta);
    op_bytes("op:loop_b", N, 0, (double)db_cts * (double)db_layout().ct_bytes + ((double)dim * 2 + (double)G * 3) * nQ * N * 8);
    timer_begin("hydia_tensor");
    hk::hydia_tensor_accumulate(stream, d_mod, N, rot.d, d_db, acc.d, G, dim, nQ, tensor_bpp, tensor_nw, db_lay);
    timer_end("hydia_tensor");
    return acc;
}
Ct Context::similarity_rot(const Ct &rot) {
    Ct acc = similarity_accumulate_rot(rot);
    relin_rescale(acc);
    return acc;
}
Ct Context::index_scenario_rot(const Ct &rot) {
    Ct acc = similarity_accumulate_rot(rot);
    return relin_compare_lanes(acc, 0.44, 10);
}
// indexScenario (sender_diag.cpp:52-63): loop A, loop B, then the per-block tails on the comparator lanes
Ct Context::index_scenario(const Ct &qc) {
    Ct acc = db_kind == 6 ? similarity_bsgs_sum(qc) : similarity_accumulate(qc);  // 2 components (relinearised) : 3
    return relin_compare_lanes(acc, 0.44 /* MATCH_THRESHOLD, include/config.h:9 */, 10 /* COMP_DEPTH, :14 */);
}
// membershipScenario (sender_diag.cpp:35-50): EvalAddManyInPlace over blocks, then EvalSum over all slots
Ct Context::membership_scenario(const Ct &qc) {
    Ct s = index_scenario(qc);
    return sum_and_evalsum(s);
}
// EvalAddManyInPlace (sender_diag.cpp:46): the batch summed into its first element's shape, one modular add per block
Ct Context::add_many(const Ct &s) {
    if (s.compact() && s.X > 2) {  // one launch: exact 128-bit sums over the batch, one reduction (the same residues as X - 1 modular additions)
        Ct m(this, 1, s.npoly, s.nl, s.scale);
        op_bytes("op:add", N, 0, (double)(s.X + 1) * s.npoly * s.nl * N * 8);
        hk::batch_sum(stream, d_mod, N, s.d, m.d, s.X, s.npoly, s.nl);
        return m;
    }
    Ct first = s.alias(s.nl);
    first.X = 1;
    Ct m = clone(first);
    for (int g = 1; g < s.X; g++) {
        Ct v = s.alias(s.nl);
        v.X = 1;
        v.d = s.d + (size_t)g * s.ct_elems();
        add_inplace(m, v);
    }
    return m;
}
// EvalSum(ct, batchSize) (sender_diag.cpp:47): log2(slots) rotate-and-add steps
Ct Context::eval_sum(const Ct &a) {
    Ct m = clone(a);
    for (int r = 1; r < slots; r <<= 1) {
        Ct t = rotate(m, r);
        add_inplace(m, t);
    }
    return m;
}
Ct Context::sum_and_evalsum(const Ct &s) {
    Ct m = add_many(s);
    for (int r = 1; r < slots; r <<= 1) {
        Ct t = rotate(m, r);
        add_inplace(m, t);
    }
    return m;
}
// Cross-shard membership reduction: partial sums of the shards are added as plain 64-bit integers (what an RCCL
// all-reduce(SUM) on int64 does: at most 16 residues below 2^60 cannot overflow) and reduced once.
void Context::add_raw_inplace(Ct &a, const u64 *other) {
    hk::add_raw(stream, d_mod, N, a.d, other, a.d, a.X * a.npoly, sel_q(a.nl), a.lstride, a.nl, a.lstride);
}
void Context::mod_reduce_inplace(Ct &a) { hk::mod_reduce(stream, d_mod, N, a.d, a.X * a.npoly, sel_q(a.nl), a.lstride); }

// ------------------------------------------------------------------ HERS sender (approach 4)
// HersSender::computeSimilarity / computeSimilarityHelper (/root/reference/src/sender/sender_hers.cpp:13-87): per block the
// dim products EvalMultNoRelin(query_i, db[m][i]) are relinearised and rescaled ONE BY ONE ("to match HERS paper approach",
// :73-75) and then summed — run here as one batch of dim ciphertexts per block.
Ct Context::hers_similarity(const Ct &qc) {
    if (!d_db || db_cts == 0 || db_kind != 4) throw StateError("hydia: no database resident (HERS column packing)");
    const int dim = prm.dim;
    if (qc.X != dim || qc.npoly != 2 || qc.nl != nQ || !qc.compact())
        throw std::runtime_error("hydia: HERS query must be vector_dim fresh ciphertexts");
    const int G = (int)(db_cts / dim);
    Ct out(this, G, 2, nQ - 1, qc.scale * delta / (double)q[nQ - 1]);
    Ct dbp(this, dim, 2, nQ, delta);
    for (int m = 0; m < G; m++) {
        db_fetch((size_t)m * dim, dbp.d, dim);
        Ct prod = mult_norelin(qc, dbp);
        relin_rescale(prod);
        hk::batch_sum(stream, d_mod, N, prod.d, out.d + (size_t)m * out.ct_elems(), dim, 2, prod.nl);
    }
    return out;
}
Ct Context::hers_index_scenario(const Ct &qc) {  // sender_hers.cpp:28-40
    Ct s = hers_similarity(qc);
    return chebyshev_compare(s, 0.44, 10);
}
Ct Context::hers_membership_scenario(const Ct &qc) {  // sender_hers.cpp:43-58
    Ct s = hers_index_scenario(qc);
    return sum_and_evalsum(s);
}

}  // namespace hydia

// image_matching_amd/csrc/ntt15.hip — negacyclic NTT / INTT for the production ring N = 2^15 on gfx950.
//
// N = 128 rows x 256 columns.  Forward = pass 1 (stages 0-6, row strides 64..1: a workgroup owns 32 adjacent columns of
// all 128 rows) then pass 2 (stages 7-14 inside 256-coefficient blocks: a workgroup owns 2048 consecutive
// coefficients); the inverse runs pass 2' then pass 1' with Gentleman-Sande butterflies.  Butterflies run in REGISTERS
// in radix-16 / radix-8 / radix-4 groups; LDS is used only to transpose between register phases (one exchange in
// pass 1, two in pass 2, padded to be bank-conflict free for ds_read/write_b64).
//
// Two arithmetic back ends, chosen per limb (workgroup-uniform) and both EXACT, so results are bit-identical:
//   IntA  64-bit integers, Shoup multiplication, Harvey lazy reduction ([0,4q) forward, [0,2q) inverse).  Used for the
//         60-bit primes (q_0 and the special primes P).  ~10 32x32 multiplies + carry chains per butterfly.
//   FpA   the 45/46-bit scaling primes fit a double's 53-bit mantissa: residues are kept as exact-integer doubles and
//         a*w mod q is  h = a*w; l = fma(a,w,-h); c = rint(a*(w/q)); r = fma(-c,q,h) + l  — six FP64 instructions, no
//         compares, no carries; the forward transform needs no reduction at all (growth <= 0.75q per stage, 15 stages
//         stay below 2^50), the inverse re-centres once per register phase.  gfx950 issues FP64 FMA at least as fast
//         as its 32x32 integer multiply, so this path is ~3x cheaper per butterfly.
// Between the two passes FpA limbs travel as raw doubles in the (uint64) buffer; every limb leaves as canonical uint64.
// Pass-1 phase-A twiddles are workgroup-uniform; the remaining pass-1 twiddles sit in 2 KiB of LDS; pass-2 twiddles are
// 16-byte pair loads shared by the TWO polynomials a workgroup transforms together.
#include "kernels.h"
#include "ntt_arith.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------ fused prologues
// canonical coefficient-form value of element `idx` (inside the limb-poly) of target limb `slot` for polynomial x
template <int LD>
DEV u64 p1_load(const NttLoad &ld, const ModC *__restrict__ mod, const ModC &M, const u64 *s, int x, int slot, size_t idx) {
    if (LD == 0) return s[idx];
    // LD == 2: rescale spread: centred residue of the dropped limb's coefficient
    const u64 ql = mod[ld.l].q, v = ld.y[(size_t)x * ld.y_outer + idx];
    return v > (ql >> 1) ? negmod(reduce64(ql - v, M), M.q) : reduce64(v, M);
}

// pass 1's operand in the arithmetic's own representation.  Rescale spread (LD 2) into an FP64 limb needs no reduction at all: the
// centred residue of the dropped 45-bit limb is an exact double well inside the transform's headroom, and the transform is linear
template <class A, int LD>
DEV typename A::T p1_load_as(const A &ar, const NttLoad &ld, const ModC *__restrict__ mod, const ModC &M, const u64 *s, int x, int slot,
                             size_t idx, std::integral_constant<int, LD>) {
    return ar.from_canon(p1_load<LD>(ld, mod, M, s, x, slot, idx));
}
DEV double p1_load_as(const FpA &ar, const NttLoad &ld, const ModC *__restrict__ mod, const ModC &M, const u64 *, int x, int, size_t idx,
                      std::integral_constant<int, 2>) {
    const u64 ql = mod[ld.l].q, v = ld.y[(size_t)x * ld.y_outer + idx];
    if (ql >> 50) {  // a 60-bit dropped limb: reduce first
        const u64 r = v > (ql >> 1) ? negmod(reduce64(ql - v, M), M.q) : reduce64(v, M);
        return ar.from_canon(r);
    }
    const double dv = FpA::u2d(v);
    return v > (ql >> 1) ? dv - FpA::u2d(ql) : dv;
}

// ------------------------------------------------------------------------------------------------ pass 1 (strided)
template <class A, bool INV, int LD>
DEV void p1_body(const A ar, const ulonglong2 *__restrict__ tw, const u64 *s, u64 *d, u64 *lds, const ulonglong2 *ltw, int t,
                 u64 sc, u64 scs, const NttLoad &ld, const ModC *__restrict__ mod, const ModC &M, int x, int slot, int c0) {
    typedef typename A::T T;
    const int col = t & 31, g = t >> 5;
    T v[16];
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[k] = p1_load_as(ar, ld, mod, M, LD == 0 ? s : nullptr, x, slot, (size_t)(g + 8 * k) * 256 + (LD == 0 ? 0 : c0) + col,
                              std::integral_constant<int, LD>());
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const int h = 8 >> st;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!(k & h)) ar.ct(v[k], v[k + h], A::tw(tw[(1 << st) + (k >> (4 - st))]));
            if (st == 2) {  // lazy 60-bit limbs: three stages between folds
#pragma unroll
                for (int k = 0; k < 16; k++) ar.fwd_fold(v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) lds[(g + 8 * k) * 32 + col] = A::to_bits(v[k]);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int h = g + 8 * hh;
            T w[8];
#pragma unroll
            for (int l = 0; l < 8; l++) w[l] = A::from_bits(lds[(8 * h + l) * 32 + col]);
            {
                const typename A::TW W = A::tw(ltw[16 + h]);
#pragma unroll
                for (int l = 0; l < 4; l++) ar.ct(w[l], w[l + 4], W);
            }
#pragma unroll
            for (int l = 0; l < 8; l++)
                if (!(l & 2)) ar.ct(w[l], w[l + 2], A::tw(ltw[32 + 2 * h + (l >> 2)]));
#pragma unroll
            for (int l = 0; l < 8; l++) ar.fwd_fold(w[l]);
#pragma unroll
            for (int l = 0; l < 8; l += 2) ar.ct(w[l], w[l + 1], A::tw(ltw[64 + 4 * h + (l >> 1)]));
#pragma unroll
            for (int l = 0; l < 8; l++) d[(size_t)(8 * h + l) * 256 + col] = A::to_bits(w[l]);  // raw: pass 2 finishes
        }
    } else {
        __syncthreads();  // ltw
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int h = g + 8 * hh;
            T w[8];
#pragma unroll
            for (int l = 0; l < 8; l++) w[l] = A::from_bits(s[(size_t)(8 * h + l) * 256 + col]);  // raw from pass 2'
#pragma unroll
            for (int l = 0; l < 8; l += 2) ar.gs(w[l], w[l + 1], A::tw(ltw[64 + 4 * h + (l >> 1)]));
#pragma unroll
            for (int l = 0; l < 8; l++)
                if (!(l & 2)) ar.gs(w[l], w[l + 2], A::tw(ltw[32 + 2 * h + (l >> 2)]));
            {
                const typename A::TW W = A::tw(ltw[16 + h]);
#pragma unroll
                for (int l = 0; l < 4; l++) ar.gs(w[l], w[l + 4], W);
            }
#pragma unroll
            for (int l = 0; l < 8; l++) {
                ar.recentre(w[l]);
                lds[(8 * h + l) * 32 + col] = A::to_bits(w[l]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = A::from_bits(lds[(g + 8 * k) * 32 + col]);
#pragma unroll
        for (int st = 3; st >= 0; st--) {
            const int h = 8 >> st;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!(k & h)) ar.gs(v[k], v[k + h], A::tw(tw[(1 << st) + (k >> (4 - st))]));
            if (st == 2) {
#pragma unroll
                for (int k = 0; k < 16; k++) ar.mid(v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) d[(size_t)(g + 8 * k) * 256 + col] = ar.fin_inv(v[k], sc, scs);
    }
}

// grid (8 column tiles, X*sel.n), 256 threads: col = t&31, g = t>>5.  Phase A rows g+8k (k<16), phase B rows 8h+l.
template <bool INV, int LD>
__global__ __launch_bounds__(256) void k_ntt15_p1(NttTables T, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so,
                                                  size_t dso, LimbSel sel, int slot0, int nsl, ScaleSel scale, NttLoad ld) {
    constexpr int N = 32768;
    __shared__ u64 lds[128 * 32];
    __shared__ ulonglong2 ltw[128];
    const int y = blockIdx.y, x = y / nsl, slot = slot0 + (y - x * nsl), m = sel.mod[slot];  // slots [slot0, slot0 + nsl) of sel
    const ModC M = T.mod[m];
    const bool fp = (T.fp_mask >> m) & 1u;  // the host's copy of the rule "FP64 tables present and q < 2^47": a kernel argument, so neither the
                                             // branch nor the twiddle addresses wait for the modulus constants to arrive
    const ulonglong2 *__restrict__ tw = (fp ? (INV ? T.itwf : T.twf) : (INV ? T.itwp : T.twp)) + (size_t)m * N;
    const int c0 = blockIdx.x * 32;
    const u64 *s = src + (size_t)x * so + (size_t)slot * N + c0;
    u64 *d = dst + (size_t)x * dso + (size_t)slot * N + c0;
    const int t = threadIdx.x;
    if (t < 128) ltw[t] = tw[t];
    if (fp) p1_body<FpA, INV, LD>(FpA(M), tw, s, d, lds, ltw, t, scale.s[slot], scale.s_sh[slot], ld, T.mod, M, x, slot, c0);
    else if ((T.pm_mask >> m) & 1u) p1_body<IntP, INV, LD>(IntP(M), tw, s, d, lds, ltw, t, scale.s[slot], scale.s_sh[slot], ld, T.mod, M, x, slot, c0);
    else p1_body<IntA, INV, LD>(IntA(M), tw, s, d, lds, ltw, t, scale.s[slot], scale.s_sh[slot], ld, T.mod, M, x, slot, c0);
}

// ------------------------------------------------------------------------------------------------ fused epilogues
DEV unsigned perm_idx(unsigned c, unsigned g) {  // evaluation-form automorphism index map for N = 2^15
    const unsigned e = ((2u * (__brev(c) >> 17) + 1u) * g) & 65535u;
    return __brev((e - 1u) >> 1) >> 17;
}
// v = canonical evaluation-form value of element idx of limb j (= slot) of polynomial xp
template <int ST>
DEV void p2_store(const NttStore &st, const ModC &M, u64 *d, int xp, int j, unsigned idx, u64 v) {
    constexpr size_t N = 32768;
    if (ST == 0) {
        d[idx] = v;
        return;
    }
    const u64 q = M.q;
    const u64 iv = st.in[((size_t)xp * st.in_ls + j) * N + idx];
    u64 r = mulmod_shoup(submod(iv, v, q), st.mul.s[j], st.mul.s_sh[j], q);
    if (ST == 1) {
        const int x = xp >> 1, p = xp & 1;
        if (st.addend && p < st.add_polys) r = addmod(r, st.addend[(size_t)x * st.add_x + (size_t)p * st.add_p + (size_t)j * N + idx], q);
        if (st.dbl) r = addmod(r, r, q);
        unsigned o = idx;
        if (st.ginv) {
            const unsigned g = st.ginv[st.same_g ? 0 : x];
            if (g != 1u) o = perm_idx(idx, g);
        }
        st.out[((size_t)xp * st.nl + j) * N + o] = r;
    } else {
        if (st.sub) r = submod(r, st.sub[((size_t)xp * st.sub_ls + j) * N + idx], q);
        if (st.has_addc && (xp % st.npoly) == 0) r = addmod(r, st.addc[j], q);
        st.out[((size_t)xp * st.nl + j) * N + idx] = r;
    }
}

// four consecutive coefficients idx..idx+3 of limb j of polynomial xp: operands prefetched BEFORE the last LDS exchange
// (the loads overlap the barrier and the last two butterfly stages), results stored 16 bytes at a time unless the
// automorphism scatters them
struct P2Pre {
    ulonglong2 in0, in1, ex0, ex1;  // `in` operand; addend (modes 1, 3) or subtrahend (mode 2)
    ulonglong2 sb0, sb1;            // subtrahend of mode 3
    bool has_ex, has_sb;
};
// operands of a fused product (ProdSrc) for four consecutive coefficients of one limb of one ciphertext: both polynomials of a and b
struct P2Prod {
    u64 a0[4], a1[4], b0[4], b1[4];
    DEV static void ld4(const u64 *p, u64 (&o)[4]) {
        const ulonglong2 u = *reinterpret_cast<const ulonglong2 *>(p), v = *reinterpret_cast<const ulonglong2 *>(p + 2);
        o[0] = u.x; o[1] = u.y; o[2] = v.x; o[3] = v.y;
    }
    DEV void load(const ProdSrc &ps, int x, int j, unsigned idx) {
        const u64 *pa = ps.a + (size_t)x * ps.a_x + (size_t)j * 32768 + idx, *pb = ps.b + (size_t)x * ps.b_x + (size_t)j * 32768 + idx;
        ld4(pa, a0);
        ld4(pa + ps.a_p, a1);
        ld4(pb, b0);
        ld4(pb + ps.b_p, b1);
    }
    // d_p of coefficient k as a small exact double (FP64 limbs) / as the canonical residue (any limb); p and k are compile-time in
    // the unrolled callers
    DEV double dp_fp(const FpA &ar, int p, int k) const {
        if (p == 0) return ar.mulmod2(FpA::u2d(a0[k]), FpA::u2d(b0[k]));
        return ar.mulmod2(FpA::u2d(a0[k]), FpA::u2d(b1[k])) + ar.mulmod2(FpA::u2d(a1[k]), FpA::u2d(b0[k]));
    }
    DEV u64 dp_int(const ModC &M, int p, int k) const {
        if (p == 0) return mulmod(a0[k], b0[k], M);
        return reduce128k((u128)a0[k] * b1[k] + (u128)a1[k] * b0[k], M);
    }
    // the subtrahend's four coefficients of polynomial p (a Chebyshev step's c): loaded where they are used
    DEV static void load_c(const ProdSrc &ps, int x, int p, int j, unsigned idx, u64 (&cv)[4]) {
        ld4(ps.c + (size_t)x * ps.c_x + (size_t)p * ps.c_p + (size_t)j * 32768 + idx, cv);
    }
};
// canonical a * b mod q in the arithmetic of the limb
DEV u64 prod_canon(const FpA &ar, const ModC &, u64 a, u64 b) { return ar.fin_fwd(ar.mulmod2(FpA::u2d(a), FpA::u2d(b))); }
DEV u64 prod_canon(const IntP &, const ModC &M, u64 a, u64 b) { return mulmod(a, b, M); }
DEV u64 prod_canon(const IntA &, const ModC &M, u64 a, u64 b) { return mulmod(a, b, M); }
// mode 5: the accumulator value of 4 consecutive coefficients of limb j, key polynomial p, rotation x — formed here instead of
// being read back: nd lazy 128-bit products per coefficient, one reduction
// The operands of one call: for every digit two pairs of digit residues and two pairs of key residues.  ALL of them are requested
// before the first product (digits beyond nd re-read the last one and are not accumulated), through a branch-free sequence: at a join
// the compiler would wait for every outstanding load, which is what made this kernel latency-bound (wait_any 0.64).
constexpr int LA_MAXD = 3;  // digits fetched up front (dnum = 3); more digits take the serial tail loop
template <bool SIX>
struct LoopAOperands {
    ulonglong2 v0[LA_MAXD], v1[LA_MAXD];
    DbRaw<SIX> k0[LA_MAXD], k1[LA_MAXD];
    DEV void fetch(const LoopAIp &la, const unsigned char *kp, size_t set_bytes, int jd, unsigned idx) {
        constexpr int N = 32768;
#pragma unroll
        for (int d = 0; d < LA_MAXD; d++) {
            const int dd = d < la.nd ? d : la.nd - 1;
            const u64 *dg = la.dig + ((size_t)dd * la.dig_rows + jd) * N + idx;
            v0[d] = *reinterpret_cast<const ulonglong2 *>(dg);
            v1[d] = *reinterpret_cast<const ulonglong2 *>(dg + 2);
            const unsigned char *kd = kp + (size_t)(2 * dd) * set_bytes;
            k0[d].template load<false>(kd);
            k1[d].template load<false>(kd + (SIX ? 12 : 16));
        }
    }
};
// RED: how the 128-bit sums leave — 0 Barrett (any modulus), 1 / 2 pseudo-Mersenne folds (IntP): canonical / lazy below 3.07 2^60
// (what the inverse transform's first two stages take: ntt_arith.h)
template <bool SIX, int RED>
DEV void loop_a_ip_int(const LoopAIp &la, const ModC &M, const unsigned char *kp, size_t set_bytes, int jd, unsigned idx, ulonglong2 &o0,
                       ulonglong2 &o1) {
    constexpr int N = 32768;
    LoopAOperands<SIX> op;
    op.fetch(la, kp, set_bytes, jd, idx);
    u128 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int d = 0; d < LA_MAXD; d++) {
        const ulonglong2 k0 = op.k0[d].get(), k1 = op.k1[d].get();
        const u64 on = d < la.nd ? ~0ull : 0ull;  // a mask, not a branch
        a0 += (u128)(op.v0[d].x & on) * k0.x;
        a1 += (u128)(op.v0[d].y & on) * k0.y;
        a2 += (u128)(op.v1[d].x & on) * k1.x;
        a3 += (u128)(op.v1[d].y & on) * k1.y;
    }
    for (int d = LA_MAXD; d < la.nd; d++) {
        const u64 *dg = la.dig + ((size_t)d * la.dig_rows + jd) * N + idx;
        const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(dg), v1 = *reinterpret_cast<const ulonglong2 *>(dg + 2);
        const unsigned char *kd = kp + (size_t)(2 * d) * set_bytes;
        const ulonglong2 k0 = db_load2<SIX, false>(kd), k1 = db_load2<SIX, false>(kd + (SIX ? 12 : 16));
        a0 += (u128)v0.x * k0.x;
        a1 += (u128)v0.y * k0.y;
        a2 += (u128)v1.x * k1.x;
        a3 += (u128)v1.y * k1.y;
    }
    if (RED == 0) {
        o0 = make_ulonglong2(reduce_lazy(a0, M, la.nd), reduce_lazy(a1, M, la.nd));
        o1 = make_ulonglong2(reduce_lazy(a2, M, la.nd), reduce_lazy(a3, M, la.nd));
    } else {
        const IntP ar(M);
        o0 = RED == 1 ? make_ulonglong2(ar.canon128(a0), ar.canon128(a1)) : make_ulonglong2(ar.fold128(a0), ar.fold128(a1));
        o1 = RED == 1 ? make_ulonglong2(ar.canon128(a2), ar.canon128(a3)) : make_ulonglong2(ar.fold128(a2), ar.fold128(a3));
    }
}
template <bool SIX>
DEV void loop_a_ip_fp(const LoopAIp &la, const FpA &ar, const unsigned char *kp, size_t set_bytes, int jd, unsigned idx, ulonglong2 &o0,
                      ulonglong2 &o1) {
    constexpr int N = 32768;
    LoopAOperands<SIX> op;
    op.fetch(la, kp, set_bytes, jd, idx);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int d = 0; d < LA_MAXD; d++) {
        const ulonglong2 k0 = op.k0[d].get(), k1 = op.k1[d].get();
        const u64 on = d < la.nd ? ~0ull : 0ull;
        a0 += ar.mulmod2(FpA::u2d(op.v0[d].x & on), FpA::u2d(k0.x));
        a1 += ar.mulmod2(FpA::u2d(op.v0[d].y & on), FpA::u2d(k0.y));
        a2 += ar.mulmod2(FpA::u2d(op.v1[d].x & on), FpA::u2d(k1.x));
        a3 += ar.mulmod2(FpA::u2d(op.v1[d].y & on), FpA::u2d(k1.y));
    }
    for (int d = LA_MAXD; d < la.nd; d++) {
        const u64 *dg = la.dig + ((size_t)d * la.dig_rows + jd) * N + idx;
        const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(dg), v1 = *reinterpret_cast<const ulonglong2 *>(dg + 2);
        const unsigned char *kd = kp + (size_t)(2 * d) * set_bytes;
        const ulonglong2 k0 = db_load2<SIX, false>(kd), k1 = db_load2<SIX, false>(kd + (SIX ? 12 : 16));
        a0 += ar.mulmod2(FpA::u2d(v0.x), FpA::u2d(k0.x));
        a1 += ar.mulmod2(FpA::u2d(v0.y), FpA::u2d(k0.y));
        a2 += ar.mulmod2(FpA::u2d(v1.x), FpA::u2d(k1.x));
        a3 += ar.mulmod2(FpA::u2d(v1.y), FpA::u2d(k1.y));
    }
    if (la.raw_fp) {  // the epilogue stays on the FP64 pipe (p2_finish5_fp): the exact small sums as they are, |a| < 3 x 0.55 q
        o0 = make_ulonglong2(FpA::to_bits(a0), FpA::to_bits(a1));
        o1 = make_ulonglong2(FpA::to_bits(a2), FpA::to_bits(a3));
        return;
    }
    o0 = make_ulonglong2(ar.fin_fwd(a0), ar.fin_fwd(a1));
    o1 = make_ulonglong2(ar.fin_fwd(a2), ar.fin_fwd(a3));
}
template <int RED = 0>
DEV void loop_a_inner_product(const LoopAIp &la, const ModC &M, int x, int p, int slot, unsigned idx, ulonglong2 &o0, ulonglong2 &o1) {
    constexpr int N = 32768;
    const unsigned char *key = reinterpret_cast<const unsigned char *>(la.keys[x]);
    const int j = la.key_row0 + slot, jd = la.dig_row0 + slot;
    const bool pk = la.packed_nQ > 0, six = pk && j > 0 && j < la.packed_nQ;
    const size_t set_bytes = pk ? key_set_bytes(N, la.packed_nQ, la.nT) : (size_t)la.nT * N * 8;
    const unsigned char *kp = key + (pk ? key_limb_offset(N, la.packed_nQ, j) : (size_t)j * N * 8) + (size_t)idx * (six ? 6 : 8) + (size_t)p * set_bytes;
    if (six) loop_a_ip_int<true, RED>(la, M, kp, set_bytes, jd, idx, o0, o1);
    else loop_a_ip_int<false, RED>(la, M, kp, set_bytes, jd, idx, o0, o1);
}
DEV void loop_a_inner_product_fp(const LoopAIp &la, const FpA &ar, int x, int p, int slot, unsigned idx, ulonglong2 &o0, ulonglong2 &o1) {
    constexpr int N = 32768;
    const unsigned char *key = reinterpret_cast<const unsigned char *>(la.keys[x]);
    const int j = la.key_row0 + slot, jd = la.dig_row0 + slot;
    const bool pk = la.packed_nQ > 0, six = pk && j > 0 && j < la.packed_nQ;
    const size_t set_bytes = pk ? key_set_bytes(N, la.packed_nQ, la.nT) : (size_t)la.nT * N * 8;
    const unsigned char *kp = key + (pk ? key_limb_offset(N, la.packed_nQ, j) : (size_t)j * N * 8) + (size_t)idx * (six ? 6 : 8) + (size_t)p * set_bytes;
    if (six) loop_a_ip_fp<true>(la, ar, kp, set_bytes, jd, idx, o0, o1);
    else loop_a_ip_fp<false>(la, ar, kp, set_bytes, jd, idx, o0, o1);
}
// LAZY: the sums feed an inverse transform (any representative its first stages take), not an epilogue (canonical)
template <bool LAZY>
DEV void loop_a_inner_product(const LoopAIp &la, const IntA &, const ModC &M, int x, int p, int j, unsigned idx, ulonglong2 &o0, ulonglong2 &o1) {
    loop_a_inner_product<0>(la, M, x, p, j, idx, o0, o1);
}
template <bool LAZY>
DEV void loop_a_inner_product(const LoopAIp &la, const IntP &, const ModC &M, int x, int p, int j, unsigned idx, ulonglong2 &o0, ulonglong2 &o1) {
    loop_a_inner_product<LAZY ? 2 : 1>(la, M, x, p, j, idx, o0, o1);
}
template <bool LAZY>
DEV void loop_a_inner_product(const LoopAIp &la, const FpA &ar, const ModC &M, int x, int p, int j, unsigned idx, ulonglong2 &o0, ulonglong2 &o1) {
    if (la.fp) loop_a_inner_product_fp(la, ar, x, p, j, idx, o0, o1);
    else loop_a_inner_product<0>(la, M, x, p, j, idx, o0, o1);
}
// accumulator of the key-switching inner product fused into pass 2 (mode 4): 128-bit lazy integer sums for the 60-bit primes,
// exactly reduced FP64 products for the primes below 2^47 (same canonical result)
template <class A>
struct IpAcc;
template <>
struct IpAcc<IntA> {
    typedef u64 V;
    u128 s = 0;
    DEV static V prep(const IntA &ar, u64 c) { return ar.fin_fwd(c); }
    DEV static V canon(u64 v) { return v; }
    DEV void mac(const IntA &, V v, u64 key) { s += (u128)v * key; }
    DEV u64 fin(const IntA &, const ModC &M, int terms) const { return reduce_lazy(s, M, terms); }
    DEV u64 fin_lazy(const IntA &ar, const ModC &M, int terms) const { return fin(ar, M, terms); }
};
template <>
struct IpAcc<IntP> {
    typedef u64 V;
    u128 s = 0;
    // round 4: the transform's lazy output (below 9q) goes into the products as it is — at most four products below 2^124 — and the
    // sum is folded (2^64 = 16c mod q) instead of Barrett-reduced: fin canonical, fin_lazy below 3.07 2^60 for the inverse transform
    DEV static V prep(const IntP &, u64 c) { return c; }
    DEV static V canon(u64 v) { return v; }
    DEV void mac(const IntP &, V v, u64 key) { s += (u128)v * key; }
    DEV u64 fin(const IntP &ar, const ModC &, int) const { return ar.canon128(s); }
    DEV u64 fin_lazy(const IntP &ar, const ModC &, int) const { return ar.fold128(s); }
};
template <>
struct IpAcc<FpA> {
    typedef double V;
    double s = 0;
    DEV static V prep(const FpA &ar, double c) {
        ar.recentre(c);
        return c;
    }
    DEV static V canon(u64 v) { return FpA::u2d(v); }
    DEV void mac(const FpA &ar, V v, u64 key) { s += ar.mulmod2(v, FpA::u2d(key)); }
    DEV u64 fin(const FpA &ar, const ModC &, int) const { return ar.fin_fwd(s); }
    DEV u64 fin_lazy(const FpA &ar, const ModC &M, int terms) const { return fin(ar, M, terms); }
};
template <int ST, class A>
DEV P2Pre p2_prefetch(const NttStore &st, const A &ar, int xp, int j, unsigned idx, const ModC &M) {
    constexpr size_t N = 32768;
    P2Pre r;
    r.has_ex = false;
    r.ex0 = r.ex1 = make_ulonglong2(0, 0);
    if (ST == 5) {
        loop_a_inner_product<false>(st.la, ar, M, xp >> 1, xp & 1, j, idx, r.in0, r.in1);
    } else {
        const u64 *pi = st.in + ((size_t)xp * st.in_ls + j) * N + idx;
        r.in0 = *reinterpret_cast<const ulonglong2 *>(pi);
        r.in1 = *reinterpret_cast<const ulonglong2 *>(pi + 2);
    }
    r.sb0 = r.sb1 = make_ulonglong2(0, 0);
    r.has_sb = false;
    if (ST == 1 || ST == 3 || ST == 5 || ST == 9 || ST == 10 || ST == 11) {
        const int x = xp >> 1, p = xp & 1;
        if (st.addend && p < st.add_polys && ST != 9 && ST != 10 && ST != 11) {
            const u64 *pa = st.addend + (size_t)x * st.add_x + (size_t)p * st.add_p + (size_t)j * N + idx;
            r.ex0 = *reinterpret_cast<const ulonglong2 *>(pa);
            r.ex1 = *reinterpret_cast<const ulonglong2 *>(pa + 2);
            r.has_ex = true;
        }
        if (ST == 3 && st.sub) {  // (mode 10 loads `sub` where it is used: 16 registers less across the last exchange -> four waves per SIMD)
            const u64 *ps = st.sub + ((size_t)xp * st.sub_ls + j) * N + idx;
            r.sb0 = *reinterpret_cast<const ulonglong2 *>(ps);
            r.sb1 = *reinterpret_cast<const ulonglong2 *>(ps + 2);
            r.has_sb = true;
        }
    } else if (ST == 2) {
        if (st.sub) {
            const u64 *ps = st.sub + ((size_t)xp * st.sub_ls + j) * N + idx;
            r.ex0 = *reinterpret_cast<const ulonglong2 *>(ps);
            r.ex1 = *reinterpret_cast<const ulonglong2 *>(ps + 2);
            r.has_ex = true;
        }
    }
    return r;
}
template <int ST>
DEV void p2_store_perm(const NttStore &st, int xp, int j, unsigned idx, const u64 (&r)[4]);
template <int ST>
DEV void p2_finish(const NttStore &st, const ModC &M, int xp, int j, unsigned idx, const u64 v[4], const P2Pre &pre, const P2Prod &po) {
    constexpr size_t N = 32768;
    const u64 q = M.q, mul = st.mul.s[j], muls = st.mul.s_sh[j];
    const u64 iv[4] = {pre.in0.x, pre.in0.y, pre.in1.x, pre.in1.y};
    u64 ev[4] = {pre.ex0.x, pre.ex0.y, pre.ex1.x, pre.ex1.y};
    u64 r[4];
    bool has_ex = pre.has_ex;
    if (ST == 9 || ST == 10 || ST == 11) {  // the addend is the product's d_p, formed here
#pragma unroll
        for (int k = 0; k < 4; k++) ev[k] = po.dp_int(M, xp & 1, k);
        has_ex = true;
        if (ST == 11) {  // d_p -= kap c_p
            u64 cv[4];
            P2Prod::load_c(st.prod, xp >> 1, xp & 1, j, idx, cv);
#pragma unroll
            for (int k = 0; k < 4; k++) ev[k] = submod(ev[k], mulmod_shoup(cv[k], st.kap.s[j], st.kap.s_sh[j], q), q);
        }
    }
    if (ST == 3 || ST == 9 || ST == 10 || ST == 11) {  // ((in P^{-1} + addend)(x2) - v) q_l^{-1} (- sub)(+ addc): ModDown and Rescale in one epilogue
        const u64 m2 = st.mul2.s[j], m2s = st.mul2.s_sh[j];
        u64 sv[4] = {pre.sb0.x, pre.sb0.y, pre.sb1.x, pre.sb1.y};
        bool has_sb = pre.has_sb;
        if (ST == 10 && st.sub) {  // loaded here, not across the last exchange (see p2_prefetch)
            const u64 *ps = st.sub + ((size_t)xp * st.sub_ls + j) * N + idx;
            const ulonglong2 s0 = *reinterpret_cast<const ulonglong2 *>(ps), s1 = *reinterpret_cast<const ulonglong2 *>(ps + 2);
            sv[0] = s0.x; sv[1] = s0.y; sv[2] = s1.x; sv[3] = s1.y;
            has_sb = true;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u64 t = mulmod_shoup(iv[k], mul, muls, q);
            if (has_ex) t = addmod(t, ev[k], q);
            if (st.dbl) t = addmod(t, t, q);
            t = mulmod_shoup(submod(t, v[k], q), m2, m2s, q);
            if (has_sb) t = st.sub_add ? addmod(t, sv[k], q) : submod(t, sv[k], q);
            if (st.has_addc && (xp % st.npoly) == 0) t = addmod(t, st.addc[j], q);
            r[k] = t;
        }
        u64 *o3 = st.out + ((size_t)xp * st.nl + j) * N;
        *reinterpret_cast<ulonglong2 *>(o3 + idx) = make_ulonglong2(r[0], r[1]);
        *reinterpret_cast<ulonglong2 *>(o3 + idx + 2) = make_ulonglong2(r[2], r[3]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        r[k] = (ST == 5 && st.la.premul) ? submod(iv[k], v[k], q) : mulmod_shoup(submod(iv[k], v[k], q), mul, muls, q);
        if (ST == 1 || ST == 5) {
            if (pre.has_ex) r[k] = addmod(r[k], ev[k], q);
            if (st.dbl) r[k] = addmod(r[k], r[k], q);
        } else {
            if (pre.has_ex) r[k] = submod(r[k], ev[k], q);
            if (st.has_addc && (xp % st.npoly) == 0) r[k] = addmod(r[k], st.addc[j], q);
        }
    }
    p2_store_perm<ST>(st, xp, j, idx, r);
}
// four finished coefficients idx .. idx+3 of limb j of polynomial xp: stored in place, or through the automorphism (modes 1, 5)
template <int ST>
DEV void p2_store_perm(const NttStore &st, int xp, int j, unsigned idx, const u64 (&r)[4]) {
    constexpr size_t N = 32768;
    u64 *o = st.out + ((size_t)xp * st.nl + j) * N;
    unsigned g = 1u;
    if ((ST == 1 || ST == 5) && st.ginv) g = st.ginv[st.same_g ? 0 : (xp >> 1)];
    if (g != 1u) {
        // In bit-reversed order the automorphism maps the aligned group of four {idx .. idx+3} onto ONE aligned group of four: the two
        // low index bits are the two high bits of brev(c), they only move the two high bits of e (g is odd), i.e. the two low bits of
        // the target.  So the lane permutes its four values in registers and stores 32 contiguous bytes instead of scattering four
        // 8-byte words: value k lands at base + brev2((t0 + brev2(k) g) mod 4), hence slot j holds k = brev2(((brev2(j) - t0) g) mod 4)
        const unsigned o0 = perm_idx(idx, g), base = o0 & ~3u;
        const unsigned t0 = ((o0 & 1u) << 1) | ((o0 >> 1) & 1u);
        u64 w[4];
#pragma unroll
        for (unsigned jj = 0; jj < 4; jj++) {
            const unsigned bj = ((jj & 1u) << 1) | (jj >> 1);
            const unsigned kb = ((bj - t0) * g) & 3u;  // = brev2(k): bit 1 of kb is bit 0 of k
            const u64 lo = (kb & 1u) ? r[2] : r[0], hi = (kb & 1u) ? r[3] : r[1];  // k bit 1 = kb bit 0
            w[jj] = (kb & 2u) ? hi : lo;                                           // k bit 0 = kb bit 1
        }
        *reinterpret_cast<ulonglong2 *>(o + base) = make_ulonglong2(w[0], w[1]);
        *reinterpret_cast<ulonglong2 *>(o + base + 2) = make_ulonglong2(w[2], w[3]);
    } else {
        *reinterpret_cast<ulonglong2 *>(o + idx) = make_ulonglong2(r[0], r[1]);
        *reinterpret_cast<ulonglong2 *>(o + idx + 2) = make_ulonglong2(r[2], r[3]);
    }
}

// Loop A's combine (mode 5, pre-scaled keys) of a limb below 2^47 on the FP64 pipe (round 4): the inner product's exact small sum minus
// the transform's UNREDUCED output (+ the addend), ONE reduction — instead of two reductions, a modular subtraction and a modular
// addition on canonical residues.  Same canonical result.
DEV void p2_finish5_fp(const NttStore &st, const FpA &ar, int xp, int j, unsigned idx, const double (&v)[4], const P2Pre &pre) {
    const u64 iv[4] = {pre.in0.x, pre.in0.y, pre.in1.x, pre.in1.y};
    const u64 ev[4] = {pre.ex0.x, pre.ex0.y, pre.ex1.x, pre.ex1.y};
    u64 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double t = FpA::from_bits(iv[k]) - v[k];
        if (pre.has_ex) t += FpA::u2d(ev[k]);
        r[k] = ar.fin_fwd(t);
    }
    p2_store_perm<5>(st, xp, j, idx, r);
}
template <class A>
DEV void p2_finish5_fp(const NttStore &, const A &, int, int, unsigned, const typename A::T (&)[4], const P2Pre &) {}

// Merged ModDown + Rescale epilogue (mode 3) of a limb below 2^47 on the FP64 pipe (round 4).  The integer form runs two Shoup products
// with their conditional subtractions per coefficient on the canonical transform output (57 instructions with fin_fwd); but
//   ((in P^{-1} + ex)(x2) - v) q_l^{-1} (+- sb)(+ addc)  =  in K1 + ex K2 - v K3 (+- sb)(+ addc),
// K1 = P^{-1} (x2) q_l^{-1}, K2 = (x2) q_l^{-1}, K3 = q_l^{-1} mod q, so three exact FP64 products with precomputed quotients take the
// operands as they are — `in`, `ex`, `sb` canonical, v the transform's UNREDUCED output (|v| < 2^50) — and ONE reduction finishes:
// |sum| < 6q, exact in a double.  35 instructions; the same canonical residue (every step is exact modulo q).
struct Epi3Fp {
    double2 k1, k2, k3, kap;
    double addc;
    template <int ST, class A>
    DEV static Epi3Fp make(const NttStore &st, const A &, const ModC &, int) {
        return Epi3Fp{};
    }
};
DEV Epi3Fp epi3fp_make(const NttStore &st, const FpA &ar, const ModC &M, int j) {
    Epi3Fp e;
    const u64 m2 = st.mul2.s[j];
    const u64 two = st.dbl ? addmod(m2, m2, M.q) : m2;  // (x2) q_l^{-1}
    e.k1 = ar.tw8(FpA::u2d(mulmod(st.mul.s[j], two, M)));
    e.k2 = ar.tw8(FpA::u2d(two));
    e.k3 = ar.tw8(FpA::u2d(m2));
    e.addc = st.has_addc ? FpA::u2d(st.addc[j]) : 0.0;
    e.kap = ar.tw8(FpA::u2d(st.has_prod && st.prod.c ? st.kap.s[j] : 0));
    return e;
}
template <>
DEV Epi3Fp Epi3Fp::make<3, FpA>(const NttStore &st, const FpA &ar, const ModC &M, int j) { return epi3fp_make(st, ar, M, j); }
template <>
DEV Epi3Fp Epi3Fp::make<9, FpA>(const NttStore &st, const FpA &ar, const ModC &M, int j) { return epi3fp_make(st, ar, M, j); }
template <>
DEV Epi3Fp Epi3Fp::make<10, FpA>(const NttStore &st, const FpA &ar, const ModC &M, int j) { return epi3fp_make(st, ar, M, j); }
template <>
DEV Epi3Fp Epi3Fp::make<11, FpA>(const NttStore &st, const FpA &ar, const ModC &M, int j) { return epi3fp_make(st, ar, M, j); }
template <bool PROD, bool LATE_SB, bool HAS_C>
DEV void p2_finish3_fp(const NttStore &st, const FpA &ar, int xp, int j, unsigned idx, const double (&v)[4], const P2Pre &pre, const Epi3Fp &e,
                       const P2Prod &po) {
    constexpr size_t N = 32768;
    const u64 iv[4] = {pre.in0.x, pre.in0.y, pre.in1.x, pre.in1.y};
    const u64 ev[4] = {pre.ex0.x, pre.ex0.y, pre.ex1.x, pre.ex1.y};
    u64 sv[4] = {pre.sb0.x, pre.sb0.y, pre.sb1.x, pre.sb1.y};
    bool has_sb = pre.has_sb;
    if (LATE_SB && st.sub) {
        const u64 *ps = st.sub + ((size_t)xp * st.sub_ls + j) * N + idx;
        const ulonglong2 s0 = *reinterpret_cast<const ulonglong2 *>(ps), s1 = *reinterpret_cast<const ulonglong2 *>(ps + 2);
        sv[0] = s0.x; sv[1] = s0.y; sv[2] = s1.x; sv[3] = s1.y;
        has_sb = true;
    }
    const bool addc = st.has_addc && (xp % st.npoly) == 0;
    u64 cv[4] = {0, 0, 0, 0};
    constexpr bool has_c = PROD && HAS_C;
    if (has_c) P2Prod::load_c(st.prod, xp >> 1, xp & 1, j, idx, cv);
    u64 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double t = ar.mulmod(FpA::u2d(iv[k]), e.k1) - ar.mulmod(v[k], e.k3);
        if (PROD && has_c) t += ar.mulmod(po.dp_fp(ar, xp & 1, k) - ar.mulmod(FpA::u2d(cv[k]), e.kap), e.k2);
        else if (PROD) t += ar.mulmod(po.dp_fp(ar, xp & 1, k), e.k2);
        else if (pre.has_ex) t += ar.mulmod(FpA::u2d(ev[k]), e.k2);
        if (has_sb) t += st.sub_add ? FpA::u2d(sv[k]) : -FpA::u2d(sv[k]);
        if (addc) t += e.addc;
        r[k] = ar.fin_fwd(t);
    }
    u64 *o3 = st.out + ((size_t)xp * st.nl + j) * N;
    *reinterpret_cast<ulonglong2 *>(o3 + idx) = make_ulonglong2(r[0], r[1]);
    *reinterpret_cast<ulonglong2 *>(o3 + idx + 2) = make_ulonglong2(r[2], r[3]);
}
template <bool PROD, bool LATE_SB, bool HAS_C, class A>
DEV void p2_finish3_fp(const NttStore &, const A &, int, int, unsigned, const typename A::T (&)[4], const P2Pre &, const Epi3Fp &, const P2Prod &) {}

// The LDS image of a pass-2 workgroup: 8 blocks x 8 rows x 32 coefficients per polynomial.  Padded (rows of 36: phase B's (row, 4k + b)
// accesses of a half-wave hit 32 distinct bank pairs) or, SWZ, unpadded with the position XOR-ed by 4 x row — the same property at
// 16 KiB instead of 18 per polynomial, which lets the three-digit fused inner product keep THREE workgroups on a CU (3 x 48 KiB) where
// the padded image allows two; a few more address instructions per access, so only that kernel takes it.
template <bool SWZ>
struct P2Lds {
    static constexpr int SIZE = SWZ ? 8 * 256 : 8 * 288;
    DEV static int at(int blk, int row, int pos) { return SWZ ? blk * 256 + row * 32 + (pos ^ (row << 2)) : blk * 288 + row * 36 + pos; }
};

// Round 5: pass 2 is WAVE-SYNCHRONOUS.  A 256-coefficient block belongs to one half-wave (blk = t >> 5) in EVERY phase — phase C takes the
// two groups of four consecutive coefficients 4w + 128 hh of its own block instead of 4t + 1024 hh of the workgroup's chunk — so every
// LDS exchange is between lanes of one wave: the LDS executes a wave's instructions in order, a ds_read after a ds_write needs no
// s_barrier, and the four waves of a workgroup never wait for each other.  Same butterflies on the same operands: bit-identical.
// WGS = true is round 4's form (chunk-wide phase C behind s_barrier), kept as a parity variant of the PLAIN transforms: HYDIA_P2_WG_SYNC=1
// runs k_ntt15_p2<*, *, 0, true> (tests/test_gpu_parity.py::test_ntt_bit_exact); -DHYDIA_P2_WG_SYNC builds every kernel that way (A/B builds).
#ifdef HYDIA_P2_WG_SYNC
constexpr bool P2_WGS_DEFAULT = true;
#else
constexpr bool P2_WGS_DEFAULT = false;
#endif
template <bool WGS>
DEV void p2_sync() {
    if (WGS) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler-only at this scope: LDS accesses stay on their side of the exchange
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
template <bool WGS>
DEV int p2_elem(int t, int hh) { return WGS ? 4 * t + 1024 * hh : (t >> 5) * 256 + 4 * (t & 31) + 128 * hh; }

// Round 5: the twiddles of pass 2's phases A / B (and B' / A') of an FP64 limb come from a WAVE-LOCAL LDS table instead of fourteen 16-byte
// vector loads per lane that fetch 2 (phase A) or 16 (phase B) distinct entries per wave.  The texture addresser spends its 16 cycles on
// every such load whatever the lanes share, and it is what the pass-2 kernels are short of (TA_BUSY 0.67 - 0.79 of the duration in the
// epilogue kernels, profiles/r05/query_counters_q20.txt): twiddle loads were ~55 % of a plain transform's vector-memory cycles.  A wave's two
// blocks bg0, bg0 + 1 need one contiguous run of the table per stage — tw[128 + bg0 ..+2), [256 + 2 bg0 ..+4), [512 + 4 bg0 ..+8),
// [1024 + 8 bg0 ..+16), [2048 + 16 bg0 ..+32), [4096 + 32 bg0 ..+64) — 126 entries at offsets 0, 2, 6, 14, 30, 62 of the wave's table,
// fetched with TWO coalesced 8-byte loads per lane from the twiddles-alone table (NttTables::twd / itwd; w / q is formed as w (1/q),
// FpA::tw8: an exact representative either way, same residues) — 1 KiB of LDS per wave, so no kernel loses a workgroup per CU.
// tools/ubench/p2_swap.hip, fifth form: 5.39 against 4.81 TB/s for a plain FP64 pass 2 in the streaming regime.  The 60-bit limbs (16-byte
// entries: a table would cost a workgroup per CU) and the workgroup-synchronous parity variant load as before.  HYDIA_NO_TW_LDS=1: off.
struct P2Tab {
    const double *g;  // the limb's twiddles-alone table in global memory, or null: no table
    double *l;        // the workgroup's [4][128] doubles in LDS
};
DEV void p2_stage_twiddles(const double *__restrict__ g, double *wt, int bg0, int lane) {
    double st[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int i = lane + 64 * r;
        const int src = i < 2 ? 128 + bg0 + i : i < 6 ? 256 + 2 * bg0 + (i - 2) : i < 14 ? 512 + 4 * bg0 + (i - 6) :
                        i < 30 ? 1024 + 8 * bg0 + (i - 14) : i < 62 ? 2048 + 16 * bg0 + (i - 30) : 4096 + 32 * bg0 + (i - 62);
        st[r] = g[i < 126 ? src : 128 + bg0];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) wt[lane + 64 * r] = st[r];
}
// the seven twiddles of phase A / A' (block hb of the wave) and of phase B / B' (row group ibw = 8 hb + a of the wave)
template <class A>
DEV void p2_tw_A(const A &, const ulonglong2 *__restrict__ tw, const double *, int bg, int, typename A::TW &W7, typename A::TW &W8a, typename A::TW &W8b,
                 typename A::TW (&W9)[4]) {
    W7 = A::tw(tw[128 + bg]);
    W8a = A::tw(tw[256 + 2 * bg]);
    W8b = A::tw(tw[256 + 2 * bg + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) W9[i] = A::tw(tw[512 + 4 * bg + i]);
}
DEV void p2_tw_A(const FpA &ar, const ulonglong2 *__restrict__ tw, const double *wt, int bg, int hb, FpA::TW &W7, FpA::TW &W8a, FpA::TW &W8b, FpA::TW (&W9)[4]) {
    if (wt) {
        W7 = ar.tw8(wt[hb]);
        W8a = ar.tw8(wt[2 + 2 * hb]);
        W8b = ar.tw8(wt[2 + 2 * hb + 1]);
#pragma unroll
        for (int i = 0; i < 4; i++) W9[i] = ar.tw8(wt[6 + 4 * hb + i]);
        return;
    }
    p2_tw_A<FpA>(ar, tw, nullptr, bg, hb, W7, W8a, W8b, W9);
}
template <class A>
DEV void p2_tw_B(const A &, const ulonglong2 *__restrict__ tw, const double *, int ib, int, typename A::TW &W10, typename A::TW &W11a, typename A::TW &W11b,
                 typename A::TW (&W12)[4]) {
    W10 = A::tw(tw[1024 + ib]);
    W11a = A::tw(tw[2048 + 2 * ib]);
    W11b = A::tw(tw[2048 + 2 * ib + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) W12[i] = A::tw(tw[4096 + 4 * ib + i]);
}
DEV void p2_tw_B(const FpA &ar, const ulonglong2 *__restrict__ tw, const double *wt, int ib, int ibw, FpA::TW &W10, FpA::TW &W11a, FpA::TW &W11b, FpA::TW (&W12)[4]) {
    if (wt) {
        W10 = ar.tw8(wt[14 + ibw]);
        W11a = ar.tw8(wt[30 + 2 * ibw]);
        W11b = ar.tw8(wt[30 + 2 * ibw + 1]);
#pragma unroll
        for (int i = 0; i < 4; i++) W12[i] = ar.tw8(wt[62 + 4 * ibw + i]);
        return;
    }
    p2_tw_B<FpA>(ar, tw, nullptr, ib, ibw, W10, W11a, W11b, W12);
}

// phases B' and A' of the inverse second pass for NPI polynomials whose phase-C' output sits in lds (caller synchronised)
template <class A, int NPI, bool SWZ = false, bool WGS = P2_WGS_DEFAULT>
DEV void p2_inverse_BA(const A &ar, const ulonglong2 *__restrict__ tw, u64 (*lds)[P2Lds<SWZ>::SIZE], u64 *const *d, int t, int B0,
                       const double *wt = nullptr /* the wave's staged table of THIS direction's twiddles (FP64 limbs), or null */) {
    typedef typename A::T T;
    typedef typename A::TW TW;
    typedef P2Lds<SWZ> LI;
    constexpr int NP = NPI;
    const int blk = t >> 5, w = t & 31;
    const int bg = (B0 >> 8) + blk;
    const int a = w >> 2, b = w & 3;
    T v[NP][8];
    // phase B': strides 4, 8, 16
    {
        const int ib = 8 * bg + a;
        TW W10, W11a, W11b, W12[4];
        p2_tw_B(ar, tw, wt, ib, 8 * (blk & 1) + a, W10, W11a, W11b, W12);
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[p][k] = A::from_bits(lds[p][LI::at(blk, a, 4 * k + b)]);
#pragma unroll
            for (int k = 0; k < 8; k += 2) ar.gs(v[p][k], v[p][k + 1], W12[k >> 1]);
            ar.gs(v[p][0], v[p][2], W11a);
            ar.gs(v[p][1], v[p][3], W11a);
            ar.gs(v[p][4], v[p][6], W11b);
            ar.gs(v[p][5], v[p][7], W11b);
#pragma unroll
            for (int k = 0; k < 4; k++) ar.gs(v[p][k], v[p][k + 4], W10);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                ar.recentre(v[p][k]);
                lds[p][LI::at(blk, a, 4 * k + b)] = A::to_bits(v[p][k]);
            }
        }
    }
    p2_sync<WGS>();
    // phase A': strides 32, 64, 128
    {
        TW W7, W8a, W8b, W9[4];
        p2_tw_A(ar, tw, wt, bg, blk & 1, W7, W8a, W8b, W9);
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[p][k] = A::from_bits(lds[p][LI::at(blk, k, w)]);
#pragma unroll
            for (int k = 0; k < 8; k += 2) ar.gs(v[p][k], v[p][k + 1], W9[k >> 1]);
            ar.gs(v[p][0], v[p][2], W8a);
            ar.gs(v[p][1], v[p][3], W8a);
            ar.gs(v[p][4], v[p][6], W8b);
            ar.gs(v[p][5], v[p][7], W8b);
#pragma unroll
            for (int k = 0; k < 4; k++) ar.gs(v[p][k], v[p][k + 4], W7);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                ar.recentre_wide(v[p][k]);
                d[p][blk * 256 + 32 * k + w] = A::to_bits(v[p][k]);  // raw: pass 1' finishes
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ pass 2 (contiguous)
// 256 threads: blk = t>>5, w = t&31.  NP polynomials (1 or 2, same modulus) share every twiddle load.  LDS rows of 32
// coefficients are padded to 36 so phase B's (a, b) reads hit 64 distinct banks.
// ST 6 = ST 4 for the special-prime limbs with the inverse transform's second pass appended: the two sums of a lane's four
// coefficients go through phase C' in registers and meet the other lanes' in lds[0], lds[1] (the digits' images there are dead by
// then: every lane reads and overwrites only its own four slots), phases B', A' follow, the raw image leaves through dinv.
template <class A, bool INV, int NP, int ST, bool SWZ = false, bool WGS = P2_WGS_DEFAULT>
DEV void p2_body(const A ar, const ulonglong2 *__restrict__ tw, const u64 *const *s, u64 *const *d, u64 (*lds)[P2Lds<SWZ>::SIZE], int t,
                 int B0, const NttStore &stp, const ModC &M, int xp0, int slot, const ulonglong2 *__restrict__ itw = nullptr,
                 u64 *const *dinv = nullptr, const P2Tab tab = P2Tab{nullptr, nullptr}) {
    typedef typename A::T T;
    typedef typename A::TW TW;
    typedef P2Lds<SWZ> LI;
    const int blk = t >> 5, w = t & 31;
    const int bg = (B0 >> 8) + blk;
    const int a = w >> 2, b = w & 3;
    // FP64 limbs: the wave's table of phase A / B twiddles (requested FIRST: the loads return in order, and the table is wanted before the data)
    const double *wt = nullptr;
    if (std::is_same<A, FpA>::value && !WGS && tab.g && tab.l) {
        double *wl = tab.l + (t >> 6) * 128;
        p2_stage_twiddles(tab.g, wl, (B0 >> 8) + 2 * (t >> 6), t & 63);
        wt = wl;
    }
    T v[NP][8];
    if (!INV) {
        // phase A: coefficients blk*256 + 32k + w ; stages 7,8,9 (strides 128, 64, 32)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int k = 0; k < 8; k++) v[p][k] = ar.from_raw(s[p][blk * 256 + 32 * k + w]);  // raw from pass 1
        {
            if (wt) p2_sync<false>();  // the table is in place (a wave's own writes, in order)
            TW W7, W8a, W8b, W9[4];
            p2_tw_A(ar, tw, wt, bg, blk & 1, W7, W8a, W8b, W9);
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 4; k++) ar.ct(v[p][k], v[p][k + 4], W7);
                ar.ct(v[p][0], v[p][2], W8a);
                ar.ct(v[p][1], v[p][3], W8a);
                ar.ct(v[p][4], v[p][6], W8b);
                ar.ct(v[p][5], v[p][7], W8b);
#pragma unroll
                for (int k = 0; k < 8; k += 2) ar.ct(v[p][k], v[p][k + 1], W9[k >> 1]);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    ar.fwd_fold(v[p][k]);
                    lds[p][LI::at(blk, k, w)] = A::to_bits(v[p][k]);
                }
            }
        }
        p2_sync<WGS>();
        // phase B: coefficients blk*256 + 32a + 4k + b ; stages 10,11,12 (strides 16, 8, 4)
        {
            const int ib = 8 * bg + a;
            TW W10, W11a, W11b, W12[4];
            p2_tw_B(ar, tw, wt, ib, 8 * (blk & 1) + a, W10, W11a, W11b, W12);
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[p][k] = A::from_bits(lds[p][LI::at(blk, a, 4 * k + b)]);
#pragma unroll
                for (int k = 0; k < 4; k++) ar.ct(v[p][k], v[p][k + 4], W10);
                ar.ct(v[p][0], v[p][2], W11a);
                ar.ct(v[p][1], v[p][3], W11a);
                ar.ct(v[p][4], v[p][6], W11b);
                ar.ct(v[p][5], v[p][7], W11b);
#pragma unroll
                for (int k = 0; k < 8; k += 2) ar.ct(v[p][k], v[p][k + 1], W12[k >> 1]);
#pragma unroll
                for (int k = 0; k < 8; k++) lds[p][LI::at(blk, a, 4 * k + b)] = A::to_bits(v[p][k]);
            }
        }
        // the merged epilogues' operands are fetched one half at a time (the second half's while the first half is finished): 109 / 120
        // instead of 136 / 150 registers, i.e. 4 instead of 3 waves per SIMD (-0.4 ms per query at 2^14, -0.8 ms at 2^20)
        constexpr bool SPLIT = ST == 5 || ST == 3 || ST == 9 || ST == 10 || ST == 11;  // (9, 10 = 3 with the addend formed from a fused product: no `ex` operand; 9 has no `sub` operand either — 126 registers, four waves per SIMD)
        const Epi3Fp epi = Epi3Fp::make<ST, A>(stp, ar, M, slot);
        P2Pre pre[2][NP];
        P2Prod po;  // mode 3 with a fused product: a0, a1, b0, b1 of the ciphertext (shared by its two polynomials when NP = 2)
        if (ST != 0 && ST != 4 && ST != 6) {
#pragma unroll
            for (int hh = 0; hh < (SPLIT ? 1 : 2); hh++)
#pragma unroll
                for (int p = 0; p < NP; p++) pre[hh][p] = p2_prefetch<ST>(stp, ar, xp0 + p, slot, (unsigned)(B0 + p2_elem<WGS>(t, hh)), M);
            if (ST == 9 || ST == 10 || ST == 11) po.load(stp.prod, xp0 >> 1, slot, (unsigned)(B0 + p2_elem<WGS>(t, 0)));
        }
        // mode 4 / 6: the key rows of the inner product
        const u64 *const ip_key = (ST == 4 || ST == 6) ? (stp.ip.keys ? stp.ip.keys[xp0] : stp.ip.key) : nullptr;  // one key, or one per ciphertext
        const int ip_t = slot, ip_m = ip_t < stp.ip.nl ? ip_t : stp.ip.nT - stp.ip.nE + ip_t;
        const int ip_own = (stp.ip.own && ip_t < stp.ip.nl) ? ip_t / stp.ip.alpha : (1 << 30);
        // Round 5 (in-kernel section timing, profiles/r05/ip_stamps.txt): an FP64 row of the fused inner product spent 84 % of its time in phase C,
        // whose vector work is a fifth of that — the group's twiddles and each digit's key residues were requested one after the other, each
        // where it is used, and every request waited out a full L2 / Infinity-Cache round trip.  The FP64 path has the registers the 60-bit path
        // sets the kernel's budget with, so it requests a group's twiddles and EVERY digit's key residues in one batch — the first group's before
        // the last exchange, the second group's at the top of its iteration.
        constexpr bool KB = (ST == 4 || ST == 6) && std::is_same<A, FpA>::value && !WGS;
        ulonglong2 kq[KB ? NP : 1][4];
        TW Wq[3];
        auto ip_request = [&](int hh) {
            const int e = p2_elem<WGS>(t, hh), gi = (B0 + e) >> 2;
            Wq[0] = A::tw(tw[8192 + gi]);
            Wq[1] = A::tw(tw[16384 + 2 * gi]);
            Wq[2] = A::tw(tw[16384 + 2 * gi + 1]);
#pragma unroll
            for (int p = 0; p < (KB ? NP : 1); p++) {
                const int dgt = p >= ip_own ? p + 1 : p;
                const u64 *kb = ip_key + (((size_t)dgt * 2) * stp.ip.nT + ip_m) * 32768 + (B0 + e);
                const u64 *ka = kb + (size_t)stp.ip.nT * 32768;
                kq[p][0] = *reinterpret_cast<const ulonglong2 *>(kb);
                kq[p][1] = *reinterpret_cast<const ulonglong2 *>(kb + 2);
                kq[p][2] = *reinterpret_cast<const ulonglong2 *>(ka);
                kq[p][3] = *reinterpret_cast<const ulonglong2 *>(ka + 2);
            }
        };
        if (KB) ip_request(0);
        p2_sync<WGS>();
        // phase C: two groups of 4 consecutive coefficients e = 4t + 1024*hh ; stages 13, 14 (strides 2, 1)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int e = p2_elem<WGS>(t, hh), u = e & 255, la = LI::at(e >> 8, u >> 5, u & 31);  // four consecutive slots (u & 31 is a multiple of 4)
            const int gi = (B0 + e) >> 2;
            if (KB && hh == 1) ip_request(1);
            const TW W13 = KB ? Wq[0] : A::tw(tw[8192 + gi]), W14a = KB ? Wq[1] : A::tw(tw[16384 + 2 * gi]), W14b = KB ? Wq[2] : A::tw(tw[16384 + 2 * gi + 1]);
            // mode 4: lazy 128-bit sums of value * key over the digits this workgroup transforms (+ the limb's own digit)
            IpAcc<A> ipb[4], ipa[4];
            if (SPLIT && hh == 1) {
#pragma unroll
                for (int p = 0; p < NP; p++) pre[0][p] = p2_prefetch<ST>(stp, ar, xp0 + p, slot, (unsigned)(B0 + p2_elem<WGS>(t, 1)), M);
                if (ST == 9 || ST == 10 || ST == 11) po.load(stp.prod, xp0 >> 1, slot, (unsigned)(B0 + p2_elem<WGS>(t, 1)));
            }
#pragma unroll
            for (int p = 0; p < NP; p++) {
                T c0 = A::from_bits(lds[p][la]), c1 = A::from_bits(lds[p][la + 1]), c2 = A::from_bits(lds[p][la + 2]),
                  c3 = A::from_bits(lds[p][la + 3]);
                ar.mid(c0); ar.mid(c1); ar.mid(c2); ar.mid(c3);
                ar.ct(c0, c2, W13);
                ar.ct(c1, c3, W13);
                ar.ct(c0, c1, W14a);
                ar.ct(c2, c3, W14b);
                if (ST == 4 || ST == 6) {
                    const typename IpAcc<A>::V vv[4] = {IpAcc<A>::prep(ar, c0), IpAcc<A>::prep(ar, c1), IpAcc<A>::prep(ar, c2), IpAcc<A>::prep(ar, c3)};
                    const int dgt = p >= ip_own ? p + 1 : p;
                    const u64 *kb = ip_key + (((size_t)dgt * 2) * stp.ip.nT + ip_m) * 32768 + (B0 + e);
                    const u64 *ka = kb + (size_t)stp.ip.nT * 32768;
                    const ulonglong2 b0 = KB ? kq[KB ? p : 0][0] : *reinterpret_cast<const ulonglong2 *>(kb), b1 = KB ? kq[KB ? p : 0][1] : *reinterpret_cast<const ulonglong2 *>(kb + 2);
                    const ulonglong2 a0 = KB ? kq[KB ? p : 0][2] : *reinterpret_cast<const ulonglong2 *>(ka), a1 = KB ? kq[KB ? p : 0][3] : *reinterpret_cast<const ulonglong2 *>(ka + 2);
                    ipb[0].mac(ar, vv[0], b0.x); ipb[1].mac(ar, vv[1], b0.y); ipb[2].mac(ar, vv[2], b1.x); ipb[3].mac(ar, vv[3], b1.y);
                    ipa[0].mac(ar, vv[0], a0.x); ipa[1].mac(ar, vv[1], a0.y); ipa[2].mac(ar, vv[2], a1.x); ipa[3].mac(ar, vv[3], a1.y);
                } else if (ST == 0) {
                    ulonglong2 o0, o1;
                    o0.x = ar.fin_fwd(c0); o0.y = ar.fin_fwd(c1);
                    o1.x = ar.fin_fwd(c2); o1.y = ar.fin_fwd(c3);
                    *reinterpret_cast<ulonglong2 *>(d[p] + e) = o0;
                    *reinterpret_cast<ulonglong2 *>(d[p] + e + 2) = o1;
                } else if (ST == 5 && std::is_same<A, FpA>::value && stp.la.raw_fp) {
                    const T cv[4] = {c0, c1, c2, c3};
                    p2_finish5_fp(stp, ar, xp0 + p, slot, (unsigned)(B0 + e), cv, pre[SPLIT ? 0 : hh][p]);
                } else if ((ST == 3 || ST == 9 || ST == 10 || ST == 11) && std::is_same<A, FpA>::value && !stp.int_epilogue) {
                    const T cv[4] = {c0, c1, c2, c3};
                    p2_finish3_fp<ST == 9 || ST == 10 || ST == 11, ST == 10, ST == 11>(stp, ar, xp0 + p, slot, (unsigned)(B0 + e), cv, pre[SPLIT ? 0 : hh][p], epi, po);
                } else {
                    const u64 vv[4] = {ar.fin_fwd(c0), ar.fin_fwd(c1), ar.fin_fwd(c2), ar.fin_fwd(c3)};
                    p2_finish<ST>(stp, M, xp0 + p, slot, (unsigned)(B0 + e), vv, pre[SPLIT ? 0 : hh][p], po);
                }
            }
            if (ST == 4 || ST == 6) {
                const size_t ci = (size_t)(B0 + e);
                if (ip_own < (1 << 30)) {
                    const u64 *cv = stp.ip.c2 + (size_t)xp0 * stp.ip.c2_xs + (size_t)ip_t * 32768 + ci;
                    const u64 *kb = ip_key + (((size_t)ip_own * 2) * stp.ip.nT + ip_m) * 32768 + ci;
                    const u64 *ka = kb + (size_t)stp.ip.nT * 32768;
                    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(cv), v1 = *reinterpret_cast<const ulonglong2 *>(cv + 2);
                    const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(kb), b1 = *reinterpret_cast<const ulonglong2 *>(kb + 2);
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(ka), a1 = *reinterpret_cast<const ulonglong2 *>(ka + 2);
                    const typename IpAcc<A>::V ov[4] = {IpAcc<A>::canon(v0.x), IpAcc<A>::canon(v0.y), IpAcc<A>::canon(v1.x), IpAcc<A>::canon(v1.y)};
                    ipb[0].mac(ar, ov[0], b0.x); ipb[1].mac(ar, ov[1], b0.y); ipb[2].mac(ar, ov[2], b1.x); ipb[3].mac(ar, ov[3], b1.y);
                    ipa[0].mac(ar, ov[0], a0.x); ipa[1].mac(ar, ov[1], a0.y); ipa[2].mac(ar, ov[2], a1.x); ipa[3].mac(ar, ov[3], a1.y);
                }
                if (ST == 6) {
                    const TW I13 = A::tw(itw[8192 + gi]), I14a = A::tw(itw[16384 + 2 * gi]), I14b = A::tw(itw[16384 + 2 * gi + 1]);
#pragma unroll
                    for (int pp = 0; pp < 2; pp++) {
                        IpAcc<A> *ip = pp == 0 ? ipb : ipa;
                        u64 f4[4];
                        if (ip_t != stp.ip.drop_l) {  // straight into the inverse transform: any representative its first two stages take
#pragma unroll
                            for (int k = 0; k < 4; k++) f4[k] = ip[k].fin_lazy(ar, M, NP + 1);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; k++) f4[k] = ip[k].fin(ar, M, NP + 1);
                        }
                        if (ip_t == stp.ip.drop_l) {  // workgroup-uniform: the limb the rescale drops — (sum P^{-1} + d_l)(x2), k_moddown_last_limb's arithmetic
                            u64 dv[4];
                            if (stp.ip.drop_has_prod) {  // d_p of the fused product at this limb
                                P2Prod dp;
                                dp.load(stp.ip.drop_prod, xp0, ip_t, (unsigned)ci);
#pragma unroll
                                for (int k = 0; k < 4; k++) dv[k] = dp.dp_int(M, pp, k);
                                if (NP >= 2 && stp.ip.drop_prod.c) {  // (Chebyshev steps with a subtrahend run at >= 9 limbs: three digits, NP >= 2)
                                    u64 cv[4];
                                    P2Prod::load_c(stp.ip.drop_prod, xp0, pp, ip_t, (unsigned)ci, cv);
#pragma unroll
                                    for (int k = 0; k < 4; k++)
                                        dv[k] = submod(dv[k], mulmod_shoup(cv[k], stp.ip.drop_prod.kap_l, stp.ip.drop_prod.kap_l_sh, M.q), M.q);
                                }
                            } else {
                                const u64 *pa = stp.ip.drop_add + (size_t)xp0 * stp.ip.drop_add_x + (size_t)pp * stp.ip.drop_add_p + (size_t)ip_t * 32768 + ci;
                                const ulonglong2 d0 = *reinterpret_cast<const ulonglong2 *>(pa), d1 = *reinterpret_cast<const ulonglong2 *>(pa + 2);
                                dv[0] = d0.x; dv[1] = d0.y; dv[2] = d1.x; dv[3] = d1.y;
                            }
                            const u64 qq = M.q;
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                f4[k] = addmod(mulmod_shoup(f4[k], stp.ip.drop_mul, stp.ip.drop_mul_sh, qq), dv[k], qq);
                                if (stp.ip.drop_dbl) f4[k] = addmod(f4[k], f4[k], qq);
                            }
                        }
                        T c0 = ar.from_canon(f4[0]), c1 = ar.from_canon(f4[1]), c2 = ar.from_canon(f4[2]), c3 = ar.from_canon(f4[3]);
                        ar.gs(c0, c1, I14a);
                        ar.gs(c2, c3, I14b);
                        ar.gs(c0, c2, I13);
                        ar.gs(c1, c3, I13);
                        ar.recentre_wide(c0); ar.recentre_wide(c1); ar.recentre_wide(c2); ar.recentre_wide(c3);
                        lds[pp][la] = A::to_bits(c0); lds[pp][la + 1] = A::to_bits(c1);
                        lds[pp][la + 2] = A::to_bits(c2); lds[pp][la + 3] = A::to_bits(c3);
                    }
                    continue;
                }
                u64 *ob = stp.ip.acc + (((size_t)xp0 * 2) * stp.ip.nE + ip_t) * 32768 + ci;
                u64 *oa = ob + (size_t)stp.ip.nE * 32768;
                *reinterpret_cast<ulonglong2 *>(ob) = make_ulonglong2(ipb[0].fin(ar, M, NP + 1), ipb[1].fin(ar, M, NP + 1));
                *reinterpret_cast<ulonglong2 *>(ob + 2) = make_ulonglong2(ipb[2].fin(ar, M, NP + 1), ipb[3].fin(ar, M, NP + 1));
                *reinterpret_cast<ulonglong2 *>(oa) = make_ulonglong2(ipa[0].fin(ar, M, NP + 1), ipa[1].fin(ar, M, NP + 1));
                *reinterpret_cast<ulonglong2 *>(oa + 2) = make_ulonglong2(ipa[2].fin(ar, M, NP + 1), ipa[3].fin(ar, M, NP + 1));
            }
        }
        if (ST == 6) {
            p2_sync<WGS>();
            p2_inverse_BA<A, 2, SWZ, WGS>(ar, itw, lds, dinv, t, B0);
        }
    } else {
        // phase C': strides 1, 2
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int e = p2_elem<WGS>(t, hh), u = e & 255, la = LI::at(e >> 8, u >> 5, u & 31);  // four consecutive slots (u & 31 is a multiple of 4)
            const int gi = (B0 + e) >> 2;
            const TW W13 = A::tw(tw[8192 + gi]), W14a = A::tw(tw[16384 + 2 * gi]), W14b = A::tw(tw[16384 + 2 * gi + 1]);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                ulonglong2 i0, i1;
                if (ST == 5) {  // the transform's input is loop A's inner product for these four coefficients
                    loop_a_inner_product<true>(stp.la, ar, M, (xp0 + p) >> 1, (xp0 + p) & 1, slot, (unsigned)(B0 + e), i0, i1);
                } else if (ST == 7) {  // ... is the dropped limb of the would-be ModDown output: (acc P^{-1} + addend)(x2), formed here
                    const int xq = xp0 + p;
                    const u64 *pa = stp.addend + (size_t)(xq >> 1) * stp.add_x + (size_t)(xq & 1) * stp.add_p + (size_t)stp.nl * 32768 + (B0 + e);
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(s[p] + e), a1 = *reinterpret_cast<const ulonglong2 *>(s[p] + e + 2);
                    const ulonglong2 d0 = *reinterpret_cast<const ulonglong2 *>(pa), d1 = *reinterpret_cast<const ulonglong2 *>(pa + 2);
                    const u64 qq = M.q, pv = stp.mul.s[slot], pvs = stp.mul.s_sh[slot];
                    i0.x = addmod(mulmod_shoup(a0.x, pv, pvs, qq), d0.x, qq);
                    i0.y = addmod(mulmod_shoup(a0.y, pv, pvs, qq), d0.y, qq);
                    i1.x = addmod(mulmod_shoup(a1.x, pv, pvs, qq), d1.x, qq);
                    i1.y = addmod(mulmod_shoup(a1.y, pv, pvs, qq), d1.y, qq);
                    if (stp.dbl) {
                        i0.x = addmod(i0.x, i0.x, qq); i0.y = addmod(i0.y, i0.y, qq);
                        i1.x = addmod(i1.x, i1.x, qq); i1.y = addmod(i1.y, i1.y, qq);
                    }
                } else if (ST == 8) {  // ... is d2 = a1 b1 of a fused product (ProdSrc); also kept for the inner product's own-digit rows
                    const int xq = xp0 + p;
                    const size_t off = (size_t)slot * 32768 + (size_t)(B0 + e);
                    const u64 *pa = stp.prod.a + (size_t)xq * stp.prod.a_x + stp.prod.a_p + off, *pb = stp.prod.b + (size_t)xq * stp.prod.b_x + stp.prod.b_p + off;
                    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(pa), a1 = *reinterpret_cast<const ulonglong2 *>(pa + 2);
                    const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(pb), b1 = *reinterpret_cast<const ulonglong2 *>(pb + 2);
                    i0 = make_ulonglong2(prod_canon(ar, M, a0.x, b0.x), prod_canon(ar, M, a0.y, b0.y));
                    i1 = make_ulonglong2(prod_canon(ar, M, a1.x, b1.x), prod_canon(ar, M, a1.y, b1.y));
                    u64 *po2 = stp.out + ((size_t)xq * stp.nl) * 32768 + off;
                    *reinterpret_cast<ulonglong2 *>(po2) = i0;
                    *reinterpret_cast<ulonglong2 *>(po2 + 2) = i1;
                } else {
                    i0 = *reinterpret_cast<const ulonglong2 *>(s[p] + e);
                    i1 = *reinterpret_cast<const ulonglong2 *>(s[p] + e + 2);
                }
                T c0 = ar.from_canon(i0.x), c1 = ar.from_canon(i0.y), c2 = ar.from_canon(i1.x), c3 = ar.from_canon(i1.y);
                ar.gs(c0, c1, W14a);
                ar.gs(c2, c3, W14b);
                ar.gs(c0, c2, W13);
                ar.gs(c1, c3, W13);
                ar.recentre_wide(c0); ar.recentre_wide(c1); ar.recentre_wide(c2); ar.recentre_wide(c3);
                lds[p][la] = A::to_bits(c0); lds[p][la + 1] = A::to_bits(c1);
                lds[p][la + 2] = A::to_bits(c2); lds[p][la + 3] = A::to_bits(c3);
            }
        }
        p2_sync<WGS>();
        p2_inverse_BA<A, NP, SWZ, WGS>(ar, tw, lds, d, t, B0, wt);
    }
}

// grid (16 chunks of 2048, (X/NP)*sel.n).  The body is a macro shared by the two kernels below: through a device function the NttStore
// kernel argument would be copied to scratch (2 KiB per lane: its per-limb tables are indexed dynamically) instead of being read with scalar
// loads from the kernel-argument segment.
// Loop A's last pass (ST 5, forward) walks the rotations FASTEST: the workgroups in flight then share ONE limb's digit tiles (3 digits x
// 16 KiB per chunk), which stay in L2 between rotations.  Limb-fastest (every other launch: its operands are per-polynomial), the
// 9.4 MB of digits were re-fetched through the fabric about every second time (PMC: 18.0 GB fetched for 12.3 GB of operands,
// profiles/r05/loop_a_pmc_before.txt).  HYDIA_LOOPA_LIMB_FASTEST restores the old order (stp.la.limb_fastest).
// (the unpadded image would let five two-polynomial workgroups share a CU instead of four: measured, no gain — 4.05 vs 4.08 ms for
// the merged epilogue's 23 launches of a 2^20 query — so the plain transforms keep the padded one)
#define P2_KERNEL_BODY(ST_, WGS_)                                                                                                         \
    constexpr int N = 32768;                                                                                                              \
    constexpr bool SWZ = false;                                                                                                           \
    __shared__ u64 lds[NP][P2Lds<SWZ>::SIZE];                                                                                             \
    __shared__ double twl[4][128];                                                                                                        \
    const int y = blockIdx.y;                                                                                                             \
    const bool rot_fastest = !INV && ST_ == 5 && !stp.la.limb_fastest;                                                                    \
    const int nxp = gridDim.y / nsl;                                                                                                      \
    const int xp = rot_fastest ? y % nxp : y / nsl, slot = slot0 + (rot_fastest ? y / nxp : y - xp * nsl), m = sel.mod[slot];            \
    const ModC M = T.mod[m];                                                                                                              \
    const bool fp = (T.fp_mask >> m) & 1u;                                                                                                \
    const ulonglong2 *__restrict__ tw = (fp ? (INV ? T.itwf : T.twf) : (INV ? T.itwp : T.twp)) + (size_t)m * N;                           \
    const int B0 = blockIdx.x * 2048;                                                                                                     \
    const u64 *s[NP];                                                                                                                     \
    u64 *d[NP];                                                                                                                           \
    _Pragma("unroll") for (int p = 0; p < NP; p++) {                                                                                      \
        s[p] = src + (size_t)(xp * NP + p) * so + (size_t)slot * N + B0;                                                                  \
        d[p] = dst + (size_t)(xp * NP + p) * dso + (size_t)slot * N + B0;                                                                 \
    }                                                                                                                                     \
    const double *const twg = (INV ? T.itwd : T.twd);                                                                                      \
    const P2Tab tab{(fp && twg && !T.no_tw_lds) ? twg + (size_t)m * N : nullptr, &twl[0][0]};                                              \
    if (fp) p2_body<FpA, INV, NP, ST_, SWZ, WGS_>(FpA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, xp * NP, slot, nullptr, nullptr, tab);   \
    else if ((T.pm_mask >> m) & 1u) p2_body<IntP, INV, NP, ST_, SWZ, WGS_>(IntP(M), tw, s, d, lds, threadIdx.x, B0, stp, M, xp * NP, slot); \
    else p2_body<IntA, INV, NP, ST_, SWZ, WGS_>(IntA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, xp * NP, slot);
template <bool INV, int NP, int ST>
// (loop A's fused inner product asks for three workgroups per CU: unbounded it takes 171 registers — two per CU, 6.24 ms per
// rotateQuery; at 167 it keeps its twelve loads per call in flight with three, 5.95 ms; capped to 128 it spills, 6.13 ms)
__global__ __launch_bounds__(256, (!INV && ST == 5) ? 3 : (!INV && ST == 10) ? 4 : 1) void k_ntt15_p2(NttTables T, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so,
                                                  size_t dso, LimbSel sel, int slot0, int nsl, NttStore stp) {
    P2_KERNEL_BODY(ST, P2_WGS_DEFAULT)
}
// the plain transform through round 4's workgroup-synchronous pass 2 (HYDIA_P2_WG_SYNC: parity variant) under its own name, so that the
// instantiations of k_ntt15_p2 keep theirs in every profile
template <bool INV, int NP>
__global__ __launch_bounds__(256) void k_ntt15_p2_wgsync(NttTables T, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so, size_t dso, LimbSel sel,
                                                          int slot0, int nsl, NttStore stp) {
    P2_KERNEL_BODY(0, true)
}

// second pass of the ModUp forward NTTs fused with the key-switching inner product: grid (16, nlimbs*X), x fastest so the
// workgroups that share a key tile follow each other.  Limb t = t0 + slot; the NP digits are all digits but the limb's own.
template <int NP, bool OWN, bool TAIL, bool SWZ = false>
DEV void p2_ip_workgroup(const NttTables &T, const u64 *__restrict__ dig, size_t dxs, int t, int x, const NttStore &stp, int bx,
                         u64 (*lds)[P2Lds<SWZ>::SIZE], double *twl) {
    constexpr int N = 32768;
    const int m = t < stp.ip.nl ? t : stp.ip.nT - stp.ip.nE + t;
    const ModC M = T.mod[m];
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.twf : T.twp) + (size_t)m * N;
    const int B0 = bx * 2048;
    const int own_d = OWN ? t / stp.ip.alpha : (1 << 30);
    const u64 *s[NP];
    u64 *d[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int dgt = p >= own_d ? p + 1 : p;
        s[p] = dig + (size_t)x * dxs + ((size_t)dgt * stp.ip.nE + t) * N + B0;
        d[p] = nullptr;
    }
    if (TAIL) {
        const ulonglong2 *__restrict__ itw = (fp ? T.itwf : T.itwp) + (size_t)m * N;
        u64 *dinv[2];
#pragma unroll
        for (int pp = 0; pp < 2; pp++)
            dinv[pp] = stp.ip.inv_out + (size_t)(2 * x + pp) * stp.ip.inv_outer + (size_t)(stp.ip.inv_row0 + t - stp.ip.nl) * N + B0;
        if (fp) p2_body<FpA, false, NP, 6, SWZ>(FpA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t, itw, dinv, P2Tab{(T.twd && !T.no_tw_lds) ? T.twd + (size_t)m * N : nullptr, twl});
        else if ((T.pm_mask >> m) & 1u) p2_body<IntP, false, NP, 6, SWZ>(IntP(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t, itw, dinv);
        else p2_body<IntA, false, NP, 6, SWZ>(IntA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t, itw, dinv);
        return;
    }
    if (fp) p2_body<FpA, false, NP, 4, SWZ>(FpA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t, nullptr, nullptr, P2Tab{(T.twd && !T.no_tw_lds) ? T.twd + (size_t)m * N : nullptr, twl});
    else if ((T.pm_mask >> m) & 1u) p2_body<IntP, false, NP, 4, SWZ>(IntP(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t);
    else p2_body<IntA, false, NP, 4, SWZ>(IntA(M), tw, s, d, lds, threadIdx.x, B0, stp, M, x, t);
}
template <int NP, bool OWN, bool TAIL = false>
__global__ __launch_bounds__(256) void k_ntt15_p2_ip(NttTables T, const u64 *__restrict__ dig, size_t dxs, int X, int t0, NttStore stp) {
    __shared__ u64 lds[(TAIL && NP < 2) ? 2 : NP][8 * 288];
    __shared__ double twl[4][128];
    const int slot = blockIdx.y / X, x = blockIdx.y - slot * X;  // x fastest: the workgroups that share a key tile follow each other
    p2_ip_workgroup<NP, OWN, TAIL>(T, dig, dxs, t0 + slot, x, stp, blockIdx.x, lds, &twl[0][0]);
}
// both halves of a relinearisation's fused inner product in ONE launch: the Q limbs (ND - 1 digits transformed, the limb's own digit
// read in place; FP64, HBM-bound) and the special-prime limbs with the inverse tail (ND digits; 60-bit integer, ALU-bound).  Rows of
// workgroups are ordered so that the two kinds run side by side on every CU: groups of xb = 8 ciphertexts, inside a group the nP
// special-prime rows spread evenly among the nl Q rows (Bresenham), inside a row x fastest (the 8 workgroups that share a key tile follow
// each other; a tile is re-fetched once per group, from the Infinity Cache).
// (two digits: held to four waves per SIMD — 128 registers; unbounded the folded reductions take 129 and the kernel loses 3 %)
template <int ND>
__global__ __launch_bounds__(256, ND == 2 ? 4 : 1) void k_ntt15_p2_ip_all(NttTables T, const u64 *__restrict__ dig, size_t dxs, int X, int xb, NttStore stp) {
    constexpr bool SWZ = ND >= 3;  // three (or four) images: the unpadded, XOR-swizzled form buys a workgroup per CU (P2Lds)
    __shared__ u64 lds[ND][P2Lds<SWZ>::SIZE];
    __shared__ double twl[4][128];
    const int nl = stp.ip.nl, nS = stp.ip.nE, nP = nS - nl;
    const int per = nS * xb, grp = blockIdx.y / per, r = blockIdx.y - grp * per;
    const int si = r / xb, x = grp * xb + (r - si * xb);
    const int pc = si * nP / nS, pc1 = (si + 1) * nP / nS;
    if (pc1 > pc) p2_ip_workgroup<ND, false, true, SWZ>(T, dig, dxs, nl + pc, x, stp, blockIdx.x, lds, &twl[0][0]);
    else if (si - pc == stp.ip.drop_l) p2_ip_workgroup<ND - 1, true, true, SWZ>(T, dig, dxs, si - pc, x, stp, blockIdx.x, lds, &twl[0][0]);  // (ND >= 2 images: enough for the tail)
    else p2_ip_workgroup<ND - 1, true, false, SWZ>(T, dig, dxs, si - pc, x, stp, blockIdx.x, lds, &twl[0][0]);
}


// ------------------------------------------------------------------------------------------------ one-pass transform (FP64 limbs)
// ONE HBM round trip per limb-polynomial instead of two: a 1024-thread workgroup keeps all 2^15 coefficients of one limb in
// registers (32 exact-integer doubles per lane) and runs the 15 stages as four register phases; LDS (132 KiB, two half-size
// rounds per exchange) only transposes between them.  With index bits A = 14..10, B = 9..5, C = 4..2, D = 1..0:
//   P1 stages 0-4    thread = (B, C, D)            registers = A     strided global access (stride 1024: coalesced), wave-uniform twiddles
//   P2 stages 5-9    thread = (A, C, D)            registers = B     31 twiddle pairs per thread that depend on A only
//   P3 stages 10-12  thread = (A, B)               registers = (C,D) radix-8 over C for the 4 values of D: 7 twiddle pairs
//   P4 stages 13-14  thread = bits 11..2           registers = (bits 14..12, D): eight groups of 4 consecutive coefficients, so
//                    global access at this end is 32 contiguous bytes per lane, lanes contiguous (the fused epilogues' operands too)
// The forward transform enters at P1 and leaves at P4 (+ the pass-2 epilogues, unchanged); the inverse runs P4' .. P1' with
// Gentleman-Sande butterflies and re-centres so that at most four stages run between reductions.  Same butterflies on the same
// operands as the two-pass path, so results are bit-identical.  Exchanges (X1: A<->B among threads with equal (C,D); X2: B<->(C,D)
// inside one A; X3: (A,B)<->P4 layout) are ds_write_b64 / ds_read_b64 on images whose row strides (1024, 33, 33 elements) keep
// every 16-lane write group and 32-lane read group on distinct banks.
constexpr int OP_LDS_ELEMS = 512 * 33;

// An index the compiler cannot see through: loads addressed with it cannot be hoisted above this point.  (The twiddle tables are
// const __restrict__, so the scheduler otherwise starts a later phase's twiddle loads phases early — 48 registers held across the
// exchanges, paid for with spills.)
DEV int op_pin(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
DEV FpA::TW op_tw(const FpA &, const ulonglong2 b) { return FpA::tw(b); }
DEV FpA::TW op_tw(const FpA &ar, const double wv) { return ar.tw8(wv); }
template <bool INV, class TWP>
DEV void op_stage_set(const FpA &ar, double (&v)[32], const TWP *__restrict__ tw, int base_shift, int hi, int nst, int first_bit) {
    // nst stages over register-index bits first_bit, first_bit-1, ...: stage s pairs (k, k + h), h = 1 << (first_bit - s),
    // twiddle tw[(base << s) + (hi << s) + (k >> (first_bit + 1 - s))] with base = 1 << base_shift
#pragma unroll
    for (int ss = 0; ss < nst; ss++) {
        const int s = INV ? nst - 1 - ss : ss;
        const int h = 1 << (first_bit - s);
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (!(k & h)) {
                const FpA::TW W = op_tw(ar, tw[(1 << (base_shift + s)) + (hi << s) + (k >> (first_bit + 1 - s))]);
                if (INV) ar.gs(v[k], v[k + h], W);
                else ar.ct(v[k], v[k + h], W);
            }
    }
}

// P2's twiddles depend on A only and a wave holds exactly two values of A (lanes 0-31 / 32-63): both candidates are fetched with
// scalar loads (wave-uniform addresses) and selected per lane half — 4 v_cndmask instead of a 16-byte vector load per twiddle,
// which would occupy the texture-address path for 16 cycles each
template <bool INV>
DEV void op_stage_set_p2(const FpA &ar, double (&v)[32], const ulonglong2 *__restrict__ tw, int A0, bool upper) {
#pragma unroll
    for (int ss = 0; ss < 5; ss++) {
        const int s = INV ? 4 - ss : ss;
        const int h = 16 >> s;
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (!(k & h)) {
                const ulonglong2 b0 = tw[(32 << s) + (A0 << s) + (k >> (5 - s))], b1 = tw[(32 << s) + ((A0 + 1) << s) + (k >> (5 - s))];
                const FpA::TW W = FpA::tw(make_ulonglong2(upper ? b1.x : b0.x, upper ? b1.y : b0.y));
                if (INV) ar.gs(v[k], v[k + h], W);
                else ar.ct(v[k], v[k + h], W);
            }
        if (INV && ss == 2) {
#pragma unroll
            for (int k = 0; k < 32; k++) ar.recentre(v[k]);
        }
    }
}

// raw operand of the first register phase and its conversion to a canonical residue (the fused prologues of p1_load, split so that
// the NEXT item's loads can be issued before the current item's last phase)
template <int LD>
DEV u64 op_raw_load(const NttLoad &ld, const u64 *s, int x, size_t idx) {
    return LD == 2 ? ld.y[(size_t)x * ld.y_outer + idx] : s[idx];
}
template <int LD>
DEV u64 op_raw_to_canon(const NttLoad &ld, const ModC *__restrict__ mod, const ModC &M, u64 v) {
    if (LD != 2) return v;
    const u64 ql = mod[ld.l].q;  // rescale spread: centred residue of the dropped limb's coefficient
    return v > (ql >> 1) ? negmod(reduce64(ql - v, M), M.q) : reduce64(v, M);
}

// One workgroup (1024 threads) per item = (limb slot, polynomial), slot-major, so that concurrently resident workgroups share a
// limb's twiddle table in L2.  (A persistent variant that walks items and prefetches the next item's operands before the last
// register phase was built and measured: the register allocator spills 300-900 bytes per lane inside the item loop and the
// transform gets 1.6x SLOWER — profiles/r02/ntt_one_pass.md.)
template <bool INV, int LD, int ST>
__global__ __launch_bounds__(1024) void k_ntt15_1p(NttTables T, const ulonglong2 *__restrict__ tw_pairs, const double *__restrict__ tw_single,
                                                   const ModC *__restrict__ modc, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so,
                                                   size_t dso, LimbSel sel, int slot0, int X, int nitems, ScaleSel scale, NttLoad ld, NttStore stp) {
    // tw_pairs / tw_single / modc are the tables of T again, as __restrict__ kernel parameters: inside the item loop only loads the
    // compiler knows cannot alias the loop's stores stay scalar (s_load) — through the struct they turn into per-lane vector loads
    constexpr int N = 32768;
    extern __shared__ u64 lds[];
    const int t = threadIdx.x;
    const bool lo = t < 512;            // wave-uniform: waves 0-7 / 8-15
    const int A = t >> 5, cd = t & 31;  // P2 / P3 thread coordinates (P3: t = (A, B))
    const int Ah = A & 15;
    const int A0 = __builtin_amdgcn_readfirstlane(t >> 6) << 1;  // the wave's first A (wave-uniform)
    const bool upperA = (t & 32) != 0;
    const int rd = (t >> 3) * 33 + (t & 7) * 4;  // P4 layout inside the X3 image
    const int item = blockIdx.x;
    if (item >= nitems) return;
    {
        const int sl = item / X, xp = item - sl * X, slot = slot0 + sl, m = sel.mod[slot];
        const ModC M = modc[m];
        const FpA ar(M);
        const ulonglong2 *__restrict__ tw = tw_pairs + (size_t)m * N;  // wave-uniform twiddles (scalar loads): pairs
        const double *__restrict__ twd = tw_single + (size_t)m * N;    // per-lane twiddles: 8 bytes each
        u64 *d = dst + (size_t)xp * dso + (size_t)slot * N;
        const u64 *s = src + (size_t)xp * so + (size_t)slot * N;
        double v[32], w[32];
        if (!INV) {
            // ---- P1
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = ar.from_canon(op_raw_to_canon<LD>(ld, modc, M, op_raw_load<LD>(ld, s, xp, (size_t)k * 1024 + t)));
            op_stage_set<false>(ar, v, tw, 0, 0, 5, 4);
            // ---- X1: element (a, b, cd) at a*1024 + b*32 + cd; halves of a
#pragma unroll
            for (int k = 0; k < 16; k++) lds[k * 1024 + t] = FpA::to_bits(v[k]);
            __syncthreads();
            if (lo) {
#pragma unroll
                for (int b = 0; b < 32; b++) w[b] = FpA::from_bits(lds[Ah * 1024 + b * 32 + cd]);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) lds[k * 1024 + t] = FpA::to_bits(v[16 + k]);
            __syncthreads();
            if (!lo) {
#pragma unroll
                for (int b = 0; b < 32; b++) w[b] = FpA::from_bits(lds[Ah * 1024 + b * 32 + cd]);
            }
            // ---- P2
            op_stage_set_p2<false>(ar, w, tw, A0, upperA);
            // ---- X2 (inside one A): element (A, b, cd) at ((A & 15) * 32 + b) * 33 + cd; halves of A = halves of the waves.  The
            // exchange is IN PLACE in w (a wave is either writer+reader of a round or idle in it), so 32 values are live, not 64
#pragma unroll
            for (int h = 0; h < 2; h++) {
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int b = 0; b < 32; b++) lds[(Ah * 32 + b) * 33 + cd] = FpA::to_bits(w[b]);
                }
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int c = 0; c < 32; c++) w[c] = FpA::from_bits(lds[(Ah * 32 + cd) * 33 + c]);  // now thread (A, B = cd)
                }
            }
            // ---- P3: t = (A, B); registers (C, D); stages 10-12 over C = register bits 4..2
            op_stage_set<false>(ar, w, twd, 10, op_pin(t), 3, 4);
            // ---- X3 + P4: thread t = index bits 11..2; group f = bits 14..12; element i at ((i >> 5) & 511) * 33 + (i & 31)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int c = 0; c < 32; c++) lds[(t & 511) * 33 + c] = FpA::to_bits(w[c]);
                }
                __syncthreads();
                // the round's twiddles, then (last round) the NEXT item's operands: every register value of this item now sits
                // in LDS, and vector loads return in issue order, so the twiddles do not wait behind the prefetch
                double W13[4], W14a[4], W14b[4];
                const int tp = op_pin(t);
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const int gi = (4 * h + f) * 1024 + tp;
                    const double2 w14 = *reinterpret_cast<const double2 *>(twd + 16384 + 2 * gi);
                    W13[f] = twd[8192 + gi];
                    W14a[f] = w14.x;
                    W14b[f] = w14.y;
                }
                P2Pre pre = {};
                if (ST != 0) pre = p2_prefetch<ST>(stp, ar, xp, slot, (unsigned)((4 * h) * 4096 + 4 * t), M);
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const int gi = (4 * h + f) * 1024 + t;
                    P2Pre nxt = {};
                    if (ST != 0 && f < 3) nxt = p2_prefetch<ST>(stp, ar, xp, slot, (unsigned)(4 * (gi + 1024)), M);
                    // one group at a time out of LDS: the other half's 32 values are still waiting in registers
                    double c0 = FpA::from_bits(lds[f * 128 * 33 + rd]), c1 = FpA::from_bits(lds[f * 128 * 33 + rd + 1]),
                           c2 = FpA::from_bits(lds[f * 128 * 33 + rd + 2]), c3 = FpA::from_bits(lds[f * 128 * 33 + rd + 3]);
                    const FpA::TW T13 = ar.tw8(W13[f]);
                    ar.ct(c0, c2, T13);
                    ar.ct(c1, c3, T13);
                    ar.ct(c0, c1, ar.tw8(W14a[f]));
                    ar.ct(c2, c3, ar.tw8(W14b[f]));
                    const u64 vv[4] = {ar.fin_fwd(c0), ar.fin_fwd(c1), ar.fin_fwd(c2), ar.fin_fwd(c3)};
                    if (ST == 0) {
                        *reinterpret_cast<ulonglong2 *>(d + 4 * gi) = make_ulonglong2(vv[0], vv[1]);
                        *reinterpret_cast<ulonglong2 *>(d + 4 * gi + 2) = make_ulonglong2(vv[2], vv[3]);
                    } else {
                        p2_finish<ST>(stp, M, xp, slot, (unsigned)(4 * gi), vv, pre, P2Prod{});  // (the one-pass kernel takes no fused product)
                        pre = nxt;
                    }
                }
            }
        } else {
            // ---- P4': eight groups of 4 consecutive coefficients; strides 1, 2
            double x[8][4];
            const int tq = op_pin(t);
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int gi = f * 1024 + tq;
                const ulonglong2 i0 = *reinterpret_cast<const ulonglong2 *>(s + 4 * gi), i1 = *reinterpret_cast<const ulonglong2 *>(s + 4 * gi + 2);
                const double2 w14 = *reinterpret_cast<const double2 *>(twd + 16384 + 2 * gi);
                const FpA::TW W13 = ar.tw8(twd[8192 + gi]), W14a = ar.tw8(w14.x), W14b = ar.tw8(w14.y);
                double c0 = ar.from_canon(i0.x), c1 = ar.from_canon(i0.y), c2 = ar.from_canon(i1.x), c3 = ar.from_canon(i1.y);
                ar.gs(c0, c1, W14a);
                ar.gs(c2, c3, W14b);
                ar.gs(c0, c2, W13);
                ar.gs(c1, c3, W13);
                ar.recentre(c0); ar.recentre(c1); ar.recentre(c2); ar.recentre(c3);
                x[f][0] = c0; x[f][1] = c1; x[f][2] = c2; x[f][3] = c3;
            }
            // ---- X3': P4 layout -> thread (A, B) with registers (C, D)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (h) __syncthreads();
#pragma unroll
                for (int f = 0; f < 4; f++)
#pragma unroll
                    for (int D = 0; D < 4; D++) lds[f * 128 * 33 + rd + D] = FpA::to_bits(x[4 * h + f][D]);
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int c = 0; c < 32; c++) v[c] = FpA::from_bits(lds[(t & 511) * 33 + c]);
                }
            }
            // ---- P3': strides 4, 8, 16
            op_stage_set<true>(ar, v, twd, 10, op_pin(t), 3, 4);
#pragma unroll
            for (int c = 0; c < 32; c++) ar.recentre(v[c]);
            // ---- X2' (inside one A, in place): thread (A, B = cd) registers (C, D) -> thread (A, (C,D) = cd) registers B
#pragma unroll
            for (int h = 0; h < 2; h++) {
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int c = 0; c < 32; c++) lds[(Ah * 32 + cd) * 33 + c] = FpA::to_bits(v[c]);
                }
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int b = 0; b < 32; b++) v[b] = FpA::from_bits(lds[(Ah * 32 + b) * 33 + cd]);
                }
            }
            // ---- P2': strides 32 .. 512 (three stages, re-centre, two stages)
            op_stage_set_p2<true>(ar, v, tw, A0, upperA);
            // ---- X1': thread (A, cd) registers B -> thread (B, cd) registers A; halves of A = halves of the waves on the write side
#pragma unroll
            for (int h = 0; h < 2; h++) {
                __syncthreads();
                if (lo == (h == 0)) {
#pragma unroll
                    for (int b = 0; b < 32; b++) lds[Ah * 1024 + b * 32 + cd] = FpA::to_bits(v[b]);
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; k++) w[16 * h + k] = FpA::from_bits(lds[k * 1024 + t]);
            }
            // ---- P1': strides 1024 .. 16384 (two stages, re-centre, three stages), scale, store
#pragma unroll
            for (int ss = 0; ss < 5; ss++) {
                const int st5 = 4 - ss, h = 16 >> st5;
#pragma unroll
                for (int k = 0; k < 32; k++)
                    if (!(k & h)) ar.gs(w[k], w[k + h], FpA::tw(tw[(1 << st5) + (k >> (5 - st5))]));
                if (ss == 1) {
#pragma unroll
                    for (int k = 0; k < 32; k++) ar.recentre(w[k]);
                }
            }
            const u64 sc = scale.s[slot], scs = scale.s_sh[slot];
#pragma unroll
            for (int k = 0; k < 32; k++) d[(size_t)k * 1024 + t] = ar.fin_inv(w[k], sc, scs);
        }
    }
}

}  // namespace

namespace hk {

bool ntt15_p2_inner_product(hipStream_t st, const NttTables &T, const ModC *mod, const u64 *dig, size_t dxs, int nd, int X, int nl,
                            int nP, int nT, int alpha, const u64 *const *keys, const u64 *key, const u64 *c2, size_t c2_xs, u64 *acc,
                            u64 *inv_out, size_t inv_outer, int inv_row0, const DropLimb *drop, bool per_x_keys) {
    const int nE = nl + nP;
    NttStore stp{};
    stp.mode = 4;
    stp.ip.key = key;
    stp.ip.keys = per_x_keys ? keys : nullptr;  // device array: the key of ciphertext x (giant steps); else `key` serves every x
    stp.ip.nT = nT;
    stp.ip.nE = nE;
    stp.ip.nl = nl;
    stp.ip.alpha = alpha;
    stp.ip.own = 1;
    stp.ip.c2 = c2;
    stp.ip.c2_xs = c2_xs;
    stp.ip.acc = acc;
    stp.ip.inv_out = inv_out;
    stp.ip.inv_outer = inv_outer;
    stp.ip.inv_row0 = inv_row0;
    stp.ip.drop_l = -1;
    // Q limbs: NP = nd - 1 pass-1 digits in (+ the limb's own residues), two accumulator rows out; special-prime limbs: nd digits in, two
    // rows out; the key tiles are shared by all x (L2)
    const double bytes_q = nd >= 2 ? ((nd - 1) + 1 + 2.0) * nl * X * 262144.0 : 0.0, bytes_p = (nd + 2.0) * nP * X * 262144.0;
    const bool merged = inv_out && nd >= 2 && nd <= 4 && !T.two_ip_launches;
    char name[64];
    if (merged) {
        if (drop && !T.no_drop_in_ip && inv_row0 >= 1 && drop->l == nl - 1) {  // the dropped limb rides the tail (one launch and a round trip of that row less)
            stp.ip.drop_l = drop->l;
            stp.ip.drop_dbl = drop->dbl;
            stp.ip.drop_mul = drop->mul;
            stp.ip.drop_mul_sh = drop->mul_sh;
            stp.ip.drop_add = drop->add;
            stp.ip.drop_add_x = drop->add_x;
            stp.ip.drop_add_p = drop->add_p;
            stp.ip.drop_has_prod = drop->prod ? 1 : 0;
            if (drop->prod) stp.ip.drop_prod = *drop->prod;
        }
        snprintf(name, sizeof name, "k_ntt15_p2_ip_all<%d>", nd);
        ledger_add(name, bytes_q + bytes_p);
        int xb = T.ip_group;  // ciphertexts per interleaving group (HYDIA_IP_GROUP; must divide X)
        if (xb < 1 || xb > X || X % xb) xb = X;
        if (nd == 2) hipLaunchKernelGGL((k_ntt15_p2_ip_all<2>), dim3(16, (nl + nP) * X), dim3(256), 0, st, T, dig, dxs, X, xb, stp);
        else if (nd == 3) hipLaunchKernelGGL((k_ntt15_p2_ip_all<3>), dim3(16, (nl + nP) * X), dim3(256), 0, st, T, dig, dxs, X, xb, stp);
        else hipLaunchKernelGGL((k_ntt15_p2_ip_all<4>), dim3(16, (nl + nP) * X), dim3(256), 0, st, T, dig, dxs, X, xb, stp);
        return stp.ip.drop_l >= 0;
    }
    if (nd >= 2) {
        snprintf(name, sizeof name, "k_ntt15_p2_ip<%d, true, false>", nd - 1 > 3 ? 3 : nd - 1);
        ledger_add(name, bytes_q);
    }
    snprintf(name, sizeof name, inv_out ? "k_ntt15_p2_ip<%d, false, true>" : "k_ntt15_p2_ip<%d, false, false>", nd > 4 ? 4 : nd);
    ledger_add(name, bytes_p);
    // Q limbs: nd - 1 digits are transformed, the limb's own digit is read from c2
    if (nd == 1) {
        LimbSel qs{};
        qs.n = nl;
        for (int j = 0; j < nl; j++) qs.mod[j] = j;
        inner_product(st, mod, 32768, dig, dxs, nd, keys, per_x_keys ? 0 : 1, nT, acc, X, qs, c2, c2_xs, alpha, nl, nE);
    } else if (nd == 2) {
        hipLaunchKernelGGL((k_ntt15_p2_ip<1, true>), dim3(16, nl * X), dim3(256), 0, st, T, dig, dxs, X, 0, stp);
    } else if (nd == 3) {
        hipLaunchKernelGGL((k_ntt15_p2_ip<2, true>), dim3(16, nl * X), dim3(256), 0, st, T, dig, dxs, X, 0, stp);
    } else {
        hipLaunchKernelGGL((k_ntt15_p2_ip<3, true>), dim3(16, nl * X), dim3(256), 0, st, T, dig, dxs, X, 0, stp);
    }
    // P limbs: every digit is transformed
    if (inv_out) {
        if (nd == 1) hipLaunchKernelGGL((k_ntt15_p2_ip<1, false, true>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
        else if (nd == 2) hipLaunchKernelGGL((k_ntt15_p2_ip<2, false, true>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
        else if (nd == 3) hipLaunchKernelGGL((k_ntt15_p2_ip<3, false, true>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
        else hipLaunchKernelGGL((k_ntt15_p2_ip<4, false, true>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
        return false;
    }
    if (nd == 1) hipLaunchKernelGGL((k_ntt15_p2_ip<1, false>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
    else if (nd == 2) hipLaunchKernelGGL((k_ntt15_p2_ip<2, false>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
    else if (nd == 3) hipLaunchKernelGGL((k_ntt15_p2_ip<3, false>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
    else hipLaunchKernelGGL((k_ntt15_p2_ip<4, false>), dim3(16, nP * X), dim3(256), 0, st, T, dig, dxs, X, nl, stp);
    return false;
}

// ---- launch plumbing.  With the one-pass kernel enabled (HYDIA_NTT_1PASS) a LimbSel is cut into maximal runs of slots whose moduli
// take the same arithmetic path: FP64 runs go to the one-pass kernel (not for the fused first-pass base conversion, which it does
// not implement), integer runs — the 60-bit q_0 and the special primes — to the two-pass kernels restricted to that slot range.
// The switches are read once per context (NttTables::one_pass*).  HYDIA_NTT_1PASS_MIN = smallest number of limb-polynomials in a launch for which
// the one-pass kernel is used: it needs one whole workgroup of 1024 threads per limb-polynomial, so small batches (the per-query
// fixed-cost work) fill the GPU better with the two-pass kernels' 24 smaller workgroups per limb-polynomial
// Measured on MI355X (profiles/r02/ntt_one_pass.md): the one-pass kernel is 1.15x faster than the two passes on plain transforms
// streamed from HBM in launches of thousands of limb-polynomials, slower on small or Infinity-Cache-resident batches and on the
// fused epilogues; inside a query it loses everywhere it was tried (whole pipeline +1.2 ms at 2^20; loop A's ModDown transforms
// alone +0.35 ms per query).  So it is OFF by default.  HYDIA_NTT_1PASS = every FP64 transform of at least HYDIA_NTT_1PASS_MIN
// limb-polynomials (default 1024); HYDIA_NTT_1PASS=plain restricts it to transforms without fused prologue / epilogue.
static bool use_one_pass(const NttTables &T, bool inv, int ld, int st, int items) {
    (void)inv;
    if (!T.one_pass) return false;
    if (T.one_pass == 2 && !(ld == 0 && st == 0)) return false;  // "plain": only transforms without a fused prologue / epilogue
    return items >= T.one_pass_min;
}
template <class F>
static void for_slot_runs(const NttTables &T, const LimbSel &sel, bool split, F fn) {
    if (!split) {
        fn(0, sel.n, false);
        return;
    }
    for (int s0 = 0; s0 < sel.n;) {
        const bool fp = (T.fp_mask >> sel.mod[s0]) & 1u;
        int s1 = s0 + 1;
        while (s1 < sel.n && (((T.fp_mask >> sel.mod[s1]) & 1u) != 0) == fp) s1++;
        fn(s0, s1 - s0, fp);
        s0 = s1;
    }
}
template <bool INV, int LD, int ST>
static void launch_1p(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &sel,
                      int slot0, int nsl, const ScaleSel &scale, const NttLoad &ld, const NttStore &stp) {
    static bool attr_done[64] = {};  // > 64 KiB of dynamic LDS has to be granted per kernel and per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt15_1p<INV, LD, ST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  OP_LDS_ELEMS * (int)sizeof(u64));
        attr_done[dev] = true;
    }
    const int nitems = X * nsl;
    {
        char name[64];
        snprintf(name, sizeof name, "k_ntt15_1p<%s, %d, %d>", INV ? "true" : "false", LD, ST);
        ledger_add(name, (2.0 + (ST == 1 ? 1.5 : ST == 2 ? 1.0 : ST == 3 ? 2.0 : ST == 9 ? 3.0 : ST == 10 ? 4.0 : ST == 11 ? 4.0 : 0.0)) * nitems * 262144.0);
    }
    hipLaunchKernelGGL((k_ntt15_1p<INV, LD, ST>), dim3(nitems), dim3(1024), OP_LDS_ELEMS * sizeof(u64), st, T,
                       INV ? T.itwf : T.twf, INV ? T.itwd : T.twd, T.mod, src, dst, so, dso, sel, slot0, X, nitems, scale, ld, stp);
}
template <int LD>
static void launch_p1_fwd(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                          const LimbSel &sel, int slot0, int nsl, const NttLoad &ld) {
    ScaleSel dummy = {};
    ledger_add(LD == 0 ? "k_ntt15_p1<false, 0>" : "k_ntt15_p1<false, 2>",
               (LD == 2 ? 1.0 + 1.0 / (nsl > 0 ? nsl : 1) : 2.0) * X * nsl * 262144.0);  // LD 2 reads ONE dropped limb per polynomial
    hipLaunchKernelGGL((k_ntt15_p1<false, LD>), dim3(8, X * nsl), dim3(256), 0, st, T, src, dst, so, dso, sel, slot0, nsl, dummy, ld);
}
// two polynomials per workgroup share the twiddle loads.  Small launches (below 4 workgroups per CU when paired — the per-query
// fixed-cost tail) run one polynomial per workgroup: twice the workgroups, half the serial work in each
static bool pair_polys(int X, int nsl) { return X % 2 == 0 && (X / 2) * nsl * 16 >= 1024; }
template <int ST>
static void launch_p2_fwd(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel, int slot0, int nsl,
                          const NttStore &stp) {
    {   // pass-1 output in, result out, + the epilogue's operands: acc & addend (1), rescale input (2), both + subtrahend (3)
        char name[64];
        snprintf(name, sizeof name, "k_ntt15_p2<false, %d, %d>", pair_polys(X, nsl) ? 2 : 1, ST);
        double per = (2.0 + (ST == 1 ? 1.5 : ST == 2 ? 1.0 : ST == 3 ? 2.0 : ST == 9 ? 3.0 : ST == 10 ? 4.0 : ST == 11 ? 4.0 : 0.0)) * 262144.0;
        if (ST == 5)  // pass-1 output in, result out, addend on every other polynomial, nd key rows (6- or 8-byte residues); digits from L2
            per = 2.5 * 262144.0 + stp.la.nd * 32768.0 * (stp.la.packed_nQ > 0 ? (6.0 * (nsl - 1) + 8.0) / nsl : 8.0);
        ledger_add(name, per * X * nsl);
    }
    NttStore sv = stp;
    sv.int_epilogue = T.int_epilogue;
    if (ST == 5) sv.la.raw_fp = (stp.la.fp && stp.la.premul && !stp.dbl && !T.int_epilogue && T.twf != nullptr) ? 1 : 0;
    if (ST == 0 && T.p2_wg_sync) {  // parity variant of the plain transform: round 4's workgroup-synchronous pass 2
        if (pair_polys(X, nsl))
            hipLaunchKernelGGL((k_ntt15_p2_wgsync<false, 2>), dim3(16, (X / 2) * nsl), dim3(256), 0, st, T, dst, dst, dso, dso, sel, slot0, nsl, sv);
        else
            hipLaunchKernelGGL((k_ntt15_p2_wgsync<false, 1>), dim3(16, X * nsl), dim3(256), 0, st, T, dst, dst, dso, dso, sel, slot0, nsl, sv);
        return;
    }
    if (pair_polys(X, nsl))
        hipLaunchKernelGGL((k_ntt15_p2<false, 2, ST>), dim3(16, (X / 2) * nsl), dim3(256), 0, st, T, dst, dst, dso, dso, sel, slot0, nsl, sv);
    else
        hipLaunchKernelGGL((k_ntt15_p2<false, 1, ST>), dim3(16, X * nsl), dim3(256), 0, st, T, dst, dst, dso, dso, sel, slot0, nsl, sv);
}
template <int LD, int ST>
static void forward_runs(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &sel,
                         const NttLoad &ld, const NttStore &stp) {
    const bool split = T.fp_mask != 0 && use_one_pass(T, false, LD, ST, X * sel.n);
    for_slot_runs(T, sel, split, [&](int s0, int n, bool fp) {
        if (fp && split && use_one_pass(T, false, LD, ST, X * n)) {
            ScaleSel dummy = {};
            launch_1p<false, LD, ST>(st, T, src, dst, so, dso, X, sel, s0, n, dummy, ld, stp);
        } else {
            launch_p1_fwd<LD>(st, T, src, dst, so, dso, X, sel, s0, n, ld);
            launch_p2_fwd<ST>(st, T, dst, dso, X, sel, s0, n, stp);
        }
    });
}

// element (x, slot) at base + x*outer + slot*N.  When X is even, pass 2 transforms polynomials 2x', 2x'+1 together.
void ntt15_forward(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                   const LimbSel &sel) {
    NttLoad ld{};
    NttStore stp{};
    forward_runs<0, 0>(st, T, src, dst, so, dso, X, sel, ld, stp);
}
// first pass only (the caller's second pass is fused with the key-switching inner product): always the two-pass kernels
void ntt15_forward_p1(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &sel) {
    NttLoad ld{};
    launch_p1_fwd<0>(st, T, src, dst, so, dso, X, sel, 0, sel.n, ld);
}
void ntt15_forward_p2(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel) {
    NttStore stp{};
    launch_p2_fwd<0>(st, T, dst, dso, X, sel, 0, sel.n, stp);
}
// second pass alone with a fused epilogue (the first pass ran inside the column-fused conversion)
void ntt15_forward_p2_fused(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel, const NttStore &stp) {
    if (stp.mode == 1) launch_p2_fwd<1>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 5) launch_p2_fwd<5>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 2) launch_p2_fwd<2>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 3 && stp.has_prod && stp.prod.c && !stp.sub) launch_p2_fwd<11>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 3 && stp.has_prod && stp.sub) launch_p2_fwd<10>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 3 && stp.has_prod) launch_p2_fwd<9>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else if (stp.mode == 3) launch_p2_fwd<3>(st, T, dst, dso, X, sel, 0, sel.n, stp);
    else launch_p2_fwd<0>(st, T, dst, dso, X, sel, 0, sel.n, stp);
}
void ntt15_forward_fused(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                         const LimbSel &sel, const NttLoad &ld, const NttStore &stp) {
    if (ld.mode == 2) {
        if (stp.mode == 2) forward_runs<2, 2>(st, T, src, dst, so, dso, X, sel, ld, stp);
        else forward_runs<2, 0>(st, T, src, dst, so, dso, X, sel, ld, stp);
        return;
    }
    if (stp.mode == 1) forward_runs<0, 1>(st, T, src, dst, so, dso, X, sel, ld, stp);
    else if (stp.mode == 5) forward_runs<0, 5>(st, T, src, dst, so, dso, X, sel, ld, stp);
    else if (stp.mode == 2) forward_runs<0, 2>(st, T, src, dst, so, dso, X, sel, ld, stp);
    else if (stp.mode == 3) forward_runs<0, 3>(st, T, src, dst, so, dso, X, sel, ld, stp);
    else forward_runs<0, 0>(st, T, src, dst, so, dso, X, sel, ld, stp);
}
void ntt15_inverse_loop_a(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel, const ScaleSel &scale,
                          const LoopAIp &la, bool p1) {
    NttLoad ld{};
    NttStore stp{};
    stp.mode = 5;
    stp.la = la;
    {   // per limb-polynomial: nd key rows (8-byte residues: special primes; 6-byte where packed) in, raw pass-2' image out
        double keyb = 0;
        for (int s = 0; s < sel.n; s++) {
            const int j = la.key_row0 + s;
            keyb += la.nd * 32768.0 * ((la.packed_nQ > 0 && j > 0 && j < la.packed_nQ) ? 6.0 : 8.0);
        }
        ledger_add((X % 2 == 0) ? "k_ntt15_p2<true, 2, 5>" : "k_ntt15_p2<true, 1, 5>", (double)X * (keyb + sel.n * 262144.0));
        if (p1) ledger_add("k_ntt15_p1<true, 0>", 2.0 * X * sel.n * 262144.0);
    }
    if (X % 2 == 0)
        hipLaunchKernelGGL((k_ntt15_p2<true, 2, 5>), dim3(16, (X / 2) * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, 0, sel.n, stp);
    else
        hipLaunchKernelGGL((k_ntt15_p2<true, 1, 5>), dim3(16, X * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, 0, sel.n, stp);
    // p1 = false: the column-fused conversion that follows runs pass 1' itself
    if (p1) hipLaunchKernelGGL((k_ntt15_p1<true, 0>), dim3(8, X * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, 0, sel.n, scale, ld);
}
void ntt15_inverse_p2(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X, const LimbSel &sel) {
    NttStore stp{};
    const bool pair = pair_polys(X, sel.n);
    ledger_add(pair ? "k_ntt15_p2<true, 2, 0>" : "k_ntt15_p2<true, 1, 0>", 2.0 * X * sel.n * 262144.0);
    if (pair)
        hipLaunchKernelGGL((k_ntt15_p2<true, 2, 0>), dim3(16, (X / 2) * sel.n), dim3(256), 0, st, T, src, dst, so, dso, sel, 0, sel.n, stp);
    else
        hipLaunchKernelGGL((k_ntt15_p2<true, 1, 0>), dim3(16, X * sel.n), dim3(256), 0, st, T, src, dst, so, dso, sel, 0, sel.n, stp);
}
void ntt15_inverse_p2_prod(hipStream_t st, const NttTables &T, const ProdSrc &ps, u64 *dst, size_t dso, int X, const LimbSel &sel, u64 *d2_out) {
    NttStore stp{};
    stp.mode = 8;
    stp.has_prod = 1;
    stp.prod = ps;
    stp.out = d2_out;
    stp.nl = sel.n;
    const bool pair = pair_polys(X, sel.n);
    ledger_add(pair ? "k_ntt15_p2<true, 2, 8>" : "k_ntt15_p2<true, 1, 8>", 4.0 * X * sel.n * 262144.0);  // a1, b1 in; d2 and the raw image out
    if (pair)
        hipLaunchKernelGGL((k_ntt15_p2<true, 2, 8>), dim3(16, (X / 2) * sel.n), dim3(256), 0, st, T, (const u64 *)nullptr, dst, (size_t)0, dso, sel, 0, sel.n, stp);
    else
        hipLaunchKernelGGL((k_ntt15_p2<true, 1, 8>), dim3(16, X * sel.n), dim3(256), 0, st, T, (const u64 *)nullptr, dst, (size_t)0, dso, sel, 0, sel.n, stp);
}
// first inverse pass of limb l of (acc P^{-1} + addend)(x2) — k_moddown_last_limb's arithmetic in the load (relin + rescale tail)
void ntt15_inverse_p2_last_limb(hipStream_t st, const NttTables &T, const u64 *acc_l, u64 *dst, size_t so, size_t dso, int XP, int l,
                                u64 pinv, u64 pinv_sh, const u64 *addend, size_t add_x, size_t add_p, int dbl) {
    NttStore stp{};
    stp.mode = 7;
    stp.nl = l;
    stp.mul.s[0] = pinv;
    stp.mul.s_sh[0] = pinv_sh;
    stp.addend = addend;
    stp.add_x = add_x;
    stp.add_p = add_p;
    stp.dbl = dbl;
    LimbSel sel{};
    sel.n = 1;
    sel.mod[0] = l;
    const bool pair = pair_polys(XP, 1);
    ledger_add(pair ? "k_ntt15_p2<true, 2, 7>" : "k_ntt15_p2<true, 1, 7>", 3.0 * XP * 262144.0);
    if (pair)
        hipLaunchKernelGGL((k_ntt15_p2<true, 2, 7>), dim3(16, XP / 2), dim3(256), 0, st, T, acc_l, dst, so, dso, sel, 0, 1, stp);
    else
        hipLaunchKernelGGL((k_ntt15_p2<true, 1, 7>), dim3(16, XP), dim3(256), 0, st, T, acc_l, dst, so, dso, sel, 0, 1, stp);
}
void ntt15_inverse_p1(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel, const ScaleSel &scale) {
    NttLoad ld{};
    if (ntt15_inverse_p1_narrow(st, T, dst, dso, X, sel, 0, sel.n, scale)) return;
    ledger_add("k_ntt15_p1<true, 0>", 2.0 * X * sel.n * 262144.0);
    hipLaunchKernelGGL((k_ntt15_p1<true, 0>), dim3(8, X * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, 0, sel.n, scale, ld);
}
void ntt15_inverse(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                   const LimbSel &sel, const ScaleSel &scale) {
    NttLoad ld{};
    NttStore stp{};
    const bool split = T.fp_mask != 0 && use_one_pass(T, true, 0, 0, X * sel.n);
    for_slot_runs(T, sel, split, [&](int s0, int n, bool fp) {
        if (fp && split && use_one_pass(T, true, 0, 0, X * n)) {
            launch_1p<true, 0, 0>(st, T, src, dst, so, dso, X, sel, s0, n, scale, ld, stp);
            return;
        }
        ledger_add(pair_polys(X, n) ? "k_ntt15_p2<true, 2, 0>" : "k_ntt15_p2<true, 1, 0>", 2.0 * X * n * 262144.0);
        if (T.p2_wg_sync) {
            if (pair_polys(X, n))
                hipLaunchKernelGGL((k_ntt15_p2_wgsync<true, 2>), dim3(16, (X / 2) * n), dim3(256), 0, st, T, src, dst, so, dso, sel, s0, n, stp);
            else
                hipLaunchKernelGGL((k_ntt15_p2_wgsync<true, 1>), dim3(16, X * n), dim3(256), 0, st, T, src, dst, so, dso, sel, s0, n, stp);
        } else if (pair_polys(X, n))
            hipLaunchKernelGGL((k_ntt15_p2<true, 2, 0>), dim3(16, (X / 2) * n), dim3(256), 0, st, T, src, dst, so, dso, sel, s0, n, stp);
        else
            hipLaunchKernelGGL((k_ntt15_p2<true, 1, 0>), dim3(16, X * n), dim3(256), 0, st, T, src, dst, so, dso, sel, s0, n, stp);
        if (ntt15_inverse_p1_narrow(st, T, dst, dso, X, sel, s0, n, scale)) return;  // small launch: 16-column tiles
        ledger_add("k_ntt15_p1<true, 0>", 2.0 * X * n * 262144.0);
        hipLaunchKernelGGL((k_ntt15_p1<true, 0>), dim3(8, X * n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, s0, n, scale, ld);
    });
}

}  // namespace hk

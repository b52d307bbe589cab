// image_matching_amd/csrc/colfuse.hip — column-fused base conversion of hybrid key switching for N = 2^15 on gfx950:
// inverse pass 1' of the sources, fast base conversion, forward pass 1 of the targets in ONE kernel (no HBM round trip in between).
// Replaces, bit for bit, k_ntt15_p1<true> + k_base_convert / k_base_convert_digits / k_moddown_rescale_conv + k_ntt15_p1<false> of
// every ModUp and ModDown: the hoisted rotations of DiagonalSender::computeSimilarity (/root/reference/src/sender/sender_diag.cpp:22-26),
// RelinearizeInPlace + RescaleInPlace (:79-80) and the ct x ct products of chebyshevCompare (src/openFHE_wrapper.cpp:143-185).
#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "kernels.h"
#include "ntt_arith.h"

namespace {

// ------------------------------------------------------------------------------------------------ column-fused conversion
// Every base conversion of hybrid key switching sits between two transforms: the sources leave the evaluation domain (inverse pass 2',
// then pass 1'), are converted coefficient by coefficient, and the targets enter it again (pass 1, then pass 2).  Pass 1', the
// conversion and pass 1 all act on COLUMNS of the 128 x 256 coefficient matrix, so the workgroup that owns 32 adjacent columns runs
// all three back to back: the sources' coefficient-form values stay in registers (16 rows x <= 4 sources per lane), each target is
// converted straight into the forward butterflies' operands, and neither the coefficient-form sources nor the converted rows ever
// exist in HBM (loop A: 8 of its 29 GiB; every ModUp / ModDown of the comparator: two round trips per row and two launches).
// Same butterflies, same conversion sums, same final reductions as k_ntt15_p1<true> + k_base_convert[_digits] /
// k_moddown_rescale_conv + k_ntt15_p1<false>: bit-identical (HYDIA_NO_COLFUSE runs those instead).
// LDS: two 32 KiB exchange images used alternately (ONE barrier per tile transform), the sources' phase-B twiddles, two alternating
// sets for the targets.  256 registers per lane -> two workgroups per CU; every global load of a workgroup is issued up front.
// Merged ModDown + Rescale (MDR) carries a fifth value per row, the dropped limb's centred residue: it lives in the second image's
// LDS (one slot per lane and row) instead of 32 more registers — 160 + working registers spilled to scratch (1.43 against 2.1 TB/s) —
// and that variant exchanges through ONE image with a second barrier per transform.
constexpr int CF_LDS_BYTES = 2 * 128 * 32 * 8 + (HY_CF_SRC + 1 + 2) * 128 * 16;  // (the wide kernel serves four sources)

// y: in = raw pass-2' values of rows 8h + l (h = g + 8 hh) at index 8 hh + l; out = canonical coefficient-form residues (times sc) of
// rows g + 8k at index k
template <class A>
DEV void cf_inverse(const A ar, const ulonglong2 *__restrict__ tw, const ulonglong2 *ltw, u64 *lds, int g, int col, u64 sc, u64 scs,
                    u64 (&y)[16]) {
    typedef typename A::T T;
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
        const int h = g + 8 * hh;
        T w[8];
#pragma unroll
        for (int l = 0; l < 8; l++) w[l] = A::from_bits(y[8 * hh + l]);
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
    T v[16];
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
    for (int k = 0; k < 16; k++) y[k] = ar.fin_inv(v[k], sc, scs);
}
// Conversion sums a = sum_s y_s f_s (four terms, y_s < 2^60, f_s < 2^60).  A 128-bit multiply-accumulate costs ~12 instructions as the
// compiler builds it (four partial products, each followed by carry handling).  With both operands cut at 30 bits —
// y = yh 2^30 + yl, f = fh 2^30 + fl — the three partial sums  ll = sum yl fl,  mid = sum (yl fh + yh fl),  hh = sum yh fh  stay below
// 2^63, so they are plain chains of v_mad_u64_u32 with no carries (16 for four terms).  They leave as TWO words,
// a = H 2^60 + L with L = ll + (mid mod 2^32) 2^30 < 2^63 and H = hh + 4 (mid >> 32), which is all either reduction needs.
// Sources are kept in the split form (yl in the low dword, yh in the high one: the same two registers).
DEV u64 cf_split30(u64 v) { return (v & 0x3FFFFFFFull) | ((v >> 30) << 32); }
template <int NS>
struct CfConstN {
    unsigned lo[NS], hi[NS];
    DEV void set(int s, u64 f) {
        lo[s] = (unsigned)f & 0x3FFFFFFFu;
        hi[s] = (unsigned)(f >> 30);
    }
};
typedef CfConstN<HY_CF_SRC> CfConst;
struct CfSum {
    u64 L, H;
    DEV u128 wide() const { return (u128)L + ((u128)H << 60); }
};
// (NS = 5, round 5: five sources BELOW 2^48 — yh < 2^18 — keep every bound: ll < 5 2^60 < 2^63, mid < 5 (2^60 + 2^48) < 2^63, hh < 5 2^48, so
// L < 2^63 and H < 2^51; the launcher refuses five sources with a wider modulus)
template <int NS>
DEV CfSum cf_macN(const u64 (&y)[NS], const CfConstN<NS> &f) {
    u64 pll = 0, pmid = 0, phh = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const unsigned yl = (unsigned)y[s], yh = (unsigned)(y[s] >> 32);
        pll += (u64)yl * f.lo[s];
        pmid += (u64)yl * f.hi[s];
        pmid += (u64)yh * f.lo[s];
        phh += (u64)yh * f.hi[s];
    }
    CfSum r;
    r.L = pll + ((pmid & 0xFFFFFFFFull) << 30);
    r.H = phh + ((pmid >> 32) << 2);
    return r;
}
DEV CfSum cf_mac4(u64 y0, u64 y1, u64 y2, u64 y3, const CfConst &f) {
    const u64 y[HY_CF_SRC] = {y0, y1, y2, y3};
    return cf_macN<HY_CF_SRC>(y, f);
}
// row k of NS sources held as y[s][k]
template <int NS, int NR>
DEV CfSum cf_mac_row(const u64 (&y)[NS][NR], int k, const CfConstN<NS> &f) {
    u64 yy[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) yy[s] = y[s][k];
    return cf_macN<NS>(yy, f);
}
// FP64 targets (q < 2^47, so f < 2^47 and H < 2^50): the transform is linear and exact on ANY representative below ~2 q in magnitude
// (growth 0.75 q per stage, 15 stages, headroom 2^52), so the sum is folded in FP64 instead of being reduced to the canonical residue:
// H (2^60 mod q) by one exact FP64 product with its quotient, the top dword of L times 2^32 with one quotient, the low dword as it is —
// |operand| <= 1.4 q (H < 2^50: the quotient estimate is off by at most 3/8, so the first term stays below 0.875 q; the second below
// 0.5 q; 1.9 q with the dropped limb's centred residue).  The canonical results after pass 2 are the same residues.
DEV double cf_fold(const FpA &ar, const CfSum a, double c60) {
    const double t1 = ar.mulmod2(FpA::u2d(a.H), c60);
    const double h = (double)(unsigned)(a.L >> 32) * 4294967296.0;  // exact
    const double t0 = __fma_rn(-rint(h * ar.qinv), ar.q, h);
    return t1 + t0 + (double)(unsigned)a.L;
}

// split: leave the residues in the 30 + 30 bit form the conversion sums take (conversion sources; not the dropped limb)
DEV void cf_inverse_any(const NttTables &T, int m, const ModC &M, const ulonglong2 *ltw, u64 *lds, int g, int col, u64 sc, u64 scs,
                        u64 (&y)[16], bool split) {
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.itwf : T.itwp) + (size_t)m * 32768;
    if (fp) cf_inverse<FpA>(FpA(M), tw, ltw, lds, g, col, sc, scs, y);
    else if ((T.pm_mask >> m) & 1u) cf_inverse<IntP>(IntP(M), tw, ltw, lds, g, col, sc, scs, y);
    else cf_inverse<IntA>(IntA(M), tw, ltw, lds, g, col, sc, scs, y);
    if (split) {
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = cf_split30(y[k]);
    }
}
// v: operands of rows g + 8k (this arithmetic's representation of canonical residues); the raw pass-1 image leaves through d
template <class A>
DEV void cf_forward(const A ar, const ulonglong2 *__restrict__ tw, const ulonglong2 *ltw, u64 *lds, int g, int col,
                    typename A::T (&v)[16], u64 *d) {
    typedef typename A::T T;
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const int h = 8 >> st;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (!(k & h)) ar.ct(v[k], v[k + h], A::tw(tw[(1 << st) + (k >> (4 - st))]));
        if (st == 2) {  // lazy 60-bit limbs: three stages between folds (ntt_arith.h)
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
}
// target value of row k: sum_s y_s f_s (+ the centred dropped limb) as the forward butterflies' operand.  Absent sources (a short
// last digit) hold zeros, so the sum is branch-free: four lazy 128-bit multiply-accumulates.
// Integer targets (the 60-bit limb 0): canonical residue, single-word Barrett on the top bits (reduce128k: at most four products of
// residues below 2^60 with constants below q).
template <class A, bool MDR, int NR, int NS>
DEV void cf_convert(const A ar, const ModC &M, const CfConstN<NS> &f, const u64 (&y)[NS][NR], const u64 *um /* [k * 256] */,
                    unsigned neg, typename A::T (&v)[NR], bool nosrc) {
#pragma unroll
    for (int k = 0; k < NR; k++) {
        u64 r = nosrc ? 0 : reduce128k(cf_mac_row(y, k, f).wide(), M);
        if (MDR) {
            const u64 c = reduce64(um[k * 256], M);
            r = addmod(r, ((neg >> k) & 1u) ? negmod(c, M.q) : c, M.q);
        }
        v[k] = ar.from_canon(r);
    }
}
// Pseudo-Mersenne targets (q_0 and the special primes, 2^60 - c): the two-word sum is FOLDED (IntP::fold_lh, four multiply-adds) to a
// lazy representative below 2.07 2^60 instead of Barrett-reduced — with the dropped limb's residue (at most q) the forward
// butterflies start below 3.1 q and reach 15.1 q at their first fold (bound 16 q).
template <bool MDR, int NR, int NS>
DEV void cf_convert(const IntP ar, const ModC &M, const CfConstN<NS> &f, const u64 (&y)[NS][NR], const u64 *um /* [k * 256] */,
                    unsigned neg, u64 (&v)[NR], bool nosrc) {
#pragma unroll
    for (int k = 0; k < NR; k++) {
        u64 r = 0;
        if (!nosrc) {
            const CfSum a = cf_mac_row(y, k, f);
            r = ar.fold_lh(a.L, a.H);
        }
        if (MDR) {
            const u64 c = reduce64(um[k * 256], M);
            r += ((neg >> k) & 1u) ? M.q - c : c;
        }
        v[k] = r;
    }
}
template <bool MDR, int NR, int NS>
DEV void cf_convert(const FpA ar, const ModC &M, const CfConstN<NS> &f, const u64 (&y)[NS][NR], const u64 *um /* [k * 256] */,
                    unsigned neg, double c60, double (&v)[NR], bool nosrc) {
#pragma unroll
    for (int k = 0; k < NR; k++) {
        double r = nosrc ? 0.0 : cf_fold(ar, cf_mac_row(y, k, f), c60);
        if (MDR) {
            const double c = FpA::u2d(um[k * 256]);  // |centred residue| < 2^59: exact only below 2^52 — dropped limbs are scaling primes (< 2^47)
            r += ((neg >> k) & 1u) ? -c : c;
        }
        v[k] = r;
    }
}

// grid (8 column tiles, XP polynomials, ncf maps x target slices), 256 threads: col = t & 31, g = t >> 5
template <bool MDR>
__global__ __launch_bounds__(256, 2) void k_ntt15_colfuse(NttTables T, const u64 *__restrict__ src, size_t so, u64 *__restrict__ dst,
                                                           size_t dso, const ColFuse *__restrict__ cfs, int slices, int tz) {
    constexpr int N = 32768;
    extern __shared__ __attribute__((aligned(16))) u64 cf_smem[];
    u64 *const img = cf_smem;  // two exchange images of 128 x 32
    ulonglong2 *const sltw = reinterpret_cast<ulonglong2 *>(cf_smem + 2 * 4096);  // [HY_CF_SRC + 1][128]
    ulonglong2 *const tltw = sltw + (HY_CF_SRC + 1) * 128;                        // [2][128]
    const int zi = blockIdx.z / slices, zs = blockIdx.z - zi * slices;
    const ColFuse &cf = cfs[zi];
    const int t_lo = zs * tz, t_hi = min(cf.nt, t_lo + tz);
    if (t_lo >= t_hi) return;  // workgroup-uniform
    const int t = threadIdx.x, col = t & 31, g = t >> 5;
    const int xp = blockIdx.y, c0 = blockIdx.x * 32;
    const u64 *sb = src + (size_t)xp * so + c0 + col;
    // ---- every global load of the workgroup, up front: the raw pass-2' values of rows 8h + l (h = g + 8 hh), where pass 1' starts
    u64 *const umem = cf_smem + 4096 + t;  // MDR: lane t's slot of row k at umem[k * 256] (the second image)
    u64 y[HY_CF_SRC][16], um[16];
    if (MDR) {
        const u64 *sp = sb + (size_t)cf.urow * N;
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
            for (int l = 0; l < 8; l++) um[8 * hh + l] = sp[(size_t)(8 * (g + 8 * hh) + l) * 256];
    }
    // (MDR: the last two sources are requested once the dropped limb has left its registers — their latency hides behind the first
    // two sources' transforms, and the peak stays inside the 256 registers)
    auto load_source = [&](int s) {
        if (s >= cf.nk) {
#pragma unroll
            for (int k = 0; k < 16; k++) y[s][k] = 0;
        } else {
            const u64 *sp = sb + (size_t)cf.srow[s] * N;
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int l = 0; l < 8; l++) y[s][8 * hh + l] = sp[(size_t)(8 * (g + 8 * hh) + l) * 256];
        }
    };
#pragma unroll
    for (int s = 0; s < (MDR ? 2 : HY_CF_SRC); s++) load_source(s);
    if (t < 128) {
#pragma unroll
        for (int s = 0; s < HY_CF_SRC; s++)
            if (s < cf.nk) {
                const int m = cf.smod[s];
                const bool fp = (T.fp_mask >> m) & 1u;
                sltw[s * 128 + t] = ((fp ? T.itwf : T.itwp) + (size_t)m * N)[t];
            }
        if (MDR) {
            const int m = cf.umod;
            const bool fp = (T.fp_mask >> m) & 1u;
            sltw[HY_CF_SRC * 128 + t] = ((fp ? T.itwf : T.itwp) + (size_t)m * N)[t];
        }
    }
    __syncthreads();
    int buf = 0;
    // MDR exchanges through image 0 only (image 1 holds the dropped limb's residues): a barrier before an image is rewritten
#define CF_NEXT_IMAGE() (MDR ? (__syncthreads(), img) : img + (buf ^= 1) * 4096)
    // ---- pass 1' of every source: raw -> canonical coefficient-form residues, in place in y / um (the dropped limb first: its
    // registers are free again before the others are transformed)
    if (MDR) {
        cf_inverse_any(T, cf.umod, cf.uM, sltw + HY_CF_SRC * 128, CF_NEXT_IMAGE(), g, col, cf.usc, cf.usc_sh, um, false);
#pragma unroll
        for (int k = 0; k < 16; k++) umem[k * 256] = um[k];
#pragma unroll
        for (int s = 2; s < HY_CF_SRC; s++) load_source(s);
    }
    if (0 < cf.nk) cf_inverse_any(T, cf.smod[0], cf.sM[0], sltw, CF_NEXT_IMAGE(), g, col, cf.ssc[0], cf.ssc_sh[0], y[0], true);
    if (1 < cf.nk) cf_inverse_any(T, cf.smod[1], cf.sM[1], sltw + 128, CF_NEXT_IMAGE(), g, col, cf.ssc[1], cf.ssc_sh[1], y[1], true);
    if (2 < cf.nk) cf_inverse_any(T, cf.smod[2], cf.sM[2], sltw + 256, CF_NEXT_IMAGE(), g, col, cf.ssc[2], cf.ssc_sh[2], y[2], true);
    if (3 < cf.nk) cf_inverse_any(T, cf.smod[3], cf.sM[3], sltw + 384, CF_NEXT_IMAGE(), g, col, cf.ssc[3], cf.ssc_sh[3], y[3], true);
    // ---- merged ModDown + Rescale: the dropped limb of the would-be ModDown output, centred (k_moddown_rescale_conv's first half)
    unsigned neg = 0;
    if (MDR) {
        const ModC Ml = cf.lM;
        const u64 half = Ml.q >> 1;
        CfConst fl;
#pragma unroll
        for (int s = 0; s < HY_CF_SRC; s++) fl.set(s, s < cf.nk ? cf.fl[s] : 0);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64 yl = cf.nk == 0 ? umem[k * 256]  // a plain Rescale: the dropped limb itself
                                      : submod(umem[k * 256], reduce128k(cf_mac4(y[0][k], y[1][k], y[2][k], y[3][k], fl).wide(), Ml), Ml.q);  // own slot: no barrier needed
            const bool ng = yl > half;
            umem[k * 256] = ng ? Ml.q - yl : yl;
            neg |= (ng ? 1u : 0u) << k;
            }
    }
    // ---- every target of this slice: conversion, pass 1, raw image out
    // (every source load has been consumed by its transform; said explicitly, because the compiler cannot prove it for a source the
    // map does not have and would otherwise wait — at the top of EVERY target — for "the loads" with vmcnt(0), i.e. for the previous
    // target's sixteen stores)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // (the twiddle request below is made and stored by all 256 lanes — the upper half duplicates the lower — so that no path leaves a
    // load pending either)
#define CF_STORE_LTW() ltw[t & 127] = ltv
    for (int tt = t_lo; tt < t_hi; tt++) {
        // (every constant of the target straight off the map — one batch of scalar loads; the phase-B twiddles are requested here and
        // stored to LDS after the conversion, where their latency has passed)
        const int m = cf.tmod[tt];
        const ModC M = cf.tM[tt];
        const bool fp = (T.fp_mask >> m) & 1u;
        const ulonglong2 *__restrict__ tw = (fp ? T.twf : T.twp) + (size_t)m * N;
        ulonglong2 *ltw = tltw + (tt & 1) * 128;
        u64 *d = dst + (size_t)xp * dso + (size_t)cf.trow[tt] * N + c0;  // cf_forward adds the lane's column
        u64 *lds = CF_NEXT_IMAGE();
        // the sources are loop-invariant: without this the compiler hoists per-source subexpressions of the conversion out of the target
        // loop and runs out of registers (an empty statement, no instruction)
#pragma unroll
        for (int s = 0; s < HY_CF_SRC; s++)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("" : "+v"(y[s][k]));
        CfConst f;
#pragma unroll
        for (int s = 0; s < HY_CF_SRC; s++) f.set(s, s < cf.nk ? cf.f[s][tt] : 0);
        const ulonglong2 ltv = tw[t & 127];
        if (fp) {
            const FpA ar(M);
            double v[16];
            cf_convert<MDR, 16>(ar, M, f, y, umem, neg, FpA::u2d(cf.t60[tt]), v, MDR && cf.nk == 0);
            CF_STORE_LTW();
            cf_forward<FpA>(ar, tw, ltw, lds, g, col, v, d);
        } else if ((T.pm_mask >> m) & 1u) {
            const IntP ar(M);
            u64 v[16];
            cf_convert<MDR, 16>(ar, M, f, y, umem, neg, v, MDR && cf.nk == 0);
            CF_STORE_LTW();
            cf_forward<IntP>(ar, tw, ltw, lds, g, col, v, d);
        } else {
            const IntA ar(M);
            u64 v[16];
            cf_convert<IntA, MDR, 16>(ar, M, f, y, umem, neg, v, MDR && cf.nk == 0);
            CF_STORE_LTW();
            cf_forward<IntA>(ar, tw, ltw, lds, g, col, v, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------ the narrow form (round 5)
// The kernel above keeps 4 sources x 16 rows per lane (128 registers) and two 32 KiB images: 236-256 registers and 78 KiB of LDS, TWO
// workgroups per CU and nothing else beside them — no third wave (r04: 82 % / 74 % of the issue slots at two waves), and no workgroup of
// ANOTHER kernel on the CU either, which is why the per-block tails' two streams never overlapped a vector-bound conversion with a
// memory-bound transform.  Here a workgroup owns 16 columns and a lane 8 rows of each source (64 registers): the 128-point column
// transform becomes 3 + 3 + 1 stages with TWO exchanges (rows g + 16k -> 16h + 2l + e -> 8g + m; g = t >> 4, h = g >> 1, e = g & 1)
// through 16 KiB images — same butterflies on the same operands, same folds: bit-identical — at <= 168 registers and 46 KiB of LDS:
// three workgroups per CU, or two beside other kernels' workgroups.  HYDIA_COLFUSE_WIDE=1 runs the kernel above.
constexpr int CF8_COLS = 16, CF8_IMG = 128 * CF8_COLS;
constexpr int CF8_LDS_BYTES = 2 * CF8_IMG * 8 + (HY_CF_SRC_MAX + 1 + 2) * 128 * 16;

// y: in = raw pass-2' values of rows 8g + m at index m; out = canonical coefficient-form residues (times sc) of rows g + 16k at index k
template <class A>
DEV void cf_inverse8(const A ar, const ulonglong2 *__restrict__ tw, const ulonglong2 *ltw, u64 *lds, int g, int col, u64 sc, u64 scs,
                     u64 (&y)[8]) {
    typedef typename A::T T;
    T w[8];
    // phase C': stage 6 (row bit 0)
#pragma unroll
    for (int m = 0; m < 8; m++) w[m] = A::from_bits(y[m]);
#pragma unroll
    for (int m = 0; m < 8; m += 2) ar.gs(w[m], w[m + 1], A::tw(ltw[64 + 4 * g + (m >> 1)]));
#pragma unroll
    for (int m = 0; m < 8; m++) {
        ar.recentre(w[m]);
        lds[(8 * g + m) * CF8_COLS + col] = A::to_bits(w[m]);
    }
    __syncthreads();
    // phase B': rows 16h + 2l + e, stages 5, 4, 3 (row bits 1, 2, 3); written back in place (a lane's own slots)
    const int h = g >> 1, e = g & 1;
#pragma unroll
    for (int l = 0; l < 8; l++) w[l] = A::from_bits(lds[(16 * h + 2 * l + e) * CF8_COLS + col]);
#pragma unroll
    for (int l = 0; l < 8; l += 2) ar.gs(w[l], w[l + 1], A::tw(ltw[32 + 4 * h + (l >> 1)]));
#pragma unroll
    for (int l = 0; l < 8; l++)
        if (!(l & 2)) ar.gs(w[l], w[l + 2], A::tw(ltw[16 + 2 * h + (l >> 2)]));
    {
        const typename A::TW W = A::tw(ltw[8 + h]);
#pragma unroll
        for (int l = 0; l < 4; l++) ar.gs(w[l], w[l + 4], W);
    }
#pragma unroll
    for (int l = 0; l < 8; l++) {
        ar.recentre(w[l]);
        lds[(16 * h + 2 * l + e) * CF8_COLS + col] = A::to_bits(w[l]);
    }
    __syncthreads();
    // phase A': rows g + 16k, stages 2, 1, 0 (row bits 4, 5, 6): workgroup-uniform twiddles
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = A::from_bits(lds[(g + 16 * k) * CF8_COLS + col]);
#pragma unroll
    for (int k = 0; k < 8; k += 2) ar.gs(w[k], w[k + 1], A::tw(tw[4 + (k >> 1)]));
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (!(k & 2)) ar.gs(w[k], w[k + 2], A::tw(tw[2 + (k >> 2)]));
    {
        const typename A::TW W = A::tw(tw[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) ar.gs(w[k], w[k + 4], W);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = ar.fin_inv(w[k], sc, scs);
}
DEV void cf_inverse8_any(const NttTables &T, int m, const ModC &M, const ulonglong2 *ltw, u64 *lds, int g, int col, u64 sc, u64 scs,
                         u64 (&y)[8], bool split) {
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.itwf : T.itwp) + (size_t)m * 32768;
    if (fp) cf_inverse8<FpA>(FpA(M), tw, ltw, lds, g, col, sc, scs, y);
    else if ((T.pm_mask >> m) & 1u) cf_inverse8<IntP>(IntP(M), tw, ltw, lds, g, col, sc, scs, y);
    else cf_inverse8<IntA>(IntA(M), tw, ltw, lds, g, col, sc, scs, y);
    if (split) {
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = cf_split30(y[k]);
    }
}
// v: operands of rows g + 16k; the raw pass-1 image (rows 8g + m of the lane's column) leaves through d
template <class A>
DEV void cf_forward8(const A ar, const ulonglong2 *__restrict__ tw, const ulonglong2 *ltw, u64 *lds, int g, int col, typename A::T (&v)[8], u64 *d) {
    typedef typename A::T T;
    // phase A: stages 0, 1, 2
    {
        const typename A::TW W = A::tw(tw[1]);
#pragma unroll
        for (int k = 0; k < 4; k++) ar.ct(v[k], v[k + 4], W);
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (!(k & 2)) ar.ct(v[k], v[k + 2], A::tw(tw[2 + (k >> 2)]));
#pragma unroll
    for (int k = 0; k < 8; k += 2) ar.ct(v[k], v[k + 1], A::tw(tw[4 + (k >> 1)]));
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ar.fwd_fold(v[k]);  // lazy 60-bit limbs: three stages between folds (ntt_arith.h)
        lds[(g + 16 * k) * CF8_COLS + col] = A::to_bits(v[k]);
    }
    __syncthreads();
    // phase B: rows 16h + 2l + e, stages 3, 4, 5
    const int h = g >> 1, e = g & 1;
    T w[8];
#pragma unroll
    for (int l = 0; l < 8; l++) w[l] = A::from_bits(lds[(16 * h + 2 * l + e) * CF8_COLS + col]);
    {
        const typename A::TW W = A::tw(ltw[8 + h]);
#pragma unroll
        for (int l = 0; l < 4; l++) ar.ct(w[l], w[l + 4], W);
    }
#pragma unroll
    for (int l = 0; l < 8; l++)
        if (!(l & 2)) ar.ct(w[l], w[l + 2], A::tw(ltw[16 + 2 * h + (l >> 2)]));
#pragma unroll
    for (int l = 0; l < 8; l += 2) ar.ct(w[l], w[l + 1], A::tw(ltw[32 + 4 * h + (l >> 1)]));
#pragma unroll
    for (int l = 0; l < 8; l++) {
        ar.fwd_fold(w[l]);
        lds[(16 * h + 2 * l + e) * CF8_COLS + col] = A::to_bits(w[l]);
    }
    __syncthreads();
    // phase C: rows 8g + m, stage 6
#pragma unroll
    for (int m = 0; m < 8; m++) w[m] = A::from_bits(lds[(8 * g + m) * CF8_COLS + col]);
#pragma unroll
    for (int m = 0; m < 8; m += 2) ar.ct(w[m], w[m + 1], A::tw(ltw[64 + 4 * g + (m >> 1)]));
#pragma unroll
    for (int m = 0; m < 8; m++) d[(size_t)(8 * g + m) * 256 + col] = A::to_bits(w[m]);  // raw: pass 2 finishes
}

// grid (16 column tiles, XP polynomials, ncf maps x target slices), 256 threads: col = t & 15, g = t >> 4
// NS: conversion sources held per lane — 4, or 5 for a ModDown over five special primes below 2^48 (two workgroups per CU then)
template <bool MDR, int NS = HY_CF_SRC>
__global__ __launch_bounds__(256, NS > HY_CF_SRC ? 2 : 3) void k_ntt15_colfuse8(NttTables T, const u64 *__restrict__ src, size_t so, u64 *__restrict__ dst,
                                                            size_t dso, const ColFuse *__restrict__ cfs, int slices, int tz) {
    constexpr int N = 32768;
    extern __shared__ __attribute__((aligned(16))) u64 cf_smem[];
    u64 *const img = cf_smem;  // two exchange images of 128 x 16
    ulonglong2 *const sltw = reinterpret_cast<ulonglong2 *>(cf_smem + 2 * CF8_IMG);  // [HY_CF_SRC_MAX + 1][128]
    ulonglong2 *const tltw = sltw + (HY_CF_SRC_MAX + 1) * 128;                          // [2][128]
    const int zi = blockIdx.z / slices, zs = blockIdx.z - zi * slices;
    const ColFuse &cf = cfs[zi];
    const int t_lo = zs * tz, t_hi = min(cf.nt, t_lo + tz);
    if (t_lo >= t_hi) return;  // workgroup-uniform
    const int t = threadIdx.x, col = t & (CF8_COLS - 1), g = t >> 4;
    const int xp = blockIdx.y, c0 = blockIdx.x * CF8_COLS;
    const u64 *sb = src + (size_t)xp * so + c0 + col;
    // ---- every global load of the workgroup, up front: the raw pass-2' values of rows 8g + m, where pass 1' starts
    u64 *const umem = cf_smem + CF8_IMG + t;  // MDR: lane t's slot of row k at umem[k * 256] (the second image)
    u64 y[NS][8], um[8];
    if (MDR) {
        const u64 *sp = sb + (size_t)cf.urow * N;
#pragma unroll
        for (int m = 0; m < 8; m++) um[m] = sp[(size_t)(8 * g + m) * 256];
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s >= cf.nk) {
#pragma unroll
            for (int k = 0; k < 8; k++) y[s][k] = 0;
        } else {
            const u64 *sp = sb + (size_t)cf.srow[s] * N;
#pragma unroll
            for (int m = 0; m < 8; m++) y[s][m] = sp[(size_t)(8 * g + m) * 256];
        }
    }
    if (t < 128) {
#pragma unroll
        for (int s = 0; s < NS; s++)
            if (s < cf.nk) {
                const int m = cf.smod[s];
                const bool fp = (T.fp_mask >> m) & 1u;
                sltw[s * 128 + t] = ((fp ? T.itwf : T.itwp) + (size_t)m * N)[t];
            }
        if (MDR) {
            const int m = cf.umod;
            const bool fp = (T.fp_mask >> m) & 1u;
            sltw[HY_CF_SRC_MAX * 128 + t] = ((fp ? T.itwf : T.itwp) + (size_t)m * N)[t];
        }
    }
    __syncthreads();
    int buf = 0;
    // MDR exchanges through image 0 only (image 1 holds the dropped limb's residues): a barrier before an image is rewritten
#define CF8_NEXT_IMAGE() (MDR ? (__syncthreads(), img) : img + (buf ^= 1) * CF8_IMG)
    if (MDR) {
        cf_inverse8_any(T, cf.umod, cf.uM, sltw + HY_CF_SRC_MAX * 128, CF8_NEXT_IMAGE(), g, col, cf.usc, cf.usc_sh, um, false);
#pragma unroll
        for (int k = 0; k < 8; k++) umem[k * 256] = um[k];
    }
    if (0 < cf.nk) cf_inverse8_any(T, cf.smod[0], cf.sM[0], sltw, CF8_NEXT_IMAGE(), g, col, cf.ssc[0], cf.ssc_sh[0], y[0], true);
    if (1 < cf.nk) cf_inverse8_any(T, cf.smod[1], cf.sM[1], sltw + 128, CF8_NEXT_IMAGE(), g, col, cf.ssc[1], cf.ssc_sh[1], y[1], true);
    if (2 < cf.nk) cf_inverse8_any(T, cf.smod[2], cf.sM[2], sltw + 256, CF8_NEXT_IMAGE(), g, col, cf.ssc[2], cf.ssc_sh[2], y[2], true);
    if (3 < cf.nk) cf_inverse8_any(T, cf.smod[3], cf.sM[3], sltw + 384, CF8_NEXT_IMAGE(), g, col, cf.ssc[3], cf.ssc_sh[3], y[3], true);
    if (NS > 4 && 4 < cf.nk) cf_inverse8_any(T, cf.smod[NS - 1], cf.sM[NS - 1], sltw + 512, CF8_NEXT_IMAGE(), g, col, cf.ssc[NS - 1], cf.ssc_sh[NS - 1], y[NS - 1], true);
    // ---- merged ModDown + Rescale: the dropped limb of the would-be ModDown output, centred (k_moddown_rescale_conv's first half)
    unsigned neg = 0;
    if (MDR) {
        const ModC Ml = cf.lM;
        const u64 half = Ml.q >> 1;
        CfConstN<NS> fl;
#pragma unroll
        for (int s = 0; s < NS; s++) fl.set(s, s < cf.nk ? cf.fl[s] : 0);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 yl = cf.nk == 0 ? umem[k * 256]  // a plain Rescale: the dropped limb itself
                                      : submod(umem[k * 256], reduce128k(cf_mac_row(y, k, fl).wide(), Ml), Ml.q);  // own slot: no barrier needed
            const bool ng = yl > half;
            umem[k * 256] = ng ? Ml.q - yl : yl;
            neg |= (ng ? 1u : 0u) << k;
        }
    }
    // ---- every target of this slice: conversion, pass 1, raw image out (see the wide kernel for the vmcnt / twiddle-store remarks)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (int tt = t_lo; tt < t_hi; tt++) {
        const int m = cf.tmod[tt];
        const ModC M = cf.tM[tt];
        const bool fp = (T.fp_mask >> m) & 1u;
        const ulonglong2 *__restrict__ tw = (fp ? T.twf : T.twp) + (size_t)m * N;
        ulonglong2 *ltw = tltw + (tt & 1) * 128;
        u64 *d = dst + (size_t)xp * dso + (size_t)cf.trow[tt] * N + c0;  // cf_forward8 adds the lane's column
        u64 *lds = CF8_NEXT_IMAGE();
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("" : "+v"(y[s][k]));
        CfConstN<NS> f;
#pragma unroll
        for (int s = 0; s < NS; s++) f.set(s, s < cf.nk ? cf.f[s][tt] : 0);
        const ulonglong2 ltv = tw[t & 127];
        if (fp) {
            const FpA ar(M);
            double v[8];
            cf_convert<MDR, 8>(ar, M, f, y, umem, neg, FpA::u2d(cf.t60[tt]), v, MDR && cf.nk == 0);
            ltw[t & 127] = ltv;
            cf_forward8<FpA>(ar, tw, ltw, lds, g, col, v, d);
        } else if ((T.pm_mask >> m) & 1u) {
            const IntP ar(M);
            u64 v[8];
            cf_convert<MDR, 8>(ar, M, f, y, umem, neg, v, MDR && cf.nk == 0);
            ltw[t & 127] = ltv;
            cf_forward8<IntP>(ar, tw, ltw, lds, g, col, v, d);
        } else {
            const IntA ar(M);
            u64 v[8];
            cf_convert<IntA, MDR, 8>(ar, M, f, y, umem, neg, v, MDR && cf.nk == 0);
            ltw[t & 127] = ltv;
            cf_forward8<IntA>(ar, tw, ltw, lds, g, col, v, d);
        }
    }
}

// The small-launch form (a one-block query's tail: too few polynomials to fill the chip): the sources' inverse transform has run as
// its own launch, ONE target per workgroup, operands loaded where they are used — nothing is kept for a next target, so the kernel
// needs a third of the registers (four workgroups per CU, no spills) and its serial chain is one conversion + one pass 1.
// grid (8 column tiles, XP polynomials, ncf maps x targets)
template <bool MDR>
__global__ __launch_bounds__(256, MDR ? 2 : 4) void k_ntt15_conv_p1(NttTables T, const u64 *__restrict__ src, size_t so, u64 *__restrict__ dst,
                                                           size_t dso, const ColFuse *__restrict__ cfs, int nt_max) {
    constexpr int N = 32768;
    __shared__ u64 lds[128 * 32];
    __shared__ ulonglong2 ltw[128];
    const int zi = blockIdx.z / nt_max, tt = blockIdx.z - zi * nt_max;
    const ColFuse &cf = cfs[zi];
    if (tt >= cf.nt) return;  // workgroup-uniform
    const int t = threadIdx.x, col = t & 31, g = t >> 5;
    const int xp = blockIdx.y, c0 = blockIdx.x * 32;
    const u64 *sb = src + (size_t)xp * so + c0 + col;
    const int m = cf.tmod[tt];
    const ModC M = cf.tM[tt];  // (off the map: no dependent load between the target's id and its constants)
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.twf : T.twp) + (size_t)m * N;
    if (t < 128) ltw[t] = tw[t];
    CfConst f, fl;
    const u64 *sp[HY_CF_SRC];
#pragma unroll
    for (int s = 0; s < HY_CF_SRC; s++) {
        const bool on = s < cf.nk;
        f.set(s, on ? cf.f[s][tt] : 0);
        fl.set(s, (MDR && on) ? cf.fl[s] : 0);
        sp[s] = sb + (size_t)cf.srow[on ? s : 0] * N;  // absent sources re-read source 0 against a zero constant
    }
    const u64 *su = MDR ? sb + (size_t)cf.urow * N : sb;
    const ModC Ml = MDR ? cf.lM : M;
    u64 *d = dst + (size_t)xp * dso + (size_t)cf.trow[tt] * N + c0;
    // one row at a time: the (up to five) operands of row g + 8k are loaded where they are used; a holds sum_s y_s f_s, the dropped
    // limb's centred residue (MDR) is formed from the same operands
    auto convert_row = [&](int k, CfSum &a, u64 &mag, bool &ng) {
        const size_t off = (size_t)(g + 8 * k) * 256;
        const u64 y0 = cf_split30(sp[0][off]), y1 = cf_split30(sp[1][off]), y2 = cf_split30(sp[2][off]), y3 = cf_split30(sp[3][off]);
        a = cf_mac4(y0, y1, y2, y3, f);
        mag = 0;
        ng = false;
        if (MDR) {
            const u64 yl = submod(su[off], reduce128k(cf_mac4(y0, y1, y2, y3, fl).wide(), Ml), Ml.q);
            ng = yl > (Ml.q >> 1);
            mag = ng ? Ml.q - yl : yl;
        }
    };
    if (fp) {
        const FpA ar(M);
        const double c60 = FpA::u2d(cf.t60[tt]);
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            CfSum a;
            u64 mag;
            bool ng;
            convert_row(k, a, mag, ng);
            double r = cf_fold(ar, a, c60);
            if (MDR) r += ng ? -FpA::u2d(mag) : FpA::u2d(mag);
            v[k] = r;
        }
        cf_forward<FpA>(ar, tw, ltw, lds, g, col, v, d);
    } else {
        const bool pm = (T.pm_mask >> m) & 1u;
        u64 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            CfSum a;
            u64 mag;
            bool ng;
            convert_row(k, a, mag, ng);
            u64 r;
            if (pm) {  // folded, lazy (cf_convert's pseudo-Mersenne form)
                r = IntP(M).fold_lh(a.L, a.H);
                if (MDR) {
                    const u64 c = reduce64(mag, M);
                    r += ng ? M.q - c : c;
                }
            } else {
                r = reduce128k(a.wide(), M);
                if (MDR) {
                    const u64 c = reduce64(mag, M);
                    r = addmod(r, ng ? negmod(c, M.q) : c, M.q);
                }
            }
            v[k] = r;
        }
        if (pm) cf_forward<IntP>(IntP(M), tw, ltw, lds, g, col, v, d);
        else cf_forward<IntA>(IntA(M), tw, ltw, lds, g, col, v, d);
    }
}

// The small-launch form on 16-column tiles (round 5): a lane converts 8 rows instead of 16 and the column transform runs as cf_forward8 —
// twice the workgroups, each with half the serial chain (the small launches are bound by one workgroup's chain, not by throughput).
// HYDIA_COLFUSE_WIDE keeps the 32-column form above.  grid (16 column tiles, XP polynomials, ncf maps x targets)
template <bool MDR>
__global__ __launch_bounds__(256, 4) void k_ntt15_conv_p1_8(NttTables T, const u64 *__restrict__ src, size_t so, u64 *__restrict__ dst,
                                                           size_t dso, const ColFuse *__restrict__ cfs, int nt_max) {
    constexpr int N = 32768;
    __shared__ u64 lds[CF8_IMG];
    __shared__ ulonglong2 ltw[128];
    const int zi = blockIdx.z / nt_max, tt = blockIdx.z - zi * nt_max;
    const ColFuse &cf = cfs[zi];
    if (tt >= cf.nt) return;  // workgroup-uniform
    const int t = threadIdx.x, col = t & (CF8_COLS - 1), g = t >> 4;
    const int xp = blockIdx.y, c0 = blockIdx.x * CF8_COLS;
    const u64 *sb = src + (size_t)xp * so + c0 + col;
    const int m = cf.tmod[tt];
    const ModC M = cf.tM[tt];  // (off the map: no dependent load between the target's id and its constants)
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.twf : T.twp) + (size_t)m * N;
    if (t < 128) ltw[t] = tw[t];
    CfConst f, fl;
    const u64 *sp[HY_CF_SRC];
#pragma unroll
    for (int s = 0; s < HY_CF_SRC; s++) {
        const bool on = s < cf.nk;
        f.set(s, on ? cf.f[s][tt] : 0);
        fl.set(s, (MDR && on) ? cf.fl[s] : 0);
        sp[s] = sb + (size_t)cf.srow[on ? s : 0] * N;  // absent sources re-read source 0 against a zero constant
    }
    const u64 *su = MDR ? sb + (size_t)cf.urow * N : sb;
    const ModC Ml = MDR ? cf.lM : M;
    u64 *d = dst + (size_t)xp * dso + (size_t)cf.trow[tt] * N + c0;
    // one row at a time: the (up to five) operands of row g + 16k are loaded where they are used; a holds sum_s y_s f_s, the dropped
    // limb's centred residue (MDR) is formed from the same operands
    auto convert_row = [&](int k, CfSum &a, u64 &mag, bool &ng) {
        const size_t off = (size_t)(g + 16 * k) * 256;
        const u64 y0 = cf_split30(sp[0][off]), y1 = cf_split30(sp[1][off]), y2 = cf_split30(sp[2][off]), y3 = cf_split30(sp[3][off]);
        a = cf_mac4(y0, y1, y2, y3, f);
        mag = 0;
        ng = false;
        if (MDR) {
            const u64 yl = submod(su[off], reduce128k(cf_mac4(y0, y1, y2, y3, fl).wide(), Ml), Ml.q);
            ng = yl > (Ml.q >> 1);
            mag = ng ? Ml.q - yl : yl;
        }
    };
    if (fp) {
        const FpA ar(M);
        const double c60 = FpA::u2d(cf.t60[tt]);
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            CfSum a;
            u64 mag;
            bool ng;
            convert_row(k, a, mag, ng);
            double r = cf_fold(ar, a, c60);
            if (MDR) r += ng ? -FpA::u2d(mag) : FpA::u2d(mag);
            v[k] = r;
        }
        cf_forward8<FpA>(ar, tw, ltw, lds, g, col, v, d);
    } else {
        const bool pm = (T.pm_mask >> m) & 1u;
        u64 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            CfSum a;
            u64 mag;
            bool ng;
            convert_row(k, a, mag, ng);
            u64 r;
            if (pm) {  // folded, lazy (cf_convert's pseudo-Mersenne form)
                r = IntP(M).fold_lh(a.L, a.H);
                if (MDR) {
                    const u64 c = reduce64(mag, M);
                    r += ng ? M.q - c : c;
                }
            } else {
                r = reduce128k(a.wide(), M);
                if (MDR) {
                    const u64 c = reduce64(mag, M);
                    r = addmod(r, ng ? negmod(c, M.q) : c, M.q);
                }
            }
            v[k] = r;
        }
        if (pm) cf_forward8<IntP>(IntP(M), tw, ltw, lds, g, col, v, d);
        else cf_forward8<IntA>(IntA(M), tw, ltw, lds, g, col, v, d);
    }
}

// inverse pass 1' ALONE on 16-column tiles (round 5; the small launches' own inverse transform: twice the workgroups of k_ntt15_p1<true, 0>,
// half the serial chain each), in place.  grid (16 column tiles, X * nsl limb-polynomials)
__global__ __launch_bounds__(256) void k_ntt15_p1inv8(NttTables T, u64 *__restrict__ dst, size_t dso, LimbSel sel, int slot0, int nsl, ScaleSel scale) {
    constexpr int N = 32768;
    __shared__ u64 lds[CF8_IMG];
    __shared__ ulonglong2 ltw[128];
    const int y = blockIdx.y, x = y / nsl, slot = slot0 + (y - x * nsl), m = sel.mod[slot];
    const ModC M = T.mod[m];
    const bool fp = (T.fp_mask >> m) & 1u;
    const ulonglong2 *__restrict__ tw = (fp ? T.itwf : T.itwp) + (size_t)m * N;
    const int t = threadIdx.x, col = t & (CF8_COLS - 1), g = t >> 4;
    u64 *d = dst + (size_t)x * dso + (size_t)slot * N + blockIdx.x * CF8_COLS + col;
    u64 yv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) yv[k] = d[(size_t)(8 * g + k) * 256];  // raw from pass 2'
    if (t < 128) ltw[t] = tw[t];
    __syncthreads();
    if (fp) cf_inverse8<FpA>(FpA(M), tw, ltw, lds, g, col, scale.s[slot], scale.s_sh[slot], yv);
    else if ((T.pm_mask >> m) & 1u) cf_inverse8<IntP>(IntP(M), tw, ltw, lds, g, col, scale.s[slot], scale.s_sh[slot], yv);
    else cf_inverse8<IntA>(IntA(M), tw, ltw, lds, g, col, scale.s[slot], scale.s_sh[slot], yv);
#pragma unroll
    for (int k = 0; k < 8; k++) d[(size_t)(g + 16 * k) * 256] = yv[k];
}

}  // namespace

namespace hk {

// launches below 256 32-column tiles take the small-launch form (pass 1' as its own launch + one target per workgroup; the crossover was
// re-measured with the narrow kernel in round 5: nothing between 0 and 256 tiles leaves the run-to-run spread)
bool ntt15_colfuse_small(int XP, int ncf) { return 8 * XP * ncf < 256; }
// inverse pass 1' of a SMALL launch (fewer than 1024 of the 32-column workgroups) on 16-column tiles; false: the caller launches k_ntt15_p1<true, 0>
bool ntt15_inverse_p1_narrow(hipStream_t st, const NttTables &T, u64 *dst, size_t dso, int X, const LimbSel &sel, int slot0, int nsl, const ScaleSel &scale) {
    if (T.cf_wide || 8 * X * nsl >= 1024) return false;
    ledger_add("k_ntt15_p1inv8", 2.0 * X * nsl * 262144.0);
    hipLaunchKernelGGL(k_ntt15_p1inv8, dim3(16, X * nsl), dim3(256), 0, st, T, dst, dso, sel, slot0, nsl, scale);
    return true;
}

void ntt15_colfuse(hipStream_t st, const NttTables &T, const u64 *src, size_t so, u64 *dst, size_t dso, int XP, const ColFuse *d_cf,
                   const ColFuse *h_cf, int ncf, bool pre) {
    static bool attr_done[64][2] = {};  // > 64 KiB of dynamic LDS has to be granted per kernel and per device
    const bool mdr = h_cf[0].mdr != 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev][mdr]) {
        if (mdr) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt15_colfuse<true>), hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS_BYTES);
        else (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt15_colfuse<false>), hipFuncAttributeMaxDynamicSharedMemorySize, CF_LDS_BYTES);
        attr_done[dev][mdr] = true;
    }
    int nt_max = 0;
    double rows = 0;  // sources in, targets out, each once
    for (int i = 0; i < ncf; i++) {
        nt_max = std::max(nt_max, h_cf[i].nt);
        rows += h_cf[i].nk + (h_cf[i].mdr ? 1 : 0) + h_cf[i].nt;
    }
    if (pre && !T.cf_wide) {  // small launch: one target per workgroup, 16-column tiles
        ledger_add(mdr ? "k_ntt15_conv_p1_8<true>" : "k_ntt15_conv_p1_8<false>", rows * XP * 262144.0);
        if (mdr) hipLaunchKernelGGL((k_ntt15_conv_p1_8<true>), dim3(16, XP, ncf * nt_max), dim3(256), 0, st, T, src, so, dst, dso, d_cf, nt_max);
        else hipLaunchKernelGGL((k_ntt15_conv_p1_8<false>), dim3(16, XP, ncf * nt_max), dim3(256), 0, st, T, src, so, dst, dso, d_cf, nt_max);
        return;
    }
    if (pre) {  // ... on round 4's 32-column tiles
        ledger_add(mdr ? "k_ntt15_conv_p1<true>" : "k_ntt15_conv_p1<false>", rows * XP * 262144.0);
        if (mdr) hipLaunchKernelGGL((k_ntt15_conv_p1<true>), dim3(8, XP, ncf * nt_max), dim3(256), 0, st, T, src, so, dst, dso, d_cf, nt_max);
        else hipLaunchKernelGGL((k_ntt15_conv_p1<false>), dim3(8, XP, ncf * nt_max), dim3(256), 0, st, T, src, so, dst, dso, d_cf, nt_max);
        return;
    }
    int nk_max = 0;
    for (int i = 0; i < ncf; i++) nk_max = std::max(nk_max, h_cf[i].nk);
    if (nk_max > HY_CF_SRC) {  // five sources (special primes below 2^48; cf_plan_* vouches): the narrow kernel's five-source instantiation only
        if (pre || nk_max > HY_CF_SRC_MAX) throw std::logic_error("hydia: column-fused conversion with more than four sources in the small-launch form");
        ledger_add(mdr ? "k_ntt15_colfuse8<true, 5>" : "k_ntt15_colfuse8<false, 5>", rows * XP * 262144.0);
        if (mdr) hipLaunchKernelGGL((k_ntt15_colfuse8<true, 5>), dim3(16, XP, ncf), dim3(256), CF8_LDS_BYTES, st, T, src, so, dst, dso, d_cf, 1, nt_max);
        else hipLaunchKernelGGL((k_ntt15_colfuse8<false, 5>), dim3(16, XP, ncf), dim3(256), CF8_LDS_BYTES, st, T, src, so, dst, dso, d_cf, 1, nt_max);
        return;
    }
    if (!T.cf_wide) {  // the narrow form: 16 column tiles of 16 columns
        ledger_add(mdr ? "k_ntt15_colfuse8<true, 4>" : "k_ntt15_colfuse8<false, 4>", rows * XP * 262144.0);  // (as rocprofv3 prints the instantiation)
        const int base8 = 16 * XP * ncf;
        int sl = 1;
        if (base8 < 768) sl = std::min(nt_max, (768 + base8 - 1) / base8);
        const int tz8 = (nt_max + sl - 1) / sl;
        sl = (nt_max + tz8 - 1) / tz8;
        if (mdr) hipLaunchKernelGGL((k_ntt15_colfuse8<true>), dim3(16, XP, ncf * sl), dim3(256), CF8_LDS_BYTES, st, T, src, so, dst, dso, d_cf, sl, tz8);
        else hipLaunchKernelGGL((k_ntt15_colfuse8<false>), dim3(16, XP, ncf * sl), dim3(256), CF8_LDS_BYTES, st, T, src, so, dst, dso, d_cf, sl, tz8);
        return;
    }
    ledger_add(mdr ? "k_ntt15_colfuse<true>" : "k_ntt15_colfuse<false>", rows * XP * 262144.0);
    // launches that cannot fill the chip (a one-block query's tail) slice their targets over grid.z: every slice re-reads the sources
    // (from L2) and serves tz targets, so the serial chain per workgroup shrinks with the launch
    const int base = 8 * XP * ncf;
    int slices = 1;
    if (base < 512) slices = std::min(nt_max, (512 + base - 1) / base);
    const int tz = (nt_max + slices - 1) / slices;
    slices = (nt_max + tz - 1) / tz;
    if (mdr)
        hipLaunchKernelGGL((k_ntt15_colfuse<true>), dim3(8, XP, ncf * slices), dim3(256), CF_LDS_BYTES, st, T, src, so, dst, dso, d_cf, slices, tz);
    else
        hipLaunchKernelGGL((k_ntt15_colfuse<false>), dim3(8, XP, ncf * slices), dim3(256), CF_LDS_BYTES, st, T, src, so, dst, dso, d_cf, slices, tz);
}

}  // namespace hk

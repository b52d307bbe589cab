// image_matching_amd/csrc/ntt_arith.h — the three exact butterfly arithmetics of the N = 2^15 transforms (ntt15.hip, colfuse.hip).
//
// Chosen per limb (workgroup-uniform) and all EXACT, so results are bit-identical:
//   IntA  64-bit integers, Shoup multiplication, Harvey lazy reduction ([0,4q) forward, [0,2q) inverse): any prime up to 61 bits.
//   IntP  q = 2^60 - c, c < 2^24 (q_0 and the special primes): values live lazily in [0, 16q), occasional three-instruction folds,
//         Shoup quotients from the high halves only.
//   FpA   the 45/46-bit scaling primes fit a double's 53-bit mantissa: residues are exact-integer doubles and a*w mod q is
//         h = a*w; l = fma(a,w,-h); c = rint(a*(w/q)); r = fma(-c,q,h) + l — six FP64 instructions, no compares, no carries.
// Between two passes FpA limbs travel as raw doubles in the (uint64) buffer; every limb leaves a transform as canonical uint64.
#pragma once
#include "kernels.h"

namespace {

struct IntA {
    typedef u64 T;
    typedef ulonglong2 TW;  // (w, floor(w 2^64 / q))
    u64 q, q2;
    DEV IntA(const ModC &M) : q(M.q), q2(2 * M.q) {}
    DEV static TW tw(const ulonglong2 b) { return b; }
    DEV T from_canon(u64 x) const { return x; }
    DEV static T from_bits(u64 x) { return x; }
    DEV static u64 to_bits(T x) { return x; }
    DEV void ct(T &a, T &b, const TW W) const {  // [0,4q) -> [0,4q)
        const u64 u = a >= q2 ? a - q2 : a;
        const u64 hi = __umul64hi(b, W.y);
        const u64 t = b * W.x - hi * q;
        a = u + t;
        b = u - t + q2;
    }
    DEV void gs(T &a, T &b, const TW W) const {  // [0,2q) -> [0,2q)
        u64 s = a + b;
        s = s >= q2 ? s - q2 : s;
        const u64 d = a - b + q2;
        const u64 hi = __umul64hi(d, W.y);
        b = d * W.x - hi * q;
        a = s;
    }
    DEV void recentre(T &) const {}
    DEV void recentre_wide(T &) const {}
    DEV void mid(T &) const {}
    DEV void fwd_fold(T &) const {}
    DEV T from_raw(u64 x) const { return x; }
    DEV u64 fin_fwd(T x) const {
        x = x >= q2 ? x - q2 : x;
        return x >= q ? x - q : x;
    }
    DEV u64 fin_inv(T x, u64 sc, u64 scs) const { return mulmod_shoup(x, sc, scs, q); }
};

// q = 2^60 - c with c < 2^24 (every 60-bit prime OpenFHE picks for this parameter set): values live lazily in [0, 16q) = [0, 2^64 - 16c)
// and a conditional subtraction (five instructions per butterfly) becomes an occasional three-instruction fold x -> (x mod 2^60) +
// (x >> 60) c, which lands any 64-bit x in [0, 2^60 + 15c] — "1+ q" below.
// Round 4: the Shoup quotient is taken from the HIGH halves only, hi' = xh wh' + hi32(xl wh') + hi32(xh wl') in [hi - 2, hi]
// (what is dropped — the low words of the two cross products and the whole of xl wl' — is below 3 2^64): two v_mul_hi_u32, one
// v_mad_u64_u32 and one 64-bit add instead of four multiplies, three register moves and an add (the exact __umul64hi), 17 instead of
// 21 instructions per butterfly.  The product is then in [0, 4q) instead of [0, 2q).
// Forward (Cooley-Tukey): a' = a + t, b' = a - t + 4q grow the bound by 4q per stage, so THREE stages run between folds
// (1+ -> 5 -> 9 -> 13 q; a fourth would pass 2^64): pass 1 folds after its third and sixth stage (fwd_fold) and leaves 5+ q, pass 2
// folds on reading (from_raw), after phase A (fwd_fold), after phase B (mid) and in fin_fwd — three folds (nine instructions per
// coefficient) more than the exact quotient needed, against 30 saved.  Inverse (Gentleman-Sande): sums double, d = a - b + 8q needs
// b < 8q: from 1+ q the bounds run 4, 8, 16 q over three stages (products below 4q never lead), which is where the folds already
// sat (the hooks the FP64 path re-centres at, plus one inside pass 1's four-stage group) — nothing added there.
// Same residues as IntA after the final reduction.
struct IntP {
    typedef u64 T;
    typedef ulonglong2 TW;
    u64 q, q4, q8;
    unsigned c;
    DEV IntP(const ModC &M) : q(M.q), q4(4 * M.q), q8(8 * M.q), c((unsigned)((1ull << 60) - M.q)) {}
    DEV static TW tw(const ulonglong2 b) { return b; }
    DEV T from_canon(u64 x) const { return x; }
    DEV static T from_bits(u64 x) { return x; }
    DEV static u64 to_bits(T x) { return x; }
    DEV u64 fold(u64 x) const { return (x & ((1ull << 60) - 1)) + (u64)(unsigned)(x >> 60) * c; }
    // Wide sums without a Barrett reduction (round 4): with 2^60 = c and 2^64 = 16c (mod q) a two-word sum folds to a lazy
    // representative in four carry-free multiply-adds — the transforms take lazy operands anyway (forward: up to 12q before a
    // stage; inverse: below 4q where two stages run to the next fold).
    // L + H 2^60, H < 2^63 (the conversion sums of colfuse.hip: L < 2^63, H < 2^62 + 2^33): result below 2.07 2^60.
    DEV u64 fold_lh(u64 L, u64 H) const {
        const u64 m1 = (u64)(unsigned)(H >> 32) * c;  // < 2^55; m1 2^32 = (m1 >> 28) 2^60 + (m1 mod 2^28) 2^32
        u64 acc = (L & ((1ull << 60) - 1)) + (u64)(unsigned)(L >> 60) * c;
        acc += (u64)(unsigned)H * c;            // < 2^56
        acc += (u64)(unsigned)(m1 >> 28) * c;   // < 2^51
        return acc + ((u64)((unsigned)m1 & 0x0FFFFFFFu) << 32);  // + (< 2^60)
    }
    // ANY 128-bit z (the key-switching inner products: up to four products of a lazy value below 2^64 with a key residue): result
    // below 3.07 2^60.
    DEV u64 fold128(u128 z) const {
        const u64 lo = (u64)z;
        const unsigned c16 = c << 4;
        const u64 t1 = (u64)(unsigned)(z >> 96) * c16;  // < 2^60; t1 2^32 split at 2^60 as above
        u64 acc = (lo & ((1ull << 60) - 1)) + (u64)(unsigned)(lo >> 60) * c;
        acc += (u64)(unsigned)(z >> 64) * c16;  // < 2^60
        acc += (u64)(unsigned)(t1 >> 28) * c;   // < 2^56
        return acc + ((u64)((unsigned)t1 & 0x0FFFFFFFu) << 32);  // + (< 2^60)
    }
    DEV u64 canon128(u128 z) const {
        const u64 x = fold(fold128(z));  // <= 2^60 + 3c < 2q
        return x >= q ? x - q : x;
    }
    // Shoup product x w - hi' q modulo 2^64 for ANY 64-bit x, in [0, 4q): -hi' q = hi' c - hi' 2^60, one 32 x 32 multiply-add onto
    // x w, and the high dword takes hi1 c - (hi0 << 28) — two multiplies instead of the three of a general 64 x 64 low product, no
    // borrow chain
    DEV u64 shoup(u64 x, const TW W) const {
#ifdef HYDIA_INTP_EXACT_QUOTIENT  // A/B builds only (tools/ab/): the exact quotient — same folds, four more instructions per butterfly
        const u64 hi = __umul64hi(x, W.y);
#else
        const unsigned xl = (unsigned)x, xh = (unsigned)(x >> 32), wl = (unsigned)W.y, wh = (unsigned)(W.y >> 32);
        const u64 hi = (u64)xh * wh + ((u64)__umulhi(xl, wh) + (u64)__umulhi(xh, wl));
#endif
        const unsigned h0 = (unsigned)hi, h1 = (unsigned)(hi >> 32);
        const u64 t = x * W.x + (u64)h0 * c;
        return ((u64)((unsigned)(t >> 32) + h1 * c - (h0 << 28)) << 32) | (unsigned)t;  // the high dword alone: no 64-bit carry chain
    }
    DEV void ct(T &a, T &b, const TW W) const {  // [0, B) -> [0, B + 4q), B <= 12q
        const u64 t = shoup(b, W);
        const u64 u = a;
        a = u + t;
        b = u - t + q4;
    }
    DEV void gs(T &a, T &b, const TW W) const {  // a, b < 8q -> a < 16q, b < 4q
        const u64 s = a + b;
        const u64 d = a - b + q8;
        b = shoup(d, W);
        a = s;
    }
    DEV void recentre(T &x) const { x = fold(x); }
    DEV void recentre_wide(T &x) const { x = fold(x); }
    DEV void mid(T &x) const { x = fold(x); }
    DEV void fwd_fold(T &x) const { x = fold(x); }
    DEV T from_raw(u64 x) const { return fold(x); }
    DEV u64 fin_fwd(T x) const {
        x = fold(x);
        return x >= q ? x - q : x;
    }
    DEV u64 fin_inv(T x, u64 sc, u64 scs) const { return mulmod_shoup(x, sc, scs, q); }
};

struct FpA {
    typedef double T;
    typedef double2 TW;  // (w, w / q)
    double q, qinv;
    bool lean;  // q < 2^45 (1 + 1/16), true for every scaling prime of the default chain: 32 q stays below 2^50.1
    DEV FpA(const ModC &M) : q((double)M.q), qinv(1.0 / (double)M.q), lean(M.q < (1ull << 45) + (1ull << 41)) {}
    DEV static TW tw(const ulonglong2 b) { return make_double2(__longlong_as_double((long long)b.x), __longlong_as_double((long long)b.y)); }
    // twiddle given alone: w / q as w * (1/q).  The quotient estimate of mulmod may then be off by one more in rare cases: the result
    // stays an exact representative of a*w (|r| <= ~1.3 q instead of 0.75 q), which the transforms' headroom (< 2^52) absorbs
    DEV TW tw8(const double wv) const { return make_double2(wv, wv * qinv); }
    // integer <-> double without the emulated 64-bit conversions: for 0 <= x < 2^52 the bit pattern 0x433.. | x IS the double 2^52 + x
    DEV static double u2d(u64 x) { return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 4503599627370496.0; }
    DEV static u64 d2u(double r) { return (u64)__double_as_longlong(r + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull; }  // r integral in [0, 2^52)
    DEV T from_canon(u64 x) const { return u2d(x); }  // x < 2^47: exact
    DEV static T from_bits(u64 x) { return __longlong_as_double((long long)x); }
    DEV static u64 to_bits(T x) { return (u64)__double_as_longlong(x); }
    DEV double mulmod(const double v, const TW W) const {  // exact v*w - c*q with |result| <= 0.75 q
        const double h = v * W.x;
        const double l = __fma_rn(v, W.x, -h);
        const double c = rint(v * W.y);
        return __fma_rn(-c, q, h) + l;
    }
    // exact a*b - c*q for canonical a, b < q < 2^47 (no precomputed b / q): |result| <= 0.55 q
    DEV double mulmod2(const double a, const double b) const {
        const double h = a * b;
        const double l = __fma_rn(a, b, -h);
        const double c = rint(h * qinv);
        return __fma_rn(-c, q, h) + l;
    }
    DEV void ct(T &a, T &b, const TW W) const {
        const double r = mulmod(b, W);
        b = a - r;
        a = a + r;
    }
    DEV void gs(T &a, T &b, const TW W) const {
        const double s = a + b, d = a - b;
        b = mulmod(d, W);
        a = s;
    }
    DEV void recentre(T &x) const { x = __fma_rn(-rint(x * qinv), q, x); }  // -> [-q/2, q/2]
    // The inverse transform doubles magnitudes on its sum path; a reduction every 3-4 stages keeps 47-bit primes below 2^52.  For
    // the lean primes two of the four reductions of the two-pass inverse can go: runs of 5 and 6 stages reach 32 q < 2^50.1, where
    // products and quotient estimates are still exact enough (|quotient error| <= 1, results exact).  Final residues are unchanged.
    DEV void recentre_wide(T &x) const {
        if (!lean) recentre(x);
    }
    DEV void mid(T &) const {}
    DEV void fwd_fold(T &) const {}
    DEV T from_raw(u64 x) const { return from_bits(x); }
    DEV u64 fin_fwd(T x) const {
        recentre(x);
        if (x < 0) x += q;
        return d2u(x);
    }
    DEV u64 fin_inv(T x, u64 sc, u64) const {
        const double s = u2d(sc);
        double r = mulmod(x, make_double2(s, s * qinv));
        if (r < 0) r += q;
        return d2u(r);
    }
};

}  // namespace

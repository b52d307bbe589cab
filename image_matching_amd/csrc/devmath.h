// image_matching_amd/csrc/devmath.h — 64-bit modular arithmetic for gfx950 device code (and host table building).
//
// CDNA4 has no 64x64->128 multiplier: every product below lowers to v_mad_u64_u32 / v_mul_hi_u32 chains.  The
// forms are picked to minimise those: single-word Barrett for reduced operands (one 128-bit product, one high
// product, one low product), Shoup for fixed multipliers (NTT twiddles), and ONE double-word Barrett per output for
// lazily accumulated 128-bit sums (tensor / inner-product / base-conversion kernels).
#pragma once
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned __int128 u128;

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#define DEV __device__ __forceinline__
#else
#define HD inline
#define DEV inline
#endif

#define HY_MAX_MODS 32   // Q limbs + P limbs
#define HY_MAX_DIGIT 8   // alpha (limbs per key-switching digit)

// Per-modulus constants; an array of these lives in __constant__-like global memory, indexed by modulus id.
struct ModC {
    u64 q;
    u64 mu;      // floor(2^(k+62) / q), k = bit length of q
    u64 r64;     // floor(2^64 / q)
    u64 r0, r1;  // floor(2^128 / q) = r1 * 2^64 + r0
    u64 ninv, ninv_sh;  // N^{-1} mod q and its Shoup companion
    int ks;      // k - 2
    int pad;
};

HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}
HD u64 addmod(u64 a, u64 b, u64 q) {
    u64 r = a + b;
    return r >= q ? r - q : r;
}
HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
HD u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// a*b mod q for a, b < 2^k (in particular a, b < q): quotient estimate is off by at most one.
HD u64 mulmod(u64 a, u64 b, const ModC &M) {
    u128 z = (u128)a * b;
    u64 x = (u64)(z >> M.ks);
    u64 qh = mulhi64(x, M.mu);
    u64 r = (u64)z - qh * M.q;
    return r >= M.q ? r - M.q : r;
}
// any 64-bit a -> a mod q
HD u64 reduce64(u64 a, const ModC &M) {
    u64 qh = mulhi64(a, M.r64);
    u64 r = a - qh * M.q;
    return r >= M.q ? r - M.q : r;
}
// any 128-bit z -> z mod q (quotient estimate off by at most two)
HD u64 reduce128(u128 z, const ModC &M) {
    u64 z0 = (u64)z, z1 = (u64)(z >> 64);
    u64 t = mulhi64(z0, M.r0);
    u128 m1 = (u128)z0 * M.r1;
    u128 m2 = (u128)z1 * M.r0;
    u128 mid = (u128)t + (u64)m1 + (u64)m2;
    u64 qh = (u64)(mid >> 64) + (u64)(m1 >> 64) + (u64)(m2 >> 64) + z1 * M.r1;
    u64 r = z0 - qh * M.q;
    if (r >= M.q) r -= M.q;
    if (r >= M.q) r -= M.q;
    return r;
}
// z mod q for z < 2^(k+62), k = bit length of q — the range of every lazy sum of at most FOUR products of reduced operands
// (4 (q-1)^2 < 2^(2k+2) <= 2^(k+62) for k <= 60): single-word Barrett on the top bits.  With x = floor(z / 2^(k-2)) < 2^64
// and mu = floor(2^(k+62) / q), floor(x mu / 2^64) is the true quotient or up to two below it, so two conditional
// subtractions finish.  7 multiplier instructions instead of reduce128's 18.
HD u64 reduce128k(u128 z, const ModC &M) {
    const u64 x = (u64)(z >> M.ks);
    const u64 qh = mulhi64(x, M.mu);
    u64 r = (u64)z - qh * M.q;
    if (r >= M.q) r -= M.q;
    if (r >= M.q) r -= M.q;
    return r;
}
// lazy sum of `nprod` products of reduced operands (+ at most one reduced addend per 4 products): the cheap form whenever its
// range holds — up to 4 products for any modulus, up to 2^16 for the <= 46-bit ones
HD u64 reduce_lazy(u128 z, const ModC &M, int nprod) {
    return (M.ks <= 44 || nprod <= 4) ? reduce128k(z, M) : reduce128(z, M);
}
// a*w mod q with the precomputed ws = floor(w * 2^64 / q); any 64-bit a
HD u64 mulmod_shoup(u64 a, u64 w, u64 ws, u64 q) {
    u64 hi = mulhi64(a, ws);
    u64 r = a * w - hi * q;
    return r >= q ? r - q : r;
}

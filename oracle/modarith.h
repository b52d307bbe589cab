/*
 * oracle/modarith.h — TEST INFRASTRUCTURE ONLY (CPU oracle of the HyDia hot path).
 *
 * 64-bit modular arithmetic on unsigned __int128, for moduli q < 2^62.
 * Nothing here is taken from the reference (it has no arithmetic of its own: every modular
 * operation happens inside the un-vendored OpenFHE v1.2.3, /root/reference/dockerfile:7,25-30);
 * these are the textbook Barrett / Shoup forms.
 */
#ifndef HYDIA_ORACLE_MODARITH_H
#define HYDIA_ORACLE_MODARITH_H

#include <stdint.h>

typedef uint64_t u64;
typedef int64_t i64;
typedef unsigned __int128 u128;
typedef __int128 i128;

static inline u64 addmod(u64 a, u64 b, u64 q) {
    u64 r = a + b;
    return r >= q ? r - q : r;
}
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

/* Slow, obviously-correct product (used for table building and as the checker of the fast forms). */
static inline u64 mulmod_slow(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }

/* Barrett constant for reducing a 128-bit value: floor(2^128 / q) as two words. */
typedef struct {
    u64 q;
    u64 r0, r1; /* floor(2^128/q) = r1*2^64 + r0 */
} barrett_t;

static inline barrett_t barrett_make(u64 q) {
    barrett_t b;
    b.q = q;
    /* 2^128 / q by long division in two 64-bit steps */
    u128 num = ((u128)1 << 64);      /* 2^64 */
    u128 hi = num / q;               /* floor(2^64/q) */
    u128 rem = num % q;
    u128 lo = (rem << 64) / q;       /* next 64 quotient bits */
    u128 full = (hi << 64) + lo;     /* floor(2^128/q) (q is not a power of two) */
    b.r0 = (u64)full;
    b.r1 = (u64)(full >> 64);
    return b;
}

/* z mod q for any 128-bit z. */
static inline u64 barrett_reduce128(u128 z, const barrett_t *b) {
    u64 z0 = (u64)z, z1 = (u64)(z >> 64);
    /* quotient estimate = floor(z * ratio / 2^128), low 64 bits suffice (quotient < 2^128/q * ... fits) */
    u128 t = ((u128)z0 * b->r0) >> 64;
    u128 m1 = (u128)z0 * b->r1;
    u128 m2 = (u128)z1 * b->r0;
    u128 mid = t + (u64)m1 + (u64)m2;
    u64 qhat = (u64)(mid >> 64) + (u64)(m1 >> 64) + (u64)(m2 >> 64) + z1 * b->r1;
    u64 r = z0 - qhat * b->q;
    while (r >= b->q) r -= b->q;
    return r;
}

static inline u64 mulmod(u64 a, u64 b, const barrett_t *bq) { return barrett_reduce128((u128)a * b, bq); }

/* Shoup form: w' = floor(w * 2^64 / q); a*w mod q with one high product (valid for any a < 2^64, q < 2^63). */
static inline u64 shoup_pre(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static inline u64 mulmod_shoup(u64 a, u64 w, u64 wp, u64 q) {
    u64 hi = (u64)(((u128)a * wp) >> 64);
    u64 r = a * w - hi * q;
    return r >= q ? r - q : r;
}

static inline u64 powmod(u64 a, u64 e, u64 q) {
    u64 r = 1;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod_slow(r, a, q);
        a = mulmod_slow(a, a, q);
        e >>= 1;
    }
    return r;
}
static inline u64 invmod(u64 a, u64 q) { return powmod(a, q - 2, q); } /* q prime */

static inline uint32_t bitrev32(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

#endif

// Host check of image_matching_amd/csrc/devmath.h against unsigned __int128 %: every reduction form, on the moduli the
// engine uses (60-bit first/special primes, 45/46-bit scaling primes) plus small and awkward ones, at range edges and
// on random operands.  Built and run by tests/test_devmath_host.py.
#include <cstdio>
#include <cstdlib>
#include "devmath.h"

static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static ModC make(u64 q) {
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
    return m;
}
int main() {
    const u64 mods[] = {0xffffffffffc0001ull, 0xfffffffff840001ull, 35184372744193ull, 35184371138561ull, 35184373006337ull,
                        (1ull << 59) + 1769473ull * 0 + 2621441ull, 65537ull, 1152921504606584833ull, 70368743489537ull, 12289ull,
                        (1ull << 47) - 115ull, (1ull << 60) - 93ull, 3ull << 58 | 1ull};
    long checks = 0;
    for (u64 q : mods) {
        const ModC M = make(q);
        const int k = 64 - __builtin_clzll(q);
        const u64 edge[] = {0, 1, 2, q - 1, q - 2, q / 2, q / 2 + 1, (1ull << (k - 1)), (1ull << (k - 1)) - 1};
        auto check_pair = [&](u64 a, u64 b) {
            a %= q; b %= q;
            const u64 want = (u64)(((u128)a * b) % q);
            if (mulmod(a, b, M) != want) { printf("mulmod q=%llu a=%llu b=%llu\n", q, a, b); exit(1); }
            const u64 ws = (u64)((((u128)b) << 64) / q);
            if (mulmod_shoup(a, b, ws, q) != want) { printf("shoup q=%llu\n", q); exit(1); }
            checks += 2;
        };
        for (u64 a : edge) for (u64 b : edge) check_pair(a, b);
        for (int i = 0; i < 200000; i++) check_pair(rnd(), rnd());
        // reduce64: any 64-bit value
        const u64 e64[] = {0, 1, q - 1, q, q + 1, 2 * q - 1, 2 * q, ~0ull, ~0ull - 1, 1ull << 63};
        for (u64 a : e64) if (reduce64(a, M) != a % q) { printf("reduce64 q=%llu a=%llu\n", q, a); exit(1); }
        for (int i = 0; i < 200000; i++) { u64 a = rnd(); if (reduce64(a, M) != a % q) { printf("reduce64 q=%llu\n", q); exit(1); } checks++; }
        // reduce128: any 128-bit value
        for (int i = 0; i < 200000; i++) {
            u128 z = ((u128)rnd() << 64) | rnd();
            if (i < 64) z >>= i;           // all magnitudes
            if (i == 64) z = ~(u128)0;
            if (reduce128(z, M) != (u64)(z % q)) { printf("reduce128 q=%llu\n", q); exit(1); }
            checks++;
        }
        // reduce128k: z < 2^(k+62); in particular sums of up to four products of reduced operands
        const u128 lim = ((u128)1) << (k + 62);
        const u128 ek[] = {0, 1, q - 1, q, (u128)q * q, lim - 1, lim - q, lim / 2, 4 * (u128)(q - 1) * (q - 1)};
        for (u128 z : ek) if (z < lim && reduce128k(z, M) != (u64)(z % q)) { printf("reduce128k edge q=%llu\n", q); exit(1); }
        for (int i = 0; i < 400000; i++) {
            u128 z;
            if (i & 1) {
                z = 0;
                for (int t = 0; t < 1 + (i >> 1) % 4; t++) z += (u128)(rnd() % q) * (rnd() % q);
                if (i % 7 == 0) z = 4 * (u128)(q - 1 - (rnd() % 3)) * (q - 1 - (rnd() % 3));
            } else {
                z = (((u128)rnd() << 64) | rnd()) % lim;
                if (i % 5 == 0) z = lim - 1 - (rnd() % 1000);
            }
            if (reduce128k(z, M) != (u64)(z % q)) { printf("reduce128k q=%llu i=%d\n", q, i); exit(1); }
            checks++;
        }
    }
    printf("devmath ok: %ld checks\n", checks);
    return 0;
}

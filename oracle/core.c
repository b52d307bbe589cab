/*
 * oracle/core.c — TEST INFRASTRUCTURE ONLY (see hydia_oracle.h).
 *
 * RNS context (what OpenFHE's GenCryptoContext derives from /root/reference/src/main.cpp:169-173:
 * HEStd_128_classic, depth 11, ScalingModSize 45, FIXEDMANUAL -> N = 2^15, one 60-bit + eleven 45-bit
 * Q primes, HYBRID key switching with dnum = 3 and four 60-bit P primes), negacyclic NTT, the
 * evaluation-form automorphism, and the ChaCha20-addressed samplers.  Prime ORDER and root choice are
 * this build's own deterministic rule (OpenFHE's is not recoverable offline); they change no decrypted
 * result.
 */
#define _GNU_SOURCE /* sincos */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hydia_oracle.h"

/* ------------------------------------------------------------------ primes */
static int is_prime_u64(u64 n) {
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (int i = 0; i < 12; i++) {
        if (n % small[i] == 0) return n == small[i];
    }
    u64 d = n - 1;
    int r = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        r++;
    }
    for (int i = 0; i < 12; i++) { /* these 12 bases are a deterministic test for all n < 2^64 */
        u64 x = powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = mulmod_slow(x, x, n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}
/* smallest prime >= lo with prime = 1 mod m */
static u64 prime_at_or_above(u64 lo, u64 m) {
    u64 c = lo + ((m + 1 - lo % m) % m);
    while (!is_prime_u64(c)) c += m;
    return c;
}
/* largest prime < hi with prime = 1 mod m */
static u64 prime_below(u64 hi, u64 m) {
    u64 c = hi - 1;
    c -= (c % m + m - 1) % m; /* largest value <= hi-1 that is 1 mod m */
    while (!is_prime_u64(c)) c -= m;
    return c;
}
static u64 primitive_root_2n(u64 q, u64 m) {
    for (u64 x = 2;; x++) {
        u64 r = powmod(x, (q - 1) / m, q);
        if (powmod(r, m / 2, q) == q - 1) return r;
    }
}

static void params_finish(hy_params *p, const u64 *roots);

/* A context on a caller-supplied prime chain (an OpenFHE context's moduli read through GetElementParams()):
 * moduli = nQ ciphertext primes (q_0 first) then nP special primes; roots (optional) = the 2N-th roots to use. */
hy_params *hyo_params_create_custom(int logN, int nQ, int nP, int scale_bits, int dnum, int dim, const u64 *moduli,
                                    const u64 *roots) {
    if (nQ < 2 || nP < 1 || nQ + nP > HY_MAX_LIMBS) return NULL;
    u64 M = 2ull << logN;
    for (int m = 0; m < nQ + nP; m++) {
        if (moduli[m] % M != 1 || (moduli[m] >> 60) || !is_prime_u64(moduli[m])) return NULL;
        for (int i = 0; i < m; i++)
            if (moduli[i] == moduli[m]) return NULL;
        if (roots && (powmod(roots[m], M / 2, moduli[m]) != moduli[m] - 1)) return NULL;
    }
    hy_params *p = (hy_params *)calloc(1, sizeof(hy_params));
    p->logN = logN;
    p->N = 1 << logN;
    p->slots = p->N / 2;
    p->nQ = nQ;
    p->nP = nP;
    p->nT = nQ + nP;
    p->dnum = dnum;
    p->alpha = (nQ + dnum - 1) / dnum;
    p->scale_bits = scale_bits;
    p->first_bits = 0;
    while ((moduli[0] >> p->first_bits) != 0) p->first_bits++;
    p->dim = dim;
    p->delta = ldexp(1.0, scale_bits);
    for (int m = 0; m < p->nT; m++) p->q[m] = moduli[m];
    params_finish(p, roots);
    return p;
}

hy_params *hyo_params_create(int logN, int mult_depth, int scale_bits, int first_bits, int dnum, int dim) {
    hy_params *p = (hy_params *)calloc(1, sizeof(hy_params));
    p->logN = logN;
    p->N = 1 << logN;
    p->slots = p->N / 2;
    p->nQ = mult_depth + 1;
    p->dnum = dnum;
    p->alpha = (p->nQ + dnum - 1) / dnum;
    p->scale_bits = scale_bits;
    p->first_bits = first_bits;
    p->dim = dim;
    p->delta = ldexp(1.0, scale_bits);
    u64 M = 2ull * p->N;

    /* scaling primes: alternate around 2^scale_bits, assigned from the LAST limb downwards */
    u64 up = prime_at_or_above(1ull << scale_bits, M), dn = up;
    int nscale = p->nQ - 1;
    for (int i = 0; i < nscale; i++) {
        int j = p->nQ - 1 - i;
        if (i == 0) {
            p->q[j] = up;
        } else if (i & 1) {
            dn = prime_below(dn, M);
            p->q[j] = dn;
        } else {
            up = prime_at_or_above(up + 1, M);
            p->q[j] = up;
        }
    }
    /* first modulus: largest prime below 2^first_bits; P primes continue downwards from it */
    u64 cur = prime_below(1ull << first_bits, M);
    if (first_bits == scale_bits) { /* keep all primes distinct */
        while (1) {
            int clash = 0;
            for (int j = 1; j < p->nQ; j++) clash |= (p->q[j] == cur);
            if (!clash) break;
            cur = prime_below(cur, M);
        }
    }
    p->q[0] = cur;
    /* P: alpha primes of 60 bits (ceil(alpha*max_q_bits / 60) in general) */
    int digit_bits = 0;
    for (int j = 0; j < p->alpha && j < p->nQ; j++) digit_bits += (j == 0 ? first_bits : scale_bits + 1);
    p->nP = (digit_bits + 59) / 60;
    if (p->nP < 1) p->nP = 1;
    u64 pc = (first_bits == 60) ? p->q[0] : (1ull << 60);
    for (int k = 0; k < p->nP; k++) {
        pc = prime_below(pc, M);
        p->q[p->nQ + k] = pc;
    }
    p->nT = p->nQ + p->nP;
    params_finish(p, NULL);
    return p;
}

/* everything derived from the prime chain */
static void params_finish(hy_params *p, const u64 *roots) {
    int N = p->N, logN = p->logN;
    u64 M = 2ull * p->N;
    for (int m = 0; m < p->nT; m++) {
        u64 q = p->q[m];
        p->bq[m] = barrett_make(q);
        p->psi[m] = roots ? roots[m] : primitive_root_2n(q, M);
        p->psi_inv[m] = invmod(p->psi[m], q);
        p->n_inv[m] = invmod((u64)N, q);
        p->n_inv_sh[m] = shoup_pre(p->n_inv[m], q);
        p->tw[m] = (u64 *)malloc(sizeof(u64) * N);
        p->tw_sh[m] = (u64 *)malloc(sizeof(u64) * N);
        p->itw[m] = (u64 *)malloc(sizeof(u64) * N);
        p->itw_sh[m] = (u64 *)malloc(sizeof(u64) * N);
        u64 *pw = (u64 *)malloc(sizeof(u64) * N), *ipw = (u64 *)malloc(sizeof(u64) * N);
        pw[0] = ipw[0] = 1;
        for (int i = 1; i < N; i++) {
            pw[i] = mulmod_slow(pw[i - 1], p->psi[m], q);
            ipw[i] = mulmod_slow(ipw[i - 1], p->psi_inv[m], q);
        }
        for (int k = 0; k < N; k++) {
            uint32_t r = bitrev32((uint32_t)k, logN);
            p->tw[m][k] = pw[r];
            p->tw_sh[m][k] = shoup_pre(pw[r], q);
            p->itw[m][k] = ipw[r];
            p->itw_sh[m][k] = shoup_pre(ipw[r], q);
        }
        free(pw);
        free(ipw);
    }
    /* P mod q_j, P^{-1} mod q_j, P-basis CRT factors */
    for (int j = 0; j < p->nQ; j++) {
        u64 q = p->q[j], prod = 1;
        for (int k = 0; k < p->nP; k++) prod = mulmod_slow(prod, p->q[p->nQ + k] % q, q);
        p->P_mod_q[j] = prod;
        p->Pinv_mod_q[j] = invmod(prod, q);
    }
    for (int k = 0; k < p->nP; k++) {
        u64 pk = p->q[p->nQ + k], prod = 1;
        for (int i = 0; i < p->nP; i++)
            if (i != k) prod = mulmod_slow(prod, p->q[p->nQ + i] % pk, pk);
        p->Phat_inv[k] = invmod(prod, pk);
        for (int j = 0; j < p->nQ; j++) {
            u64 q = p->q[j], pr = 1;
            for (int i = 0; i < p->nP; i++)
                if (i != k) pr = mulmod_slow(pr, p->q[p->nQ + i] % q, q);
            p->Phat_mod_q[k][j] = pr;
        }
    }
    for (int l = 0; l < p->nQ; l++)
        for (int j = 0; j < l; j++) p->ql_inv[l][j] = invmod(p->q[l] % p->q[j], p->q[j]);

    /* canonical-embedding tables */
    p->rot_group = (uint32_t *)malloc(sizeof(uint32_t) * p->slots);
    u64 g = 1;
    for (int j = 0; j < p->slots; j++) {
        p->rot_group[j] = (uint32_t)g;
        g = (g * 5) % M;
    }
    p->ksi_re = (double *)malloc(sizeof(double) * (M + 1));
    p->ksi_im = (double *)malloc(sizeof(double) * (M + 1));
    for (u64 k = 0; k <= M; k++) {
        /* glibc sincos() on both sides of the parity tests: a compiler may or may not fuse separate sin()/cos()
         * calls into sincos(), and the two entry points are not guaranteed to round identically */
        double ang = 2.0 * M_PI * (double)k / (double)M, sn, cs;
        sincos(ang, &sn, &cs);
        p->ksi_re[k] = cs;
        p->ksi_im[k] = sn;
    }
}

void hyo_params_free(hy_params *p) {
    if (!p) return;
    for (int m = 0; m < p->nT; m++) {
        free(p->tw[m]);
        free(p->tw_sh[m]);
        free(p->itw[m]);
        free(p->itw_sh[m]);
    }
    free(p->rot_group);
    free(p->ksi_re);
    free(p->ksi_im);
    free(p);
}
void hyo_get_moduli(const hy_params *p, u64 *out) {
    for (int m = 0; m < p->nT; m++) out[m] = p->q[m];
}
void hyo_get_roots(const hy_params *p, u64 *out) {
    for (int m = 0; m < p->nT; m++) out[m] = p->psi[m];
}
int hyo_get_info(const hy_params *p, int *o) {
    o[0] = p->logN;
    o[1] = p->N;
    o[2] = p->nQ;
    o[3] = p->nP;
    o[4] = p->dnum;
    o[5] = p->alpha;
    o[6] = p->dim;
    o[7] = p->slots;
    return 0;
}

/* ------------------------------------------------------------------ NTT
 * Forward: Cooley-Tukey, natural-order input -> bit-reversed output; out[j] = a(psi^{2*bitrev(j)+1}).
 * Inverse: Gentleman-Sande, bit-reversed input -> natural order, scaled by N^{-1}. */
void hyo_ntt_fwd(const hy_params *p, u64 *a, int m) {
    const u64 q = p->q[m];
    const u64 *w = p->tw[m], *ws = p->tw_sh[m];
    int N = p->N, t = N;
    for (int mm = 1; mm < N; mm <<= 1) {
        t >>= 1;
        for (int i = 0; i < mm; i++) {
            int j1 = 2 * i * t;
            u64 W = w[mm + i], Ws = ws[mm + i];
            for (int j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = mulmod_shoup(a[j + t], W, Ws, q);
                a[j] = addmod(U, V, q);
                a[j + t] = submod(U, V, q);
            }
        }
    }
}
void hyo_ntt_inv(const hy_params *p, u64 *a, int m) {
    const u64 q = p->q[m];
    const u64 *w = p->itw[m], *ws = p->itw_sh[m];
    int N = p->N, t = 1;
    for (int mm = N; mm > 1; mm >>= 1) {
        int h = mm >> 1, j1 = 0;
        for (int i = 0; i < h; i++) {
            u64 W = w[h + i], Ws = ws[h + i];
            for (int j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = a[j + t];
                a[j] = addmod(U, V, q);
                a[j + t] = mulmod_shoup(submod(U, V, q), W, Ws, q);
            }
            j1 += 2 * t;
        }
        t <<= 1;
    }
    for (int j = 0; j < N; j++) a[j] = mulmod_shoup(a[j], p->n_inv[m], p->n_inv_sh[m], q);
}

/* Galois element of a left rotation by `rot` slots: 5^rot mod 2N (rot may be negative). */
u64 hyo_galois_elt(const hy_params *p, int rot) {
    u64 M = 2ull * p->N;
    int r = ((rot % p->slots) + p->slots) % p->slots;
    u64 g = 1;
    for (int i = 0; i < r; i++) g = (g * 5) % M;
    return g;
}
/* sigma_g on evaluation form is a pure index permutation: out[j] = in[j'] with
 * 2*bitrev(j')+1 = g*(2*bitrev(j)+1) mod 2N  (what OpenFHE's PrecomputeAutoMap tabulates). */
void hyo_automorph_eval(const hy_params *p, const u64 *in, u64 *out, u64 g) {
    int N = p->N, logN = p->logN;
    u64 mask = 2ull * N - 1;
    for (int j = 0; j < N; j++) {
        u64 e = 2ull * bitrev32((uint32_t)j, logN) + 1;
        u64 e2 = (e * g) & mask;
        uint32_t jp = bitrev32((uint32_t)((e2 - 1) >> 1), logN);
        out[j] = in[jp];
    }
}
/* coefficient form: X^i -> X^{i g mod 2N}, with X^N = -1 */
void hyo_automorph_coeff(const hy_params *p, const u64 *in, u64 *out, u64 g, u64 q) {
    int N = p->N;
    u64 mask = 2ull * N - 1;
    for (int i = 0; i < N; i++) {
        u64 e = ((u64)i * g) & mask;
        if (e < (u64)N)
            out[e] = in[i];
        else
            out[e - N] = negmod(in[i], q);
    }
}

/* ------------------------------------------------------------------ ChaCha20 */
#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d) \
    a += b; d ^= a; d = ROTL32(d, 16); \
    c += d; b ^= c; b = ROTL32(b, 12); \
    a += b; d ^= a; d = ROTL32(d, 8);  \
    c += d; b ^= c; b = ROTL32(b, 7);

void hyo_chacha_block(const uint8_t seed[32], u64 stream, u64 block, uint32_t out[16]) {
    uint32_t s[16], x[16];
    s[0] = 0x61707865; s[1] = 0x3320646e; s[2] = 0x79622d32; s[3] = 0x6b206574;
    for (int i = 0; i < 8; i++)
        s[4 + i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) |
                   ((uint32_t)seed[4 * i + 3] << 24);
    s[12] = (uint32_t)block; s[13] = (uint32_t)(block >> 32);
    s[14] = (uint32_t)stream; s[15] = (uint32_t)(stream >> 32);
    memcpy(x, s, sizeof(s));
    for (int r = 0; r < 10; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

/* coefficient c = 4*block + t takes words 4t..4t+3 as a 128-bit integer, reduced mod q (bias < 2^-67) */
void hyo_sample_uniform(const uint8_t seed[32], u64 stream, u64 q, u64 *out, int n) {
    barrett_t b = barrett_make(q);
    for (int blk = 0; blk * 4 < n; blk++) {
        uint32_t w[16];
        hyo_chacha_block(seed, stream, (u64)blk, w);
        for (int t = 0; t < 4 && blk * 4 + t < n; t++) {
            u64 lo = (u64)w[4 * t] | ((u64)w[4 * t + 1] << 32);
            u64 hi = (u64)w[4 * t + 2] | ((u64)w[4 * t + 3] << 32);
            out[blk * 4 + t] = barrett_reduce128(((u128)hi << 64) | lo, &b);
        }
    }
}
/* coefficient c = 16*block + t takes word t: floor(3*w / 2^32) - 1 in {-1,0,1} */
void hyo_sample_ternary(const uint8_t seed[32], u64 stream, int8_t *out, int n) {
    for (int blk = 0; blk * 16 < n; blk++) {
        uint32_t w[16];
        hyo_chacha_block(seed, stream, (u64)blk, w);
        for (int t = 0; t < 16 && blk * 16 + t < n; t++) out[blk * 16 + t] = (int8_t)((((u64)w[t] * 3) >> 32)) - 1;
    }
}
#include "gauss_cdt.h"
static const u64 GAUSS_CDT[HYDIA_GAUSS_CDT_LEN] = HYDIA_GAUSS_CDT_VALUES;
/* coefficient c = 8*block + t takes the 64-bit r = w[2t] | w[2t+1]<<32: sign = r&1, magnitude by CDT on r>>1 */
void hyo_sample_gauss(const uint8_t seed[32], u64 stream, int32_t *out, int n) {
    for (int blk = 0; blk * 8 < n; blk++) {
        uint32_t w[16];
        hyo_chacha_block(seed, stream, (u64)blk, w);
        for (int t = 0; t < 8 && blk * 8 + t < n; t++) {
            u64 r = (u64)w[2 * t] | ((u64)w[2 * t + 1] << 32);
            u64 u = r >> 1;
            int k = 0;
            while (u >= GAUSS_CDT[k]) k++;
            out[blk * 8 + t] = (r & 1) ? -k : k;
        }
    }
}

/*
 * oracle/ckks.c — TEST INFRASTRUCTURE ONLY (see hydia_oracle.h).
 *
 * CKKS-RNS client and evaluator: the OpenFHE calls the reference makes on its HyDia path
 * (SURVEY.md §2.2), restated from the published algorithms:
 *   MakeCKKSPackedPlaintext + Encrypt   <- src/openFHE_wrapper.cpp:74-77
 *   Decrypt + GetRealPackedValue        <- src/openFHE_wrapper.cpp:81-85
 *   KeyGen / EvalMultKeyGen / EvalRotateKeyGen <- src/main.cpp:184-206
 *   EvalFastRotationPrecompute / EvalFastRotation <- src/sender/sender_diag.cpp:22,25
 *   EvalMultNoRelin / EvalAddInPlace / RelinearizeInPlace / RescaleInPlace <- sender_diag.cpp:76-80,93
 * Hybrid key switching (Han-Ki): digits of alpha limbs, ModUp to Q_l u P, inner product with the key,
 * ModDown by P.  FIXEDMANUAL scaling: scale tracked as a double, Rescale divides by the dropped prime.
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hydia_oracle.h"

#define PARFOR _Pragma("omp parallel for schedule(static) if (!omp_in_parallel())")

int hyo_num_threads(void) { return omp_get_max_threads(); }
/* size of the OpenMP teams of every later parallel region (bench.py: the CPU share this job really has) */
void hyo_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

/* ------------------------------------------------------------------ encode / decode */
typedef struct { double re, im; } cplx;

static void bitrev_cplx(cplx *v, int size) {
    int bits = 0;
    while ((1 << bits) < size) bits++;
    for (int i = 0; i < size; i++) {
        int j = (int)bitrev32((uint32_t)i, bits);
        if (i < j) {
            cplx t = v[i];
            v[i] = v[j];
            v[j] = t;
        }
    }
}
/* slots -> coefficients direction of the canonical embedding (HEAAN "fftSpecialInv") */
static void fft_special_inv(const hy_params *p, cplx *v, int size) {
    int M = 2 * p->N;
    for (int len = size; len >= 2; len >>= 1) {
        int lenh = len >> 1, lenq = len << 2, gap = M / lenq;
        for (int i = 0; i < size; i += len)
            for (int j = 0; j < lenh; j++) {
                int idx = (lenq - (int)(p->rot_group[j] % (uint32_t)lenq)) * gap;
                double wr = p->ksi_re[idx], wi = p->ksi_im[idx];
                cplx a = v[i + j], b = v[i + j + lenh];
                double ur = a.re + b.re, ui = a.im + b.im;
                double dr = a.re - b.re, di = a.im - b.im;
                double m0 = dr * wr, m1 = di * wi, m2 = dr * wi, m3 = di * wr;
                v[i + j].re = ur;
                v[i + j].im = ui;
                v[i + j + lenh].re = m0 - m1;
                v[i + j + lenh].im = m2 + m3;
            }
    }
    bitrev_cplx(v, size);
    double inv = 1.0 / (double)size;
    for (int i = 0; i < size; i++) {
        v[i].re *= inv;
        v[i].im *= inv;
    }
}
/* coefficients -> slots ("fftSpecial") */
static void fft_special(const hy_params *p, cplx *v, int size) {
    int M = 2 * p->N;
    bitrev_cplx(v, size);
    for (int len = 2; len <= size; len <<= 1) {
        int lenh = len >> 1, lenq = len << 2, gap = M / lenq;
        for (int i = 0; i < size; i += len)
            for (int j = 0; j < lenh; j++) {
                int idx = (int)(p->rot_group[j] % (uint32_t)lenq) * gap;
                double wr = p->ksi_re[idx], wi = p->ksi_im[idx];
                cplx a = v[i + j], b = v[i + j + lenh];
                double m0 = b.re * wr, m1 = b.im * wi, m2 = b.re * wi, m3 = b.im * wr;
                double tr = m0 - m1, ti = m2 + m3;
                v[i + j].re = a.re + tr;
                v[i + j].im = a.im + ti;
                v[i + j + lenh].re = a.re - tr;
                v[i + j + lenh].im = a.im - ti;
            }
    }
}

void hyo_encode_coeffs(const hy_params *p, const double *slots, int n_in, double scale, int64_t *coeffs) {
    int Nh = p->slots;
    cplx *v = (cplx *)calloc(Nh, sizeof(cplx));
    for (int i = 0; i < n_in && i < Nh; i++) v[i].re = slots[i];
    fft_special_inv(p, v, Nh);
    for (int i = 0; i < Nh; i++) {
        coeffs[i] = (int64_t)llrint(v[i].re * scale);
        coeffs[i + Nh] = (int64_t)llrint(v[i].im * scale);
    }
    free(v);
}
static inline u64 i64_to_mod(int64_t x, u64 q) {
    if (x >= 0) return (u64)x % q;
    u64 r = ((u64)(-(x + 1)) + 1) % q;
    return r ? q - r : 0;
}
/* out: [nl][N] in evaluation form */
void hyo_encode(const hy_params *p, const double *slots, int n_in, double scale, int nl, u64 *out) {
    int N = p->N;
    int64_t *co = (int64_t *)malloc(sizeof(int64_t) * N);
    hyo_encode_coeffs(p, slots, n_in, scale, co);
    PARFOR
    for (int j = 0; j < nl; j++) {
        u64 *o = out + (size_t)j * N;
        for (int c = 0; c < N; c++) o[c] = i64_to_mod(co[c], p->q[j]);
        hyo_ntt_fwd(p, o, j);
    }
    free(co);
}
/* poly_coeff: [nl][N] COEFFICIENT form (limb j <-> q_j); only the first min(nl,2) limbs are needed because
 * |m + e| << q_0 q_1 / 2.  out: slots doubles (real parts). */
void hyo_decode(const hy_params *p, const u64 *poly, int nl, double scale, double *out) {
    int N = p->N, Nh = p->slots;
    double *co = (double *)malloc(sizeof(double) * N);
    if (nl >= 2) {
        u64 q0 = p->q[0], q1 = p->q[1];
        u64 q0inv = invmod(q0 % q1, q1);
        u128 Q = (u128)q0 * q1, half = Q >> 1;
        for (int c = 0; c < N; c++) {
            u64 r0 = poly[c], r1 = poly[(size_t)N + c];
            u64 d = submod(r1, r0 % q1, q1);
            u64 t = mulmod_slow(d, q0inv, q1);
            u128 x = (u128)r0 + (u128)q0 * t;
            int neg = x > half;
            u128 mag = neg ? Q - x : x;
            double v = (double)(u64)(mag >> 64) * 18446744073709551616.0 + (double)(u64)mag;
            co[c] = (neg ? -v : v) / scale;
        }
    } else {
        u64 q0 = p->q[0], half = q0 >> 1;
        for (int c = 0; c < N; c++) {
            u64 r0 = poly[c];
            int neg = r0 > half;
            double v = (double)(neg ? q0 - r0 : r0);
            co[c] = (neg ? -v : v) / scale;
        }
    }
    cplx *v = (cplx *)malloc(sizeof(cplx) * Nh);
    for (int i = 0; i < Nh; i++) {
        v[i].re = co[i];
        v[i].im = co[i + Nh];
    }
    fft_special(p, v, Nh);
    for (int i = 0; i < Nh; i++) out[i] = v[i].re;
    free(v);
    free(co);
}

/* ------------------------------------------------------------------ ciphertext helpers */
hy_ct *hyo_ct_alloc(const hy_params *p, int npoly, int nl, double scale) {
    hy_ct *c = (hy_ct *)malloc(sizeof(hy_ct));
    c->npoly = npoly;
    c->nl = nl;
    c->scale = scale;
    c->d = (u64 *)calloc((size_t)npoly * nl * p->N, sizeof(u64));
    return c;
}
hy_ct *hyo_ct_clone(const hy_params *p, const hy_ct *a) {
    hy_ct *c = hyo_ct_alloc(p, a->npoly, a->nl, a->scale);
    memcpy(c->d, a->d, sizeof(u64) * (size_t)a->npoly * a->nl * p->N);
    return c;
}
void hyo_ct_free(hy_ct *c) {
    if (!c) return;
    free(c->d);
    free(c);
}
u64 *hyo_ct_data(hy_ct *c) { return c->d; }
int hyo_ct_nl(const hy_ct *c) { return c->nl; }
int hyo_ct_npoly(const hy_ct *c) { return c->npoly; }
double hyo_ct_scale(const hy_ct *c) { return c->scale; }
#define CT(c, pidx, j) ((c)->d + ((size_t)(pidx) * (c)->nl + (j)) * (size_t)N)

/* small signed coefficients -> [nl limbs given by mods[]] evaluation form */
static void small_to_ntt(const hy_params *p, const int32_t *e, int m, u64 *out) {
    int N = p->N;
    u64 q = p->q[m];
    for (int c = 0; c < N; c++) out[c] = e[c] >= 0 ? (u64)e[c] : q - (u64)(-e[c]);
    hyo_ntt_fwd(p, out, m);
}

/* ------------------------------------------------------------------ key generation */
/* hybrid switching key FROM the secret s_from (Q limbs) TO the secret s_enc (Q u P limbs):
 * digit d: (b_d, a_d) = (-a_d s_enc + e_d + P [limb in digit d] s_from, a_d) */
static u64 *gen_evk(const hy_params *p, const uint8_t seed[32], const u64 *s_ntt, const u64 *s_from, int key_id) {
    int N = p->N, nT = p->nT;
    u64 *evk = (u64 *)malloc(sizeof(u64) * (size_t)p->dnum * 2 * nT * N);
    for (int d = 0; d < p->dnum; d++) {
        int32_t *e = (int32_t *)malloc(sizeof(int32_t) * N);
        hyo_sample_gauss(seed, HY_STREAM(HY_DOM_EVK_E, key_id, d, 0), e, N);
        PARFOR
        for (int m = 0; m < nT; m++) {
            u64 q = p->q[m];
            const barrett_t *bq = &p->bq[m];
            u64 *b = evk + (((size_t)d * 2 + 0) * nT + m) * N;
            u64 *a = evk + (((size_t)d * 2 + 1) * nT + m) * N;
            hyo_sample_uniform(seed, HY_STREAM(HY_DOM_EVK_A, key_id, d, m), q, a, N);
            small_to_ntt(p, e, m, b);
            const u64 *s = s_ntt + (size_t)m * N;
            int in_digit = (m < p->nQ) && (m / p->alpha == d);
            for (int c = 0; c < N; c++) {
                u64 v = submod(b[c], mulmod(a[c], s[c], bq), q);
                if (in_digit) v = addmod(v, mulmod(p->P_mod_q[m], s_from[(size_t)m * N + c], bq), q);
                b[c] = v;
            }
        }
        free(e);
    }
    return evk;
}

static u64 inv_mod_pow2(u64 g, u64 M) { /* g odd, M power of two */
    u64 x = 1;
    for (int i = 0; i < 6; i++) x = x * (2 - g * x);
    return x & (M - 1);
}

hy_keys *hyo_keygen(const hy_params *p, const uint8_t seed[32], const int *rot_idx, int n_rot) {
    int N = p->N, nT = p->nT, nQ = p->nQ;
    hy_keys *k = (hy_keys *)calloc(1, sizeof(hy_keys));
    k->s_coeff = (int8_t *)malloc(N);
    hyo_sample_ternary(seed, HY_STREAM(HY_DOM_SK, 0, 0, 0), k->s_coeff, N);
    k->s_ntt = (u64 *)malloc(sizeof(u64) * (size_t)nT * N);
    int32_t *s32 = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int c = 0; c < N; c++) s32[c] = k->s_coeff[c];
    PARFOR
    for (int m = 0; m < nT; m++) small_to_ntt(p, s32, m, k->s_ntt + (size_t)m * N);
    free(s32);
    /* public key (b, a) = (-a s + e, a) over Q */
    k->pk = (u64 *)malloc(sizeof(u64) * 2 * (size_t)nQ * N);
    int32_t *e = (int32_t *)malloc(sizeof(int32_t) * N);
    hyo_sample_gauss(seed, HY_STREAM(HY_DOM_PK_E, 0, 0, 0), e, N);
    PARFOR
    for (int j = 0; j < nQ; j++) {
        u64 q = p->q[j];
        u64 *b = k->pk + (size_t)j * N, *a = k->pk + ((size_t)nQ + j) * N;
        hyo_sample_uniform(seed, HY_STREAM(HY_DOM_PK_A, 0, 0, j), q, a, N);
        small_to_ntt(p, e, j, b);
        for (int c = 0; c < N; c++) b[c] = submod(b[c], mulmod(a[c], k->s_ntt[(size_t)j * N + c], &p->bq[j]), q);
    }
    free(e);
    /* relinearisation key: s^2 -> s */
    u64 *s2 = (u64 *)malloc(sizeof(u64) * (size_t)nQ * N);
    for (int j = 0; j < nQ; j++)
        for (int c = 0; c < N; c++) {
            u64 s = k->s_ntt[(size_t)j * N + c];
            s2[(size_t)j * N + c] = mulmod(s, s, &p->bq[j]);
        }
    k->relin = gen_evk(p, seed, k->s_ntt, s2, 0);
    /* rotation keys: key r switches s -> sigma_g^{-1}(s), g = 5^r (it is an encryption of P*s under
     * sigma_g^{-1}(s)), so that Rot_r(ct) = sigma_g( c0 + KS_0(c1), KS_1(c1) ) with ONE permutation at the end:
     * sigma_g(c0 + k0) + sigma_g(k1) s = sigma_g(c0 + k0 + k1 sigma_g^{-1}(s)) = sigma_g(c0 + c1 s). */
    k->n_rot = n_rot;
    k->rot_idx = (int *)malloc(sizeof(int) * (n_rot > 0 ? n_rot : 1));
    k->rot = (u64 **)calloc(n_rot > 0 ? n_rot : 1, sizeof(u64 *));
    u64 M = 2ull * N;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < n_rot; i++) {
        k->rot_idx[i] = rot_idx[i];
        u64 g = hyo_galois_elt(p, rot_idx[i]);
        u64 ginv = inv_mod_pow2(g, M);
        u64 *se = (u64 *)malloc(sizeof(u64) * (size_t)nT * N);
        for (int m = 0; m < nT; m++) hyo_automorph_eval(p, k->s_ntt + (size_t)m * N, se + (size_t)m * N, ginv);
        k->rot[i] = gen_evk(p, seed, se, k->s_ntt, rot_idx[i]);
        free(se);
    }
    free(s2);
    return k;
}
void hyo_keys_free(hy_keys *k) {
    if (!k) return;
    free(k->s_coeff);
    free(k->s_ntt);
    free(k->pk);
    free(k->relin);
    for (int i = 0; i < k->n_rot; i++) free(k->rot[i]);
    free(k->rot);
    free(k->rot_idx);
    free(k);
}
const u64 *hyo_keys_rot(const hy_keys *k, int rot) {
    for (int i = 0; i < k->n_rot; i++)
        if (k->rot_idx[i] == rot) return k->rot[i];
    return NULL;
}

/* ------------------------------------------------------------------ encrypt / decrypt */
hy_ct *hyo_encrypt(const hy_params *p, const hy_keys *k, const double *slots, int n_in, const uint8_t seed[32],
                   u64 nonce) {
    int N = p->N, nQ = p->nQ;
    hy_ct *ct = hyo_ct_alloc(p, 2, nQ, p->delta);
    int64_t *m = (int64_t *)malloc(sizeof(int64_t) * N);
    hyo_encode_coeffs(p, slots, n_in, p->delta, m);
    int8_t *u8 = (int8_t *)malloc(N);
    int32_t *u = (int32_t *)malloc(sizeof(int32_t) * N), *e0 = (int32_t *)malloc(sizeof(int32_t) * N),
            *e1 = (int32_t *)malloc(sizeof(int32_t) * N);
    hyo_sample_ternary(seed, HY_STREAM(HY_DOM_ENC_U, nonce, 0, 0), u8, N);
    for (int c = 0; c < N; c++) u[c] = u8[c];
    hyo_sample_gauss(seed, HY_STREAM(HY_DOM_ENC_E0, nonce, 0, 0), e0, N);
    hyo_sample_gauss(seed, HY_STREAM(HY_DOM_ENC_E1, nonce, 0, 0), e1, N);
    PARFOR
    for (int j = 0; j < nQ; j++) {
        u64 q = p->q[j];
        const barrett_t *bq = &p->bq[j];
        u64 *c0 = CT(ct, 0, j), *c1 = CT(ct, 1, j);
        u64 *un = (u64 *)malloc(sizeof(u64) * N), *t = (u64 *)malloc(sizeof(u64) * N);
        small_to_ntt(p, u, j, un);
        /* c0 = b u + NTT(e0 + m) ; c1 = a u + NTT(e1) */
        for (int c = 0; c < N; c++) t[c] = addmod(i64_to_mod(m[c], q), i64_to_mod(e0[c], q), q);
        hyo_ntt_fwd(p, t, j);
        const u64 *b = k->pk + (size_t)j * N, *a = k->pk + ((size_t)nQ + j) * N;
        for (int c = 0; c < N; c++) c0[c] = addmod(mulmod(b[c], un[c], bq), t[c], q);
        small_to_ntt(p, e1, j, t);
        for (int c = 0; c < N; c++) c1[c] = addmod(mulmod(a[c], un[c], bq), t[c], q);
        free(un);
        free(t);
    }
    free(m);
    free(u8);
    free(u);
    free(e0);
    free(e1);
    return ct;
}
void hyo_decrypt(const hy_params *p, const hy_keys *k, const hy_ct *ct, double *out) {
    int N = p->N;
    int nl = ct->nl < 2 ? ct->nl : 2;
    u64 *t = (u64 *)malloc(sizeof(u64) * (size_t)nl * N);
    for (int j = 0; j < nl; j++) {
        u64 q = p->q[j];
        const barrett_t *bq = &p->bq[j];
        const u64 *s = k->s_ntt + (size_t)j * N;
        u64 *o = t + (size_t)j * N;
        for (int c = 0; c < N; c++) {
            u64 acc = CT(ct, ct->npoly - 1, j)[c];
            for (int pi = ct->npoly - 2; pi >= 0; pi--) acc = addmod(mulmod(acc, s[c], bq), CT(ct, pi, j)[c], q);
            o[c] = acc;
        }
        hyo_ntt_inv(p, o, j);
    }
    hyo_decode(p, t, nl, ct->scale, out);
    free(t);
}

/* ------------------------------------------------------------------ hybrid key switching */
static inline int ext_mod(const hy_params *p, int nl, int t) { return t < nl ? t : p->nQ + (t - nl); }

/* ModUp of every digit of c (evaluation form, nl limbs) to Q_l u P: returns [nd][nl+nP][N], evaluation form */
u64 *hyo_hoist_precompute(const hy_params *p, const u64 *c, int nl) {
    int N = p->N, nE = nl + p->nP, nd = (nl + p->alpha - 1) / p->alpha;
    u64 *dig = (u64 *)malloc(sizeof(u64) * (size_t)nd * nE * N);
    for (int d = 0; d < nd; d++) {
        int lo = d * p->alpha, hi = lo + p->alpha < nl ? lo + p->alpha : nl, sz = hi - lo;
        /* y_j = INTT(c_j) * (D/q_j)^{-1} mod q_j */
        u64 *y = (u64 *)malloc(sizeof(u64) * (size_t)sz * N);
        PARFOR
        for (int jj = 0; jj < sz; jj++) {
            int j = lo + jj;
            u64 q = p->q[j], prod = 1;
            for (int i = lo; i < hi; i++)
                if (i != j) prod = mulmod_slow(prod, p->q[i] % q, q);
            u64 inv = invmod(prod, q);
            u64 *yy = y + (size_t)jj * N;
            memcpy(yy, c + (size_t)j * N, sizeof(u64) * N);
            hyo_ntt_inv(p, yy, j);
            for (int cidx = 0; cidx < N; cidx++) yy[cidx] = mulmod(yy[cidx], inv, &p->bq[j]);
        }
        PARFOR
        for (int t = 0; t < nE; t++) {
            int m = ext_mod(p, nl, t);
            u64 *o = dig + ((size_t)d * nE + t) * N;
            if (m >= lo && m < hi) {
                memcpy(o, c + (size_t)m * N, sizeof(u64) * N);
                continue;
            }
            u64 qt = p->q[m];
            u64 f[HY_MAX_LIMBS];
            for (int jj = 0; jj < sz; jj++) {
                u64 prod = 1;
                for (int i = lo; i < hi; i++)
                    if (i != lo + jj) prod = mulmod_slow(prod, p->q[i] % qt, qt);
                f[jj] = prod;
            }
            for (int cidx = 0; cidx < N; cidx++) {
                u64 acc = 0;
                for (int jj = 0; jj < sz; jj++)
                    acc = addmod(acc, mulmod(y[(size_t)jj * N + cidx] % qt, f[jj], &p->bq[m]), qt);
                o[cidx] = acc;
            }
            hyo_ntt_fwd(p, o, m);
        }
        free(y);
    }
    return dig;
}

/* acc [nl+nP][N] (evaluation form over Q_l u P)  ->  out [nl][N] = round-ish(acc / P) */
static void mod_down(const hy_params *p, u64 *acc, int nl, u64 *out) {
    int N = p->N, nP = p->nP;
    u64 *y = (u64 *)malloc(sizeof(u64) * (size_t)nP * N);
    PARFOR
    for (int k = 0; k < nP; k++) {
        int m = p->nQ + k;
        u64 *yy = y + (size_t)k * N;
        memcpy(yy, acc + (size_t)(nl + k) * N, sizeof(u64) * N);
        hyo_ntt_inv(p, yy, m);
        for (int c = 0; c < N; c++) yy[c] = mulmod(yy[c], p->Phat_inv[k], &p->bq[m]);
    }
    PARFOR
    for (int j = 0; j < nl; j++) {
        u64 q = p->q[j];
        const barrett_t *bq = &p->bq[j];
        u64 *o = out + (size_t)j * N;
        for (int c = 0; c < N; c++) {
            u64 a = 0;
            for (int k = 0; k < nP; k++) a = addmod(a, mulmod(y[(size_t)k * N + c] % q, p->Phat_mod_q[k][j], bq), q);
            o[c] = a;
        }
        hyo_ntt_fwd(p, o, j);
        const u64 *aj = acc + (size_t)j * N;
        for (int c = 0; c < N; c++) o[c] = mulmod(submod(aj[c], o[c], q), p->Pinv_mod_q[j], bq);
    }
    free(y);
}

/* <digits, evk> over Q_l u P, then ModDown.  evk: [dnum][2][nT][N] */
static void ks_inner_moddown(const hy_params *p, const u64 *dig, int nl, const u64 *evk, u64 *out0, u64 *out1) {
    int N = p->N, nE = nl + p->nP, nd = (nl + p->alpha - 1) / p->alpha, nT = p->nT;
    u64 *acc0 = (u64 *)malloc(sizeof(u64) * (size_t)nE * N), *acc1 = (u64 *)malloc(sizeof(u64) * (size_t)nE * N);
    PARFOR
    for (int t = 0; t < nE; t++) {
        int m = ext_mod(p, nl, t);
        u64 q = p->q[m];
        const barrett_t *bq = &p->bq[m];
        u64 *a0 = acc0 + (size_t)t * N, *a1 = acc1 + (size_t)t * N;
        for (int c = 0; c < N; c++) a0[c] = a1[c] = 0;
        for (int d = 0; d < nd; d++) {
            const u64 *x = dig + ((size_t)d * nE + t) * N;
            const u64 *kb = evk + (((size_t)d * 2 + 0) * nT + m) * N;
            const u64 *ka = evk + (((size_t)d * 2 + 1) * nT + m) * N;
            for (int c = 0; c < N; c++) {
                a0[c] = addmod(a0[c], mulmod(x[c], kb[c], bq), q);
                a1[c] = addmod(a1[c], mulmod(x[c], ka[c], bq), q);
            }
        }
    }
    mod_down(p, acc0, nl, out0);
    mod_down(p, acc1, nl, out1);
    free(acc0);
    free(acc1);
}

void hyo_keyswitch(const hy_params *p, const u64 *c, int nl, const u64 *evk, u64 *out0, u64 *out1) {
    u64 *dig = hyo_hoist_precompute(p, c, nl);
    ks_inner_moddown(p, dig, nl, evk, out0, out1);
    free(dig);
}

/* EvalFastRotation (sender_diag.cpp:25): digits were made once from c1 by hyo_hoist_precompute */
hy_ct *hyo_rotate_hoisted(const hy_params *p, const hy_ct *ct, const u64 *digits, const u64 *evk, int rot) {
    int N = p->N, nl = ct->nl;
    u64 g = hyo_galois_elt(p, rot);
    u64 *k0 = (u64 *)malloc(sizeof(u64) * (size_t)nl * N), *k1 = (u64 *)malloc(sizeof(u64) * (size_t)nl * N);
    ks_inner_moddown(p, digits, nl, evk, k0, k1);
    hy_ct *out = hyo_ct_alloc(p, 2, nl, ct->scale);
    PARFOR
    for (int j = 0; j < nl; j++) {
        u64 q = p->q[j];
        u64 *t = k0 + (size_t)j * N;
        const u64 *c0 = CT(ct, 0, j);
        for (int c = 0; c < N; c++) t[c] = addmod(t[c], c0[c], q);
        hyo_automorph_eval(p, t, CT(out, 0, j), g);
        hyo_automorph_eval(p, k1 + (size_t)j * N, CT(out, 1, j), g);
    }
    free(k0);
    free(k1);
    return out;
}
hy_ct *hyo_rotate(const hy_params *p, const hy_keys *k, const hy_ct *ct, int rot) {
    const u64 *evk = hyo_keys_rot(k, rot);
    if (!evk) {
        fprintf(stderr, "hydia oracle: no rotation key for %d\n", rot);
        return NULL;
    }
    int N = p->N;
    u64 *dig = hyo_hoist_precompute(p, CT(ct, 1, 0), ct->nl);
    hy_ct *o = hyo_rotate_hoisted(p, ct, dig, evk, rot);
    free(dig);
    return o;
}

/* ------------------------------------------------------------------ arithmetic */
void hyo_drop_to(const hy_params *p, hy_ct *a, int nl) {
    int N = p->N;
    if (nl >= a->nl) return;
    u64 *nd = (u64 *)malloc(sizeof(u64) * (size_t)a->npoly * nl * N);
    for (int pi = 0; pi < a->npoly; pi++)
        memcpy(nd + (size_t)pi * nl * N, a->d + (size_t)pi * a->nl * N, sizeof(u64) * (size_t)nl * N);
    free(a->d);
    a->d = nd;
    a->nl = nl;
}
/* EvalMultNoRelin (sender_diag.cpp:93): (a0 b0, a0 b1 + a1 b0, a1 b1) */
hy_ct *hyo_mult_norelin(const hy_params *p, const hy_ct *a, const hy_ct *b) {
    int N = p->N, nl = a->nl;
    if (b->nl != nl || a->npoly != 2 || b->npoly != 2) {
        fprintf(stderr, "hydia oracle: mult_norelin level/shape mismatch\n");
        return NULL;
    }
    hy_ct *o = hyo_ct_alloc(p, 3, nl, a->scale * b->scale);
    PARFOR
    for (int j = 0; j < nl; j++) {
        u64 q = p->q[j];
        const barrett_t *bq = &p->bq[j];
        const u64 *a0 = CT(a, 0, j), *a1 = CT(a, 1, j), *b0 = CT(b, 0, j), *b1 = CT(b, 1, j);
        u64 *d0 = CT(o, 0, j), *d1 = CT(o, 1, j), *d2 = CT(o, 2, j);
        for (int c = 0; c < N; c++) {
            d0[c] = mulmod(a0[c], b0[c], bq);
            d1[c] = addmod(mulmod(a0[c], b1[c], bq), mulmod(a1[c], b0[c], bq), q);
            d2[c] = mulmod(a1[c], b1[c], bq);
        }
    }
    return o;
}
void hyo_add_inplace(const hy_params *p, hy_ct *a, const hy_ct *b) {
    int N = p->N;
    if (a->nl != b->nl || a->npoly != b->npoly) {
        fprintf(stderr, "hydia oracle: add level/shape mismatch (%d,%d) vs (%d,%d)\n", a->npoly, a->nl, b->npoly, b->nl);
        return;
    }
    PARFOR
    for (int t = 0; t < a->npoly * a->nl; t++) {
        u64 q = p->q[t % a->nl];
        u64 *x = a->d + (size_t)t * N;
        const u64 *y = b->d + (size_t)t * N;
        for (int c = 0; c < N; c++) x[c] = addmod(x[c], y[c], q);
    }
}
void hyo_sub_inplace(const hy_params *p, hy_ct *a, const hy_ct *b) {
    int N = p->N;
    if (a->nl != b->nl || a->npoly != b->npoly) {
        fprintf(stderr, "hydia oracle: sub level/shape mismatch\n");
        return;
    }
    PARFOR
    for (int t = 0; t < a->npoly * a->nl; t++) {
        u64 q = p->q[t % a->nl];
        u64 *x = a->d + (size_t)t * N;
        const u64 *y = b->d + (size_t)t * N;
        for (int c = 0; c < N; c++) x[c] = submod(x[c], y[c], q);
    }
}
/* RelinearizeInPlace (sender_diag.cpp:79) */
void hyo_relin_inplace(const hy_params *p, const hy_keys *k, hy_ct *a) {
    int N = p->N, nl = a->nl;
    if (a->npoly != 3) return;
    u64 *k0 = (u64 *)malloc(sizeof(u64) * (size_t)nl * N), *k1 = (u64 *)malloc(sizeof(u64) * (size_t)nl * N);
    hyo_keyswitch(p, CT(a, 2, 0), nl, k->relin, k0, k1);
    u64 *nd = (u64 *)malloc(sizeof(u64) * 2 * (size_t)nl * N);
    PARFOR
    for (int j = 0; j < nl; j++) {
        u64 q = p->q[j];
        for (int c = 0; c < N; c++) {
            nd[(size_t)j * N + c] = addmod(CT(a, 0, j)[c], k0[(size_t)j * N + c], q);
            nd[((size_t)nl + j) * N + c] = addmod(CT(a, 1, j)[c], k1[(size_t)j * N + c], q);
        }
    }
    free(a->d);
    free(k0);
    free(k1);
    a->d = nd;
    a->npoly = 2;
}
/* RescaleInPlace (sender_diag.cpp:80): drop q_l with rounding to nearest */
void hyo_rescale_inplace(const hy_params *p, hy_ct *a) {
    int N = p->N, nl = a->nl, l = nl - 1;
    if (nl < 2) {
        fprintf(stderr, "hydia oracle: rescale with one limb left\n");
        return;
    }
    u64 ql = p->q[l], half = ql >> 1;
    u64 *nd = (u64 *)malloc(sizeof(u64) * (size_t)a->npoly * l * N);
    for (int pi = 0; pi < a->npoly; pi++) {
        u64 *t = (u64 *)malloc(sizeof(u64) * N);
        memcpy(t, CT(a, pi, l), sizeof(u64) * N);
        hyo_ntt_inv(p, t, l);
        PARFOR
        for (int j = 0; j < l; j++) {
            u64 q = p->q[j];
            const barrett_t *bq = &p->bq[j];
            u64 *o = nd + ((size_t)pi * l + j) * N;
            for (int c = 0; c < N; c++) {
                u64 v = t[c];
                o[c] = v > half ? negmod((ql - v) % q, q) : v % q;
            }
            hyo_ntt_fwd(p, o, j);
            const u64 *x = CT(a, pi, j);
            for (int c = 0; c < N; c++) o[c] = mulmod(submod(x[c], o[c], q), p->ql_inv[l][j], bq);
        }
        free(t);
    }
    free(a->d);
    a->d = nd;
    a->nl = l;
    a->scale /= (double)ql;
}

/* residue of the real number v (already multiplied by its scale) modulo q; v is rounded to the nearest
 * integer (ties to even) when |v| < 2^63 and is an exact integer above that */
static u64 double_to_mod(double v, u64 q) {
    int neg = v < 0;
    double a = fabs(v);
    u64 r;
    if (a < 9223372036854775808.0) {
        r = (u64)llrint(a) % q;
    } else {
        int e;
        double m = frexp(a, &e); /* a = m * 2^e, m in [0.5,1) */
        u64 mant = (u64)ldexp(m, 53);
        r = mulmod_slow(mant % q, powmod(2, (u64)(e - 53), q), q);
    }
    return neg ? negmod(r, q) : r;
}
/* EvalAddInPlace(ct, double) (openFHE_wrapper.cpp:182): constant polynomial = the same residue in every
 * evaluation slot */
void hyo_add_const(const hy_params *p, hy_ct *a, double c) {
    int N = p->N;
    for (int j = 0; j < a->nl; j++) {
        u64 q = p->q[j], r = double_to_mod(c * a->scale, q);
        u64 *x = CT(a, 0, j);
        for (int i = 0; i < N; i++) x[i] = addmod(x[i], r, q);
    }
}
hy_ct *hyo_mul_const(const hy_params *p, const hy_ct *a, double c, double const_scale) {
    int N = p->N;
    hy_ct *o = hyo_ct_alloc(p, a->npoly, a->nl, a->scale * const_scale);
    PARFOR
    for (int t = 0; t < a->npoly * a->nl; t++) {
        int j = t % a->nl;
        u64 r = double_to_mod(c * const_scale, p->q[j]);
        const u64 *x = a->d + (size_t)t * N;
        u64 *y = o->d + (size_t)t * N;
        for (int i = 0; i < N; i++) y[i] = mulmod(x[i], r, &p->bq[j]);
    }
    return o;
}
/* a -= K * b for an integer K (reduced per limb): aligns b's scale to a's before a subtraction (comparator steps) */
void hyo_submul_int(const hy_params *p, hy_ct *a, const hy_ct *b, u64 K) {
    int N = p->N;
    if (a->nl != b->nl || a->npoly != b->npoly) {
        fprintf(stderr, "hydia oracle: submul level/shape mismatch\n");
        return;
    }
    PARFOR
    for (int t = 0; t < a->npoly * a->nl; t++) {
        int j = t % a->nl;
        u64 q = p->q[j], r = K % q;
        u64 *x = a->d + (size_t)t * N;
        const u64 *y = b->d + (size_t)t * N;
        for (int c = 0; c < N; c++) x[c] = submod(x[c], mulmod(y[c], r, &p->bq[j]), q);
    }
}
/* ct x ct with level alignment, relinearisation and rescale */
hy_ct *hyo_mult(const hy_params *p, const hy_keys *k, const hy_ct *a, const hy_ct *b) {
    int nl = a->nl < b->nl ? a->nl : b->nl;
    hy_ct *x = hyo_ct_clone(p, a), *y = hyo_ct_clone(p, b);
    hyo_drop_to(p, x, nl);
    hyo_drop_to(p, y, nl);
    hy_ct *o = hyo_mult_norelin(p, x, y);
    hyo_ct_free(x);
    hyo_ct_free(y);
    hyo_relin_inplace(p, k, o);
    hyo_rescale_inplace(p, o);
    return o;
}

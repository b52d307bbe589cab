/*
 * oracle/path.c — TEST INFRASTRUCTURE ONLY (see hydia_oracle.h).
 *
 * The HyDia (approach 5) roles and the comparator, following the reference's orchestration:
 *   DiagonalEnroller   /root/reference/src/enroller/enroller_diag.cpp:12-166
 *   DiagonalReceiver   /root/reference/src/receiver/receiver_diag.cpp:13-26
 *   HersReceiver::decrypt{Membership,Index}  /root/reference/src/receiver/receiver_hers.cpp:26-54
 *   DiagonalSender     /root/reference/src/sender/sender_diag.cpp:12-94
 *   OpenFHEWrapper::chebyshevCompare  /root/reference/src/openFHE_wrapper.cpp:143-185
 *   VectorUtils::plaintextNormalize   /root/reference/src/vector_utils.cpp:32-51
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hydia_oracle.h"

/* include/config.h:9,14,30 */
#define MATCH_THRESHOLD 0.44
#define COMP_DEPTH 10

/* ------------------------------------------------------------------ comparator: plaintext side */
/* EvalChebyshevCoefficients as called from EvalChebyshevFunction (openFHE_wrapper.cpp:173-174):
 * interpolation of f(x) = (x >= delta ? 1 : -1) at the degree+1 Chebyshev nodes of [-1,1]. The series is
 * sum_j c_j T_j(x) with c_0 already halved. */
void hyo_chebyshev_step_coeffs(double delta, int degree, double *coeffs) {
    int n = degree + 1;
    double *f = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) {
        double x = cos(M_PI * (i + 0.5) / n);
        f[i] = (x >= delta) ? 1.0 : -1.0;
    }
    for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += f[i] * cos(M_PI * j * (i + 0.5) / n);
        coeffs[j] = s * 2.0 / n;
    }
    coeffs[0] *= 0.5;
    free(f);
}
/* F4_COEFS, openFHE_wrapper.cpp:158-169 (Cheon et al. 2019/1234) */
static const double F4[10] = {0.0, 315.0 / 128.0, 0.0, -420.0 / 128.0, 0.0, 378.0 / 128.0, 0.0, -180.0 / 128.0, 0.0, 35.0 / 128.0};

/* the function chebyshevCompare approximates, evaluated in plain doubles (Clenshaw + Horner) */
double hyo_compare_plain(double x, double delta, int degree) {
    double *c = (double *)malloc(sizeof(double) * (degree + 1));
    hyo_chebyshev_step_coeffs(delta, degree, c);
    double b1 = 0, b2 = 0;
    for (int j = degree; j >= 1; j--) {
        double t = 2 * x * b1 - b2 + c[j];
        b2 = b1;
        b1 = t;
    }
    double y = x * b1 - b2 + c[0];
    free(c);
    double r = 0;
    for (int i = 9; i >= 0; i--) r = r * y + F4[i];
    return r + 1.0;
}

/* ------------------------------------------------------------------ comparator: encrypted side */
typedef struct {
    const hy_params *p;
    const hy_keys *k;
    hy_ct *T[9];   /* T[1..8] */
    hy_ct *G[16];  /* G[i] = T_{8*2^i}, G[0] aliases T[8] */
    int ng;
} cheb_ctx;

/* 2*a*b - c (c may be NULL meaning the constant 1): one ct x ct product, doubled before the rescale.  Scale management:
 * the product sits at scale s_a*s_b, c at s_c; c is brought to the product's scale by the INTEGER factor
 * K = round(s_a*s_b/s_c) (about 2^45, relative rounding 1e-14) and subtracted BEFORE the rescale, so both operands of the
 * subtraction carry the same scale exactly and no drift enters the Chebyshev recurrence. */
static hy_ct *cheb_step(const hy_params *p, const hy_keys *k, const hy_ct *a, const hy_ct *b, const hy_ct *c) {
    int nl = a->nl < b->nl ? a->nl : b->nl;
    hy_ct *x = hyo_ct_clone(p, a), *y = hyo_ct_clone(p, b);
    hyo_drop_to(p, x, nl);
    hyo_drop_to(p, y, nl);
    hy_ct *o = hyo_mult_norelin(p, x, y);
    hyo_ct_free(x);
    hyo_ct_free(y);
    hyo_relin_inplace(p, k, o);
    hyo_add_inplace(p, o, o); /* x2 */
    if (c) {
        hy_ct *cc = hyo_ct_clone(p, c);
        hyo_drop_to(p, cc, o->nl);
        hyo_submul_int(p, o, cc, (u64)llround(o->scale / cc->scale));
        hyo_ct_free(cc);
        hyo_rescale_inplace(p, o);
    } else {
        hyo_rescale_inplace(p, o);
        hyo_add_const(p, o, -1.0);
    }
    return o;
}

/* limbs in use by the terms of a leaf (before its rescale) */
static int leaf_nl(cheb_ctx *cx, const double *c, int deg) {
    int nl = cx->T[1]->nl;
    for (int j = 1; j <= deg; j++)
        if (c[j] != 0.0 && cx->T[j]->nl < nl) nl = cx->T[j]->nl;
    return nl;
}
/* sum_{j<=deg} c_j T_j with deg <= 7: constants encoded at target*q_l/scale(T_j) so that the rescaled result has scale
 * exactly `target` whatever the scales of the T_j */
static hy_ct *cheb_leaf(cheb_ctx *cx, const double *c, int deg, double target) {
    const hy_params *p = cx->p;
    int nl = leaf_nl(cx, c, deg);
    int any = 0;
    for (int j = 1; j <= deg; j++)
        if (c[j] != 0.0) any = 1;
    double S = target * (double)p->q[nl - 1];
    hy_ct *acc = NULL;
    int last = any ? deg : 1; /* a pure constant is encoded as 0*T_1 + c_0 */
    for (int j = 1; j <= last; j++) {
        double cj = any ? c[j] : 0.0;
        if (cj == 0.0 && any) continue;
        hy_ct *t = hyo_ct_clone(p, cx->T[j]);
        hyo_drop_to(p, t, nl);
        hy_ct *m = hyo_mul_const(p, t, cj, S / t->scale);
        m->scale = S;
        hyo_ct_free(t);
        if (!acc)
            acc = m;
        else {
            hyo_add_inplace(p, acc, m);
            hyo_ct_free(m);
        }
    }
    hyo_add_const(p, acc, c[0]);
    hyo_rescale_inplace(p, acc);
    acc->scale = target;
    return acc;
}

static void cheb_split(const double *c, int deg, int g, double *qc, double *rc) {
    /* c = q * T_g + r using T_j = 2 T_{j-g} T_g - T_{2g-j}  (g < j < 2g) */
    for (int j = 0; j < g; j++) {
        qc[j] = 0.0;
        rc[j] = c[j];
    }
    qc[0] = c[g];
    for (int j = g + 1; j <= deg; j++) {
        qc[j - g] = 2.0 * c[j];
        rc[2 * g - j] -= c[j];
    }
}
/* limbs the result of cheb_node will have (dry run of the recursion below) */
static int cheb_node_nl(cheb_ctx *cx, const double *c, int deg, int gi) {
    while (deg > 0 && c[deg] == 0.0) deg--;
    if (deg < 8) return leaf_nl(cx, c, deg) - 1;
    int g = 8 << gi;
    if (deg < g) return cheb_node_nl(cx, c, deg, gi - 1);
    double *qc = (double *)calloc(g, sizeof(double)), *rc = (double *)calloc(g, sizeof(double));
    cheb_split(c, deg, g, qc, rc);
    int nq = cheb_node_nl(cx, qc, deg - g, gi - 1), nr = cheb_node_nl(cx, rc, g - 1, gi - 1);
    free(qc);
    free(rc);
    int np = (nq < cx->G[gi]->nl ? nq : cx->G[gi]->nl) - 1;
    return np < nr ? np : nr;
}
/* evaluates sum_{j<=deg} c_j T_j at scale `target`, deg < 2*g where g = 8*2^gi (or deg < 8 when gi < 0).  The target
 * scale is pushed DOWN the recursion (quotient: target*q_l/scale(T_g); remainder: the product's scale), so every
 * addition in the tree joins operands of identical scale. */
static hy_ct *cheb_node(cheb_ctx *cx, const double *c, int deg, int gi, double target) {
    const hy_params *p = cx->p;
    while (deg > 0 && c[deg] == 0.0) deg--;
    if (deg < 8) return cheb_leaf(cx, c, deg, target);
    int g = 8 << gi;
    if (deg < g) return cheb_node(cx, c, deg, gi - 1, target);
    double *qc = (double *)calloc(g, sizeof(double)), *rc = (double *)calloc(g, sizeof(double));
    cheb_split(c, deg, g, qc, rc);
    int nq = cheb_node_nl(cx, qc, deg - g, gi - 1);
    int lp = nq < cx->G[gi]->nl ? nq : cx->G[gi]->nl; /* limbs of the product before its rescale */
    hy_ct *Q = cheb_node(cx, qc, deg - g, gi - 1, target * (double)p->q[lp - 1] / cx->G[gi]->scale);
    hy_ct *prod = hyo_mult(p, cx->k, Q, cx->G[gi]);
    hy_ct *R = cheb_node(cx, rc, g - 1, gi - 1, prod->scale);
    int nl = prod->nl < R->nl ? prod->nl : R->nl;
    hyo_drop_to(p, prod, nl);
    hyo_drop_to(p, R, nl);
    hyo_add_inplace(p, prod, R);
    hyo_ct_free(Q);
    hyo_ct_free(R);
    free(qc);
    free(rc);
    return prod;
}

/* EvalChebyshevFunction's evaluation half (openFHE_wrapper.cpp:174) on [-1,1]: a baby-step/giant-step
 * (Paterson-Stockmeyer) evaluation in the Chebyshev basis with babies T_1..T_8 and giants T_16, T_32, ...
 * Depth ceil(log2(degree+1)) (6 for degree 59). */
hy_ct *hyo_eval_chebyshev63(const hy_params *p, const hy_keys *k, const hy_ct *x, const double *coeffs, int degree) {
    cheb_ctx cx;
    memset(&cx, 0, sizeof(cx));
    cx.p = p;
    cx.k = k;
    cx.T[1] = hyo_ct_clone(p, x);
    int top = degree < 8 ? degree : 8;
    if (top >= 2) cx.T[2] = cheb_step(p, k, cx.T[1], cx.T[1], NULL);
    if (top >= 3) cx.T[3] = cheb_step(p, k, cx.T[2], cx.T[1], cx.T[1]);
    if (top >= 4) cx.T[4] = cheb_step(p, k, cx.T[2], cx.T[2], NULL);
    if (top >= 5) cx.T[5] = cheb_step(p, k, cx.T[3], cx.T[2], cx.T[1]);
    if (top >= 6) cx.T[6] = cheb_step(p, k, cx.T[3], cx.T[3], NULL);
    if (top >= 7) cx.T[7] = cheb_step(p, k, cx.T[4], cx.T[3], cx.T[1]);
    if (top >= 8) cx.T[8] = cheb_step(p, k, cx.T[4], cx.T[4], NULL);
    int gi = -1;
    if (degree >= 8) {
        cx.G[0] = cx.T[8];
        gi = 0;
        while ((8 << (gi + 1)) <= degree) {
            cx.G[gi + 1] = cheb_step(p, k, cx.G[gi], cx.G[gi], NULL);
            gi++;
        }
    }
    cx.ng = gi + 1;
    hy_ct *r = cheb_node(&cx, coeffs, degree, gi, p->delta);
    for (int j = 1; j <= 8; j++) hyo_ct_free(cx.T[j]);
    for (int i = 1; i < cx.ng; i++) hyo_ct_free(cx.G[i]);
    return r;
}

/* EvalPoly(ct, F4_COEFS) (openFHE_wrapper.cpp:179) in depth 4:
 * f4(y) = (c1 y + c3 y^3) + y^4 (c5 y + c7 y^3) + (c9 y) y^8 */
hy_ct *hyo_eval_f4(const hy_params *p, const hy_keys *k, const hy_ct *y) {
    hy_ct *y2 = hyo_mult(p, k, y, y);
    hy_ct *y3 = hyo_mult(p, k, y2, y);
    hy_ct *y4 = hyo_mult(p, k, y2, y2);
    hy_ct *y8 = hyo_mult(p, k, y4, y4);
    int nl = y3->nl;
    hy_ct *yd = hyo_ct_clone(p, y);
    hyo_drop_to(p, yd, nl);
    /* v = c5 y + c7 y^3 at scale Delta; a = v y^4 fixes the scale every other summand is steered to */
    double S = p->delta * (double)p->q[nl - 1];
    hy_ct *v = hyo_mul_const(p, yd, F4[5], S / yd->scale);
    hy_ct *t = hyo_mul_const(p, y3, F4[7], S / y3->scale);
    v->scale = t->scale = S;
    hyo_add_inplace(p, v, t);
    hyo_ct_free(t);
    hyo_rescale_inplace(p, v);
    v->scale = p->delta;
    hy_ct *a = hyo_mult(p, k, v, y4);
    /* u = c1 y + c3 y^3 at a's scale */
    double Su = a->scale * (double)p->q[nl - 1];
    hy_ct *u = hyo_mul_const(p, yd, F4[1], Su / yd->scale);
    t = hyo_mul_const(p, y3, F4[3], Su / y3->scale);
    u->scale = t->scale = Su;
    hyo_add_inplace(p, u, t);
    hyo_ct_free(t);
    hyo_rescale_inplace(p, u);
    u->scale = a->scale;
    hyo_ct_free(yd);
    /* w = c9 y at the scale that makes b = w y^8 come out at a's scale */
    int lb = (y->nl - 1) < y8->nl ? (y->nl - 1) : y8->nl;
    double wt = a->scale * (double)p->q[lb - 1] / y8->scale;
    double S0 = wt * (double)p->q[y->nl - 1];
    hy_ct *w = hyo_mul_const(p, y, F4[9], S0 / y->scale);
    w->scale = S0;
    hyo_rescale_inplace(p, w);
    w->scale = wt;
    hy_ct *b = hyo_mult(p, k, w, y8);
    int fl = a->nl < b->nl ? a->nl : b->nl;
    if (u->nl < fl) fl = u->nl;
    hyo_drop_to(p, a, fl);
    hyo_drop_to(p, b, fl);
    hyo_drop_to(p, u, fl);
    hyo_add_inplace(p, a, b);
    hyo_add_inplace(p, a, u);
    hyo_ct_free(b);
    hyo_ct_free(u);
    hyo_ct_free(v);
    hyo_ct_free(w);
    hyo_ct_free(y2);
    hyo_ct_free(y3);
    hyo_ct_free(y4);
    hyo_ct_free(y8);
    return a;
}

/* OpenFHEWrapper::chebyshevCompare, openFHE_wrapper.cpp:143-185 */
hy_ct *hyo_chebyshev_compare(const hy_params *p, const hy_keys *k, const hy_ct *x, double delta, int sign_depth) {
    if (sign_depth < 7 || sign_depth > 15) { /* :146-149 — message, return the input unchanged */
        fprintf(stderr, "Error: chebshevCompare requires a depth parameter between 7 and 15\n");
        return hyo_ct_clone(p, x);
    }
    static const int DEPTH_TO_DEGREE[12] = {-1, -1, -1, 5, 13, 27, 59, 119, 247, 495, 1007, 2031}; /* :153-155 */
    int degree = DEPTH_TO_DEGREE[sign_depth - 4];
    double *c = (double *)malloc(sizeof(double) * (degree + 1));
    hyo_chebyshev_step_coeffs(delta, degree, c);
    hy_ct *y = hyo_eval_chebyshev63(p, k, x, c, degree);
    free(c);
    hy_ct *r = hyo_eval_f4(p, k, y);
    hyo_ct_free(y);
    hyo_add_const(p, r, 1.0); /* :182 */
    return r;
}

/* ------------------------------------------------------------------ enroller */
/* VectorUtils::plaintextNormalize, vector_utils.cpp:42-51 (zero vector passes through) */
void hyo_normalize(double *x, int dim) {
    double m = 0.0;
    for (int i = 0; i < dim; i++) m += x[i] * x[i];
    m = sqrt(m);
    if (m != 0)
        for (int i = 0; i < dim; i++) x[i] = x[i] / m;
}
/* concatenateRows' output count, enroller_diag.cpp:120-122 */
size_t hyo_enroll_num_cts(const hy_params *p, size_t n) {
    size_t dim = p->dim, per = p->slots / dim;
    size_t nblk = (n + dim - 1) / dim;
    return ((nblk + per - 1) / per) * dim;
}
/* slot vector of ciphertext t = g*dim + i: splitIntoSquareMatrices (:57-86, zero padded),
 * preprocessToDiagonalForm (:99-115, diag[i][r] = M[r][(r+i) mod dim]), concatenateRows (:118-156,
 * slots[j*dim + r] = diag_{block g*per + j}[i][r]) */
void hyo_enroll_layout_row(const hy_params *p, const double *db, size_t n, size_t t, double *slots) {
    size_t dim = p->dim, per = p->slots / dim;
    size_t g = t / dim, i = t % dim;
    for (size_t j = 0; j < per; j++) {
        size_t blk = g * per + j;
        for (size_t r = 0; r < dim; r++) {
            size_t v = blk * dim + r;
            slots[j * dim + r] = v < n ? db[v * dim + (r + i) % dim] : 0.0;
        }
    }
}
#define DB_NONCE_BASE (1ull << 36)
/* DiagonalEnroller::serializeDB, enroller_diag.cpp:12-53 — normalises db IN PLACE like the reference;
 * the ciphertexts stay in memory instead of serial/db_diagonal/index<t>.bin */
hy_ct **hyo_enroll(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32], size_t *n_cts) {
    size_t dim = p->dim;
#pragma omp parallel for
    for (size_t v = 0; v < n; v++) hyo_normalize(db + v * dim, (int)dim);
    size_t T = hyo_enroll_num_cts(p, n);
    hy_ct **out = (hy_ct **)calloc(T, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (size_t t = 0; t < T; t++) {
        double *slots = (double *)malloc(sizeof(double) * p->slots);
        hyo_enroll_layout_row(p, db, n, t, slots);
        out[t] = hyo_encrypt(p, k, slots, p->slots, seed, DB_NONCE_BASE + t);
        free(slots);
    }
    *n_cts = T;
    return out;
}

/* ------------------------------------------------------------------ receiver */
/* DiagonalReceiver::encryptQuery, receiver_diag.cpp:13-26 */
hy_ct *hyo_encrypt_query(const hy_params *p, const hy_keys *k, const double *query, const uint8_t seed[32], u64 nonce) {
    int dim = p->dim;
    double *qn = (double *)malloc(sizeof(double) * dim);
    memcpy(qn, query, sizeof(double) * dim);
    hyo_normalize(qn, dim);
    double *batch = (double *)malloc(sizeof(double) * p->slots);
    for (int i = 0; i < p->slots; i += dim) memcpy(batch + i, qn, sizeof(double) * dim);
    hy_ct *ct = hyo_encrypt(p, k, batch, p->slots, seed, nonce);
    free(qn);
    free(batch);
    return ct;
}
/* HersReceiver::decryptMembership, receiver_hers.cpp:26-35 */
int hyo_decrypt_membership(const hy_params *p, const hy_keys *k, const hy_ct *c) {
    double *v = (double *)malloc(sizeof(double) * p->slots);
    hyo_decrypt(p, k, c, v);
    int r = v[0] >= 1.0;
    free(v);
    return r;
}
/* HersReceiver::decryptIndex, receiver_hers.cpp:37-54 */
size_t hyo_decrypt_index(const hy_params *p, const hy_keys *k, hy_ct **cts, size_t n_cts, size_t *out, size_t cap) {
    double *v = (double *)malloc(sizeof(double) * p->slots);
    size_t cnt = 0;
    for (size_t i = 0; i < n_cts; i++) {
        hyo_decrypt(p, k, cts[i], v);
        for (size_t j = 0; j < (size_t)p->slots; j++)
            if (v[j] >= 1.0) {
                if (cnt < cap) out[cnt] = j + i * p->slots;
                cnt++;
            }
    }
    free(v);
    return cnt;
}

/* ------------------------------------------------------------------ sender */
/* loop A, sender_diag.cpp:20-26: rot[0] = q, rot[i] = EvalFastRotation(q, i, 2N, precomp) */
hy_ct **hyo_rotate_query(const hy_params *p, const hy_keys *k, const hy_ct *q) {
    int dim = p->dim, N = p->N;
    hy_ct **rot = (hy_ct **)calloc(dim, sizeof(hy_ct *));
    rot[0] = hyo_ct_clone(p, q);
    u64 *dig = hyo_hoist_precompute(p, q->d + (size_t)q->nl * N, q->nl);
#pragma omp parallel for schedule(dynamic)
    for (int i = 1; i < dim; i++) {
        const u64 *evk = hyo_keys_rot(k, i);
        rot[i] = evk ? hyo_rotate_hoisted(p, q, dig, evk, i) : NULL;
        if (!evk) fprintf(stderr, "hydia oracle: missing rotation key %d\n", i);
    }
    free(dig);
    return rot;
}
/* computeSimilarityMatrix + computeSimilarityThread, sender_diag.cpp:66-94: dim products without
 * relinearisation, summed, ONE relinearise and ONE rescale */
hy_ct *hyo_similarity_block(const hy_params *p, const hy_keys *k, hy_ct **rot, hy_ct **db_block) {
    int dim = p->dim;
    hy_ct **score = (hy_ct **)calloc(dim, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < dim; i++) score[i] = hyo_mult_norelin(p, rot[i], db_block[i]);
    for (int i = 1; i < dim; i++) {
        hyo_add_inplace(p, score[0], score[i]);
        hyo_ct_free(score[i]);
    }
    hy_ct *acc = score[0];
    free(score);
    hyo_relin_inplace(p, k, acc);
    hyo_rescale_inplace(p, acc);
    return acc;
}
/* DiagonalSender::computeSimilarity, sender_diag.cpp:12-33 */
hy_ct **hyo_compute_similarity(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out) {
    size_t G = (n + p->slots - 1) / p->slots; /* :16 */
    hy_ct **rot = hyo_rotate_query(p, k, q);
    hy_ct **sim = (hy_ct **)calloc(G, sizeof(hy_ct *));
    for (size_t m = 0; m < G; m++) sim[m] = hyo_similarity_block(p, k, rot, db + m * p->dim);
    hyo_ct_array_free(rot, p->dim);
    *n_out = G;
    return sim;
}
/* DiagonalSender::indexScenario, sender_diag.cpp:52-63 */
hy_ct **hyo_index_scenario(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out) {
    hy_ct **score = hyo_compute_similarity(p, k, q, db, n, n_out);
#pragma omp parallel for
    for (size_t i = 0; i < *n_out; i++) {
        hy_ct *c = hyo_chebyshev_compare(p, k, score[i], MATCH_THRESHOLD, COMP_DEPTH);
        hyo_ct_free(score[i]);
        score[i] = c;
    }
    return score;
}
/* ------------------------------------------------------------------ baby-step / giant-step form of the same mat-vec
 * BASELINE.json's north_star asks for "BSGS rotations of the diagonalized mat-vec"; the reference itself hoists all dim - 1
 * rotations (sender_diag.cpp:22-26, SURVEY "fact 2", which allows the reorganisation when decrypted results stay within 1e-4).
 * With i = b + B g (B babies, dim / B giants) and Rot_i = Rot_{Bg} o Rot_b:
 *     sum_i Rot_i(q) . db_i  =  sum_g Rot_{Bg}( sum_b Rot_b(q) . Rot_{-Bg}(db_{b + Bg}) )
 * The enroller rotates diagonal i by -B (i div B) slots IN THE CLEAR before encrypting it (same ciphertext order, same nonces), the
 * sender needs B - 1 hoisted rotations of the query instead of dim - 1, relinearises the dim / B partial sums of a block and
 * rotates them by B g (ordinary key switches with the rotation keys B, 2B, ...), adds, rescales.  Worth it while the blocks on a
 * GPU are few (the per-block giant steps cost more than the per-query babies they replace beyond ~4 blocks). */
/* smallest power of two B with B*B >= dim: the classic split (32 babies x 16 giants at dim 512).  Any power of two B dividing dim
 * is a valid split — more babies cost rotations per QUERY, fewer giants save key switches per BLOCK — and B = dim is the
 * reference's own all-hoisted form (no pre-rotation, no giant step). */
int hyo_bsgs_babies(const hy_params *p) {
    int B = 1;
    while (B * B < p->dim) B <<= 1;
    return B;
}
void hyo_enroll_layout_row_bsgs(const hy_params *p, const double *db, size_t n, size_t t, double *slots, int B) {
    size_t S = p->slots, i = t % p->dim, sh = (size_t)B * (i / (size_t)B);
    double *plain = (double *)malloc(sizeof(double) * S);
    hyo_enroll_layout_row(p, db, n, t, plain);
    for (size_t s = 0; s < S; s++) slots[s] = plain[(s + S - sh % S) % S]; /* Rot_{-sh}: slot s takes slot s - sh */
    free(plain);
}
hy_ct **hyo_enroll_bsgs(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32], size_t *n_cts, int B) {
    size_t dim = p->dim;
#pragma omp parallel for
    for (size_t v = 0; v < n; v++) hyo_normalize(db + v * dim, (int)dim);
    size_t T = hyo_enroll_num_cts(p, n);
    hy_ct **out = (hy_ct **)calloc(T, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (size_t t = 0; t < T; t++) {
        double *slots = (double *)malloc(sizeof(double) * p->slots);
        hyo_enroll_layout_row_bsgs(p, db, n, t, slots, B);
        out[t] = hyo_encrypt(p, k, slots, p->slots, seed, DB_NONCE_BASE + t);
        free(slots);
    }
    *n_cts = T;
    return out;
}
hy_ct **hyo_compute_similarity_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out, int B) {
    const int dim = p->dim, N = p->N, NG = (dim + B - 1) / B;
    size_t G = (n + p->slots - 1) / p->slots;
    hy_ct **rot = (hy_ct **)calloc(B, sizeof(hy_ct *));
    rot[0] = hyo_ct_clone(p, q);
    u64 *dig = hyo_hoist_precompute(p, q->d + (size_t)q->nl * N, q->nl);
#pragma omp parallel for schedule(dynamic)
    for (int b = 1; b < B; b++) rot[b] = hyo_rotate_hoisted(p, q, dig, hyo_keys_rot(k, b), b);
    free(dig);
    hy_ct **sim = (hy_ct **)calloc(G, sizeof(hy_ct *));
    for (size_t m = 0; m < G; m++) {
        hy_ct **part = (hy_ct **)calloc(NG, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
        for (int g = 0; g < NG; g++) {
            hy_ct *acc = NULL;
            for (int b = 0; b < B && g * B + b < dim; b++) {
                hy_ct *t = hyo_mult_norelin(p, rot[b], db[m * dim + g * B + b]);
                if (!acc) acc = t;
                else {
                    hyo_add_inplace(p, acc, t);
                    hyo_ct_free(t);
                }
            }
            hyo_relin_inplace(p, k, acc);
            if (g > 0) {
                hy_ct *r = hyo_rotate(p, k, acc, g * B);
                hyo_ct_free(acc);
                acc = r;
            }
            part[g] = acc;
        }
        for (int g = 1; g < NG; g++) {
            hyo_add_inplace(p, part[0], part[g]);
            hyo_ct_free(part[g]);
        }
        sim[m] = part[0];
        free(part);
        hyo_rescale_inplace(p, sim[m]);
    }
    hyo_ct_array_free(rot, B);
    *n_out = G;
    return sim;
}
hy_ct **hyo_index_scenario_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out, int B) {
    hy_ct **score = hyo_compute_similarity_bsgs(p, k, q, db, n, n_out, B);
#pragma omp parallel for
    for (size_t i = 0; i < *n_out; i++) {
        hy_ct *c = hyo_chebyshev_compare(p, k, score[i], MATCH_THRESHOLD, COMP_DEPTH);
        hyo_ct_free(score[i]);
        score[i] = c;
    }
    return score;
}
hy_ct *hyo_membership_scenario_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, int B) {
    size_t G;
    hy_ct **score = hyo_index_scenario_bsgs(p, k, q, db, n, &G, B);
    hy_ct *m = score[0];
    for (size_t i = 1; i < G; i++) {
        hyo_add_inplace(p, m, score[i]);
        hyo_ct_free(score[i]);
    }
    free(score);
    for (int r = 1; r < p->slots; r <<= 1) {
        hy_ct *t = hyo_rotate(p, k, m, r);
        hyo_add_inplace(p, m, t);
        hyo_ct_free(t);
    }
    return m;
}

/* ---- the reference's on-disk hand-off between enroller and sender.  DiagonalEnroller::serializeDBThread writes one file per
 * ciphertext, "serial/db_diagonal/index<t>.bin" (enroller_diag.cpp:158-166), and computeSimilarityThread reads its file back
 * INSIDE the timed parallel loop of every query (sender_diag.cpp:85-94).  OpenFHE's cereal BINARY layout is undocumented in the
 * reference and not reproduced: a file here is a small header + the raw residues.  Used by bench.py's cpu_baseline to time the
 * variant that pays the reference's per-query I/O. */
int hyo_db_write_files(const hy_params *p, hy_ct **db, size_t count, const char *dir) {
    int ok = 1;
#pragma omp parallel for schedule(dynamic) reduction(&& : ok)
    for (size_t t = 0; t < count; t++) {
        char path[4096];
        snprintf(path, sizeof path, "%s/index%zu.bin", dir, t);
        FILE *f = fopen(path, "wb");
        if (!f) {
            ok = 0;
            continue;
        }
        int hdr[2] = {db[t]->npoly, db[t]->nl};
        size_t n = (size_t)db[t]->npoly * db[t]->nl * p->N;
        ok = ok && fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(&db[t]->scale, sizeof(double), 1, f) == 1 &&
             fwrite(db[t]->d, sizeof(u64), n, f) == n;
        fclose(f);
    }
    return ok ? 0 : -1;
}
static hy_ct *read_ct_file(const hy_params *p, const char *dir, size_t t) {
    char path[4096];
    snprintf(path, sizeof path, "%s/index%zu.bin", dir, t);
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "Error: Cannot read serialization from %s\n", path); /* sender_diag.cpp:89-91 */
        return NULL;
    }
    int hdr[2];
    double scale;
    hy_ct *c = NULL;
    if (fread(hdr, sizeof hdr, 1, f) == 1 && fread(&scale, sizeof scale, 1, f) == 1) {
        c = hyo_ct_alloc(p, hdr[0], hdr[1], scale);
        size_t n = (size_t)hdr[0] * hdr[1] * p->N;
        if (fread(c->d, sizeof(u64), n, f) != n) {
            hyo_ct_free(c);
            c = NULL;
        }
    }
    fclose(f);
    return c;
}
/* indexScenario whose loop B deserialises index<m*dim+i>.bin per product (sender_diag.cpp:66-94), then compares (:57-60) */
hy_ct **hyo_index_scenario_files(const hy_params *p, const hy_keys *k, const hy_ct *q, const char *dir, size_t n, size_t *n_out) {
    size_t G = (n + p->slots - 1) / p->slots;
    int dim = p->dim;
    hy_ct **rot = hyo_rotate_query(p, k, q);
    hy_ct **score = (hy_ct **)calloc(G, sizeof(hy_ct *));
    for (size_t m = 0; m < G; m++) {
        hy_ct **prod = (hy_ct **)calloc(dim, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
        for (int i = 0; i < dim; i++) {
            hy_ct *dbc = read_ct_file(p, dir, m * dim + i);
            if (dbc) {
                prod[i] = hyo_mult_norelin(p, rot[i], dbc);
                hyo_ct_free(dbc);
            }
        }
        hy_ct *acc = NULL;
        for (int i = 0; i < dim; i++) {
            if (!prod[i]) continue;
            if (!acc) acc = prod[i];
            else {
                hyo_add_inplace(p, acc, prod[i]);
                hyo_ct_free(prod[i]);
            }
        }
        free(prod);
        if (acc) {
            hyo_relin_inplace(p, k, acc);
            hyo_rescale_inplace(p, acc);
        }
        score[m] = acc;
    }
    hyo_ct_array_free(rot, dim);
#pragma omp parallel for
    for (size_t i = 0; i < G; i++) {
        if (!score[i]) continue;
        hy_ct *c = hyo_chebyshev_compare(p, k, score[i], MATCH_THRESHOLD, COMP_DEPTH);
        hyo_ct_free(score[i]);
        score[i] = c;
    }
    *n_out = G;
    return score;
}
/* DiagonalSender::membershipScenario, sender_diag.cpp:35-50: EvalAddManyInPlace then EvalSum over batchSize */
hy_ct *hyo_membership_scenario(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n) {
    size_t G;
    hy_ct **score = hyo_index_scenario(p, k, q, db, n, &G);
    hy_ct *m = score[0];
    for (size_t i = 1; i < G; i++) {
        hyo_add_inplace(p, m, score[i]);
        hyo_ct_free(score[i]);
    }
    free(score);
    for (int r = 1; r < p->slots; r <<= 1) {
        hy_ct *t = hyo_rotate(p, k, m, r);
        hyo_add_inplace(p, m, t);
        hyo_ct_free(t);
    }
    return m;
}

/* ------------------------------------------------------------------ HERS (approach 4), SURVEY §8f-4
 * The paper's main comparison on the same primitives: index-batched (column) packing, one query ciphertext per
 * dimension, relinearise + rescale after EVERY product.
 *   HersEnroller::serializeDB / encryptDBThread   /root/reference/src/enroller/enroller_hers.cpp:40-121
 *   HersReceiver::encryptQuery / encryptQueryThread  /root/reference/src/receiver/receiver_hers.cpp:13-24, :58-63
 *   HersSender::computeSimilarity / Helper / Serial  /root/reference/src/sender/sender_hers.cpp:13-98 */
#define HERS_NONCE_BASE (1ull << 37)
/* ciphertext (m, j): slot k holds coordinate j of database vector m*slots + k (enroller_hers.cpp:108-113) */
void hyo_hers_layout_row(const hy_params *p, const double *db, size_t n, size_t t, double *slots) {
    size_t dim = p->dim, S = p->slots, m = t / dim, j = t % dim;
    for (size_t k = 0; k < S; k++) {
        size_t v = m * S + k;
        slots[k] = v < n ? db[v * dim + j] : 0.0;
    }
}
hy_ct **hyo_hers_enroll(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32], size_t *n_cts) {
    size_t dim = p->dim;
#pragma omp parallel for
    for (size_t v = 0; v < n; v++) hyo_normalize(db + v * dim, (int)dim); /* enroller_hers.cpp:75-78 */
    size_t G = (n + p->slots - 1) / p->slots, T = G * dim;               /* :59-60 */
    hy_ct **out = (hy_ct **)calloc(T, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (size_t t = 0; t < T; t++) {
        double *slots = (double *)malloc(sizeof(double) * p->slots);
        hyo_hers_layout_row(p, db, n, t, slots);
        out[t] = hyo_encrypt(p, k, slots, p->slots, seed, HERS_NONCE_BASE + t);
        free(slots);
    }
    *n_cts = T;
    return out;
}
/* one ciphertext per dimension, the normalised coordinate broadcast to every slot (receiver_hers.cpp:13-24, :58-63) */
hy_ct **hyo_hers_encrypt_query(const hy_params *p, const hy_keys *k, const double *query, const uint8_t seed[32], u64 nonce0) {
    int dim = p->dim;
    double *qn = (double *)malloc(sizeof(double) * dim);
    memcpy(qn, query, sizeof(double) * dim);
    hyo_normalize(qn, dim);
    hy_ct **out = (hy_ct **)calloc(dim, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < dim; i++) {
        double *v = (double *)malloc(sizeof(double) * p->slots);
        for (int s = 0; s < p->slots; s++) v[s] = qn[i];
        out[i] = hyo_encrypt(p, k, v, p->slots, seed, nonce0 + (u64)i);
        free(v);
    }
    free(qn);
    return out;
}
/* computeSimilarityHelper (sender_hers.cpp:62-87): EvalMultNoRelin, Relinearize, Rescale per dimension, then the sum */
hy_ct *hyo_hers_similarity_block(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db_block) {
    int dim = p->dim;
    hy_ct **score = (hy_ct **)calloc(dim, sizeof(hy_ct *));
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < dim; i++) {
        score[i] = hyo_mult_norelin(p, q[i], db_block[i]);
        hyo_relin_inplace(p, k, score[i]);
        hyo_rescale_inplace(p, score[i]);
    }
    for (int i = 1; i < dim; i++) {
        hyo_add_inplace(p, score[0], score[i]);
        hyo_ct_free(score[i]);
    }
    hy_ct *acc = score[0];
    free(score);
    return acc;
}
hy_ct **hyo_hers_compute_similarity(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n, size_t *n_out) {
    size_t G = (n + p->slots - 1) / p->slots; /* sender_hers.cpp:16 */
    hy_ct **sim = (hy_ct **)calloc(G, sizeof(hy_ct *));
    for (size_t m = 0; m < G; m++) sim[m] = hyo_hers_similarity_block(p, k, q, db + m * p->dim);
    *n_out = G;
    return sim;
}
/* indexScenario / membershipScenario of HersSender (sender_hers.cpp:28-58): same tails as the diagonal sender */
hy_ct **hyo_hers_index_scenario(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n, size_t *n_out) {
    hy_ct **score = hyo_hers_compute_similarity(p, k, q, db, n, n_out);
#pragma omp parallel for
    for (size_t i = 0; i < *n_out; i++) {
        hy_ct *c = hyo_chebyshev_compare(p, k, score[i], MATCH_THRESHOLD, COMP_DEPTH);
        hyo_ct_free(score[i]);
        score[i] = c;
    }
    return score;
}
hy_ct *hyo_hers_membership_scenario(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n) {
    size_t G;
    hy_ct **score = hyo_hers_index_scenario(p, k, q, db, n, &G);
    hy_ct *m = score[0];
    for (size_t i = 1; i < G; i++) {
        hyo_add_inplace(p, m, score[i]);
        hyo_ct_free(score[i]);
    }
    free(score);
    for (int r = 1; r < p->slots; r <<= 1) {
        hy_ct *t = hyo_rotate(p, k, m, r);
        hyo_add_inplace(p, m, t);
        hyo_ct_free(t);
    }
    return m;
}

hy_ct *hyo_ct_at(hy_ct **arr, size_t i) { return arr[i]; }
void hyo_ct_array_free(hy_ct **arr, size_t n) {
    for (size_t i = 0; i < n; i++) hyo_ct_free(arr[i]);
    free(arr);
}

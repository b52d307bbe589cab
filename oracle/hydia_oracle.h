/*
 * oracle/hydia_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the HyDia (approach 5) hot path of n7koirala/image_matching and of the
 * CKKS-RNS arithmetic it stands on.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (image_matching_amd/) never links, imports or calls it.
 *
 * Parity status: the ORCHESTRATION follows the reference line by line (each function cites the
 * file:line it restates).  The ARITHMETIC lives in OpenFHE v1.2.3, a third-party dependency that is
 * not vendored under /root/reference (dockerfile:7, :25-30) and is absent from this image, so it is
 * restated from the published CKKS-RNS / hybrid key-switching algorithms.  Ciphertext-level parity
 * with OpenFHE is therefore UNPINNED; what IS pinned (tests/test_oracle_golden.py) are the decrypted
 * results the reference's own files hold: test/2_10.dat, test/2_11.dat -> membership true, index [0],
 * scores within 1e-4 of plaintext cosine (src/main_accuracy.cpp:359-360), and the comparator transfer
 * curve tools/figures/signApprox.csv (column "combined").
 */
#ifndef HYDIA_ORACLE_H
#define HYDIA_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "modarith.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HY_MAX_LIMBS 32

/* ---- deterministic randomness: ChaCha20 keyed by a 32-byte seed; (stream, block) addressable ---- */
enum {
    HY_DOM_SK = 1,
    HY_DOM_PK_A = 2,
    HY_DOM_PK_E = 3,
    HY_DOM_EVK_A = 4,
    HY_DOM_EVK_E = 5,
    HY_DOM_ENC_U = 6,
    HY_DOM_ENC_E0 = 7,
    HY_DOM_ENC_E1 = 8
};
#define HY_STREAM(dom, a, b, c) \
    (((u64)(dom) << 56) | ((u64)(a) << 16) | ((u64)(b) << 8) | (u64)(c))
/* key ids inside HY_DOM_EVK_*: 0 = relinearisation key, r >= 1 = rotation by r slots */

typedef struct hy_params {
    int logN, N, nQ, nP, nT, dnum, alpha, scale_bits, first_bits, dim, slots;
    double delta; /* 2^scale_bits */
    u64 q[HY_MAX_LIMBS];
    barrett_t bq[HY_MAX_LIMBS];
    u64 psi[HY_MAX_LIMBS], psi_inv[HY_MAX_LIMBS], n_inv[HY_MAX_LIMBS], n_inv_sh[HY_MAX_LIMBS];
    u64 *tw[HY_MAX_LIMBS], *tw_sh[HY_MAX_LIMBS];   /* psi^{bitrev(k)}   , k < N */
    u64 *itw[HY_MAX_LIMBS], *itw_sh[HY_MAX_LIMBS]; /* psi^{-bitrev(k)}  , k < N */
    u64 P_mod_q[HY_MAX_LIMBS], Pinv_mod_q[HY_MAX_LIMBS];
    u64 Phat_inv[HY_MAX_LIMBS];                    /* (P/p_k)^{-1} mod p_k */
    u64 Phat_mod_q[HY_MAX_LIMBS][HY_MAX_LIMBS];    /* [k][j] = (P/p_k) mod q_j */
    u64 ql_inv[HY_MAX_LIMBS][HY_MAX_LIMBS];        /* [l][j] = q_l^{-1} mod q_j, j < l */
    uint32_t *rot_group;                           /* 5^j mod 2N, j < N/2 */
    double *ksi_re, *ksi_im;                       /* exp(2 pi i k / 2N), k <= 2N */
} hy_params;

/* ciphertext in evaluation (NTT, bit-reversed) form: d[(p*nl + j)*N + c], limb j <-> modulus q_j */
typedef struct hy_ct {
    int npoly, nl;
    double scale;
    u64 *d;
} hy_ct;

typedef struct hy_keys {
    int8_t *s_coeff; /* ternary secret, coefficient form */
    u64 *s_ntt;      /* [nT][N] */
    u64 *pk;         /* [2][nQ][N]: (b, a) */
    u64 *relin;      /* [dnum][2][nT][N] */
    int n_rot;
    int *rot_idx;    /* rotation amounts */
    u64 **rot;       /* each [dnum][2][nT][N] */
} hy_keys;

/* ---- params / primitives ---- */
hy_params *hyo_params_create(int logN, int mult_depth, int scale_bits, int first_bits, int dnum, int dim);
hy_params *hyo_params_create_custom(int logN, int nQ, int nP, int scale_bits, int dnum, int dim, const u64 *moduli,
                                    const u64 *roots);
void hyo_params_free(hy_params *p);
void hyo_get_moduli(const hy_params *p, u64 *out);
void hyo_get_roots(const hy_params *p, u64 *out);
int hyo_get_info(const hy_params *p, int *out8);
void hyo_ntt_fwd(const hy_params *p, u64 *a, int m);
void hyo_ntt_inv(const hy_params *p, u64 *a, int m);
u64 hyo_galois_elt(const hy_params *p, int rot);
void hyo_automorph_eval(const hy_params *p, const u64 *in, u64 *out, u64 g);
void hyo_automorph_coeff(const hy_params *p, const u64 *in, u64 *out, u64 g, u64 q);
void hyo_chacha_block(const uint8_t seed[32], u64 stream, u64 block, uint32_t out[16]);
void hyo_sample_uniform(const uint8_t seed[32], u64 stream, u64 q, u64 *out, int n);
void hyo_sample_ternary(const uint8_t seed[32], u64 stream, int8_t *out, int n);
void hyo_sample_gauss(const uint8_t seed[32], u64 stream, int32_t *out, int n);

/* ---- CKKS client ---- */
void hyo_encode(const hy_params *p, const double *slots, int n_in, double scale, int nl, u64 *out);
void hyo_encode_coeffs(const hy_params *p, const double *slots, int n_in, double scale, int64_t *coeffs);
void hyo_decode(const hy_params *p, const u64 *poly_coeff, int nl, double scale, double *out);
hy_keys *hyo_keygen(const hy_params *p, const uint8_t seed[32], const int *rot_idx, int n_rot);
void hyo_keys_free(hy_keys *k);
const u64 *hyo_keys_rot(const hy_keys *k, int rot);
hy_ct *hyo_ct_alloc(const hy_params *p, int npoly, int nl, double scale);
hy_ct *hyo_ct_clone(const hy_params *p, const hy_ct *a);
void hyo_ct_free(hy_ct *c);
u64 *hyo_ct_data(hy_ct *c);
int hyo_ct_nl(const hy_ct *c);
int hyo_ct_npoly(const hy_ct *c);
double hyo_ct_scale(const hy_ct *c);
hy_ct *hyo_encrypt(const hy_params *p, const hy_keys *k, const double *slots, int n_in,
                   const uint8_t seed[32], u64 nonce);
void hyo_decrypt(const hy_params *p, const hy_keys *k, const hy_ct *c, double *out);

/* ---- CKKS evaluation ---- */
void hyo_keyswitch(const hy_params *p, const u64 *c, int nl, const u64 *evk, u64 *out0, u64 *out1);
u64 *hyo_hoist_precompute(const hy_params *p, const u64 *c1, int nl);
hy_ct *hyo_rotate_hoisted(const hy_params *p, const hy_ct *c, const u64 *digits, const u64 *evk, int rot);
hy_ct *hyo_rotate(const hy_params *p, const hy_keys *k, const hy_ct *c, int rot);
hy_ct *hyo_mult_norelin(const hy_params *p, const hy_ct *a, const hy_ct *b);
void hyo_add_inplace(const hy_params *p, hy_ct *a, const hy_ct *b);
void hyo_sub_inplace(const hy_params *p, hy_ct *a, const hy_ct *b);
void hyo_submul_int(const hy_params *p, hy_ct *a, const hy_ct *b, u64 K);
void hyo_relin_inplace(const hy_params *p, const hy_keys *k, hy_ct *a);
void hyo_rescale_inplace(const hy_params *p, hy_ct *a);
void hyo_drop_to(const hy_params *p, hy_ct *a, int nl);
void hyo_add_const(const hy_params *p, hy_ct *a, double c);
hy_ct *hyo_mul_const(const hy_params *p, const hy_ct *a, double c, double const_scale);
hy_ct *hyo_mult(const hy_params *p, const hy_keys *k, const hy_ct *a, const hy_ct *b);

/* ---- comparator (src/openFHE_wrapper.cpp:143-185) ---- */
void hyo_chebyshev_step_coeffs(double delta, int degree, double *coeffs);
double hyo_compare_plain(double x, double delta, int degree);
hy_ct *hyo_eval_chebyshev63(const hy_params *p, const hy_keys *k, const hy_ct *x, const double *coeffs, int degree);
hy_ct *hyo_eval_f4(const hy_params *p, const hy_keys *k, const hy_ct *y);
hy_ct *hyo_chebyshev_compare(const hy_params *p, const hy_keys *k, const hy_ct *x, double delta, int sign_depth);

/* ---- HyDia roles (src/{enroller,receiver,sender}/..._diag.cpp, src/receiver/receiver_hers.cpp) ---- */
void hyo_normalize(double *x, int dim);
size_t hyo_enroll_num_cts(const hy_params *p, size_t n);
void hyo_enroll_layout_row(const hy_params *p, const double *db_norm, size_t n, size_t t, double *slots);
hy_ct **hyo_enroll(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32],
                   size_t *n_cts);
hy_ct *hyo_encrypt_query(const hy_params *p, const hy_keys *k, const double *query, const uint8_t seed[32],
                         u64 nonce);
hy_ct **hyo_rotate_query(const hy_params *p, const hy_keys *k, const hy_ct *q);
hy_ct *hyo_similarity_block(const hy_params *p, const hy_keys *k, hy_ct **rot, hy_ct **db_block);
hy_ct **hyo_compute_similarity(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n,
                               size_t *n_out);
hy_ct **hyo_index_scenario(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n,
                           size_t *n_out);
hy_ct *hyo_membership_scenario(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n);
/* baby-step / giant-step form (path.c): pre-rotated diagonals at enrolment, B - 1 hoisted rotations + dim / B giant steps per block */
int hyo_bsgs_babies(const hy_params *p);
/* B = babies: a power of two dividing dim (hyo_bsgs_babies gives the classic square-root split; B = dim is the hoisted form) */
void hyo_enroll_layout_row_bsgs(const hy_params *p, const double *db_norm, size_t n, size_t t, double *slots, int B);
hy_ct **hyo_enroll_bsgs(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32], size_t *n_cts, int B);
hy_ct **hyo_compute_similarity_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out, int B);
hy_ct **hyo_index_scenario_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, size_t *n_out, int B);
hy_ct *hyo_membership_scenario_bsgs(const hy_params *p, const hy_keys *k, const hy_ct *q, hy_ct **db, size_t n, int B);
/* the reference's per-ciphertext file hand-off (enroller_diag.cpp:158-166, sender_diag.cpp:85-94); own raw format */
int hyo_db_write_files(const hy_params *p, hy_ct **db, size_t count, const char *dir);
hy_ct **hyo_index_scenario_files(const hy_params *p, const hy_keys *k, const hy_ct *q, const char *dir, size_t n, size_t *n_out);
int hyo_decrypt_membership(const hy_params *p, const hy_keys *k, const hy_ct *c);
size_t hyo_decrypt_index(const hy_params *p, const hy_keys *k, hy_ct **cts, size_t n_cts, size_t *out,
                         size_t cap);
/* ---- HERS (approach 4): src/{enroller,receiver,sender}/..._hers.cpp ---- */
void hyo_hers_layout_row(const hy_params *p, const double *db_norm, size_t n, size_t t, double *slots);
hy_ct **hyo_hers_enroll(const hy_params *p, const hy_keys *k, double *db, size_t n, const uint8_t seed[32], size_t *n_cts);
hy_ct **hyo_hers_encrypt_query(const hy_params *p, const hy_keys *k, const double *query, const uint8_t seed[32], u64 nonce0);
hy_ct *hyo_hers_similarity_block(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db_block);
hy_ct **hyo_hers_compute_similarity(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n, size_t *n_out);
hy_ct **hyo_hers_index_scenario(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n, size_t *n_out);
hy_ct *hyo_hers_membership_scenario(const hy_params *p, const hy_keys *k, hy_ct **q, hy_ct **db, size_t n);
hy_ct *hyo_ct_at(hy_ct **arr, size_t i);
void hyo_ct_array_free(hy_ct **arr, size_t n);
int hyo_num_threads(void);
void hyo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif

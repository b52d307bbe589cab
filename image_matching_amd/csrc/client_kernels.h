// image_matching_amd/csrc/client_kernels.h — launch interface of client_kernels.hip
#pragma once
#include "kernels.h"

struct ChaChaKey {
    unsigned k[8];
};
// stream ids of the deterministic sampler specification (DESIGN.md §"Randomness")
#define HY_DOM_SK 1ull
#define HY_DOM_PK_A 2ull
#define HY_DOM_PK_E 3ull
#define HY_DOM_EVK_A 4ull
#define HY_DOM_EVK_E 5ull
#define HY_DOM_ENC_U 6ull
#define HY_DOM_ENC_E0 7ull
#define HY_DOM_ENC_E1 8ull
#define HY_STREAM(dom, a, b, c) (((u64)(dom) << 56) | ((u64)(a) << 16) | ((u64)(b) << 8) | (u64)(c))

namespace hc {
void sample_uniform(hipStream_t st, const ChaChaKey &key, const ModC *mod, int N, u64 sbase, u64 step_y, u64 step_z,
                    u64 *dst, size_t stride_y, size_t stride_z, const LimbSel &ysel, int nz);
void sample_ternary(hipStream_t st, const ChaChaKey &key, int N, u64 sbase, u64 step, int *dst, int X);
void sample_gauss(hipStream_t st, const ChaChaKey &key, int N, u64 sbase, u64 step, int *dst, int X);
void small_to_limbs(hipStream_t st, const ModC *mod, int N, const int *a, const long long *m, u64 *out,
                    size_t out_x_stride, int X, const LimbSel &sel);
void automorph(hipStream_t st, int logN, const u64 *in, u64 *out, unsigned g, int nlimbs);
void mul(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, const LimbSel &sel);
void pk_combine(hipStream_t st, const ModC *mod, int N, int nQ, u64 *b, const u64 *a, const u64 *s);
void evk_combine(hipStream_t st, const ModC *mod, int N, int nT, int nQ, int alpha, int dnum, u64 *key, const u64 *e,
                 const u64 *s_enc, const u64 *s_from, const ScaleSel &pmodq);
void enc_combine(hipStream_t st, const ModC *mod, int N, int nQ, const u64 *pk, const u64 *u, const u64 *t0, const u64 *t1,
                 u64 *ct, int X);
void dec_dot(hipStream_t st, const ModC *mod, int N, int npoly, int nl, const u64 *ct, const u64 *s, u64 *t, int nu, int X);
void encode(hipStream_t st, const double *slots, double2 *work, long long *coeffs, int N, int X, double scale,
            const unsigned *rot_group, const double2 *ksi);
void decode(hipStream_t st, const ModC *mod, const u64 *t, int nu, int N, int X, double scale, u64 q0inv_mod_q1,
            double2 *work, double *out, const unsigned *rot_group, const double2 *ksi);
void diag_pack(hipStream_t st, const double *dbg, long long rows_left, int dim, int Nh, double *slots, int babies = 0);
// HERS (approach 4): slots[j][k] = dbg[k][j] (column packing); query coordinate i broadcast to every slot of vector i
void hers_pack(hipStream_t st, const double *dbg, long long rows_left, int dim, int Nh, double *slots);
void broadcast_rows(hipStream_t st, const double *vals, int dim, int Nh, double *slots);
}  // namespace hc

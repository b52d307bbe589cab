// image_matching_amd/csrc/kernels.h — launch interface of the gfx950 kernels (kernels.hip, client_kernels.hip).
//
// Data layout everywhere: limb-major uint64 residues, [batch][poly][limb][N] contiguous, evaluation form in
// bit-reversed order (out[j] = a(psi^{2*bitrev(j)+1})) unless a comment says "coefficient form".  One wave reads 64
// (or 128) consecutive coefficients of ONE limb, so the modulus constants are wave-uniform (SGPRs).
#pragma once
#include <hip/hip_runtime.h>
#include "devmath.h"

// which modulus each limb slot of a buffer uses
struct LimbSel {
    int n;
    int mod[HY_MAX_MODS];
};

struct NttTables {
    const u64 *tw, *tw_sh;    // [nT][N] psi^{bitrev(k)} and Shoup companions
    const u64 *itw, *itw_sh;  // [nT][N] psi^{-bitrev(k)}
    const ModC *mod;          // [nT]
    const ulonglong2 *twp, *itwp;  // the same tables interleaved as (w, w_shoup) pairs: one 16-byte load per twiddle
};

// base conversion table: out[t] = sum_s y[s] * f[s][t] mod q_{dst t}
struct ConvTab {
    int ns, nt;
    int skip_lo, skip_hi;  // target slots in [skip_lo, skip_hi) are left untouched (the digit's own limbs)
    u64 f[HY_MAX_DIGIT][HY_MAX_MODS];
};

struct ScaleSel {  // per-limb multiplier applied by the inverse NTT's last pass (N^{-1} * extra)
    u64 s[HY_MAX_MODS], s_sh[HY_MAX_MODS];
};

namespace hk {

// ---- NTT: X limb-polys of N coefficients; element (x, slot) lives at base + x*outer + slot*N, slot < sel.n
void ntt_forward(hipStream_t st, const NttTables &T, int logN, const u64 *src, u64 *dst, size_t src_outer,
                 size_t dst_outer, int X, const LimbSel &sel);
void ntt_inverse(hipStream_t st, const NttTables &T, int logN, const u64 *src, u64 *dst, size_t src_outer,
                 size_t dst_outer, int X, const LimbSel &sel, const ScaleSel &scale);

// N = 2^15 register-radix fast path (ntt15.hip); ntt_forward / ntt_inverse dispatch to it when logN == 15
void ntt15_forward(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t src_outer, size_t dst_outer, int X,
                   const LimbSel &sel);
void ntt15_inverse(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t src_outer, size_t dst_outer, int X,
                   const LimbSel &sel, const ScaleSel &scale);

// ---- element-wise over [X][sel.n][N]
// XP polynomials; operand t's polynomial xp starts at xp * t_ls * N (limb-strided views of dropped ciphertexts)
void add(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int XP, const LimbSel &sel, int a_ls,
         int b_ls, int o_ls);
void sub(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int XP, const LimbSel &sel, int a_ls,
         int b_ls, int o_ls);
void mul_scalar(hipStream_t st, const ModC *mod, int N, const u64 *a, u64 *o, int XP, const LimbSel &sel,
                const ScaleSel &c, int a_ls, int o_ls);  // o = a * c[slot]
void add_scalar(hipStream_t st, const ModC *mod, int N, u64 *a, size_t outer, int X, const LimbSel &sel,
                const ScaleSel &c);  // a[x][slot] += c[slot]  (first sel.n slots of each outer block)
void copy_limbs(hipStream_t st, int N, const u64 *src, u64 *dst, size_t src_outer, size_t dst_outer, int X,
                int nlimbs);
// (a0 b0, a0 b1 + a1 b0, a1 b1) for X ciphertext pairs at nl limbs; o: [X][3][nl][N]
void tensor(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int X, int nl, int a_ls, int b_ls);

// ---- key switching
// out[x][t][c] = sum_s y[x][s][c] * tab.f[s][t] mod q_{dsel.mod[t]} ; y coefficient form, residues < 2^60
void base_convert(hipStream_t st, const ModC *mod, int N, const u64 *y, size_t y_outer, u64 *out, size_t out_outer,
                  int X, const ConvTab &tab, const LimbSel &dsel);
// acc[x][p][t][c] = sum_d dig[(x*dig_x_stride) + d][t][c] * key_x[d][p][mod(t)][c];  keys[x] -> [dnum][2][nT][N]
// (same_key: every x uses keys[0])
void inner_product(hipStream_t st, const ModC *mod, int N, const u64 *dig, size_t dig_x_stride, int nd,
                   const u64 *const *keys, int same_key, int nT, u64 *acc, int X, const LimbSel &esel);
// out[x][p][j][c'] = ((acc[x][p][j][c] - conv[x][p][j][c]) * pinv[j] + (addend ? addend[x*add_x + p*add_ps + j*N + c] : 0)),
// c = perm_g(c') when galois[x] != 1 (evaluation-form automorphism), acc rows have stride acc_limbs*N
void moddown_combine(hipStream_t st, const ModC *mod, int logN, const u64 *acc, int acc_limbs, const u64 *conv,
                     const u64 *addend, size_t add_x_stride, size_t add_poly_stride, int add_polys, u64 *out, int X, int nl,
                     const ScaleSel &pinv, const unsigned *galois /* device [X] (or [1] with same_g) or null */,
                     int same_g);
// rescale: t = last limb in coefficient form [X][N]; tmp[x][j][c] = centred t mod q_j (coefficient form)
void rescale_spread(hipStream_t st, const ModC *mod, int N, const u64 *t, u64 *tmp, int X, int l);
// out[x][j][c] = (in[x][j][c] - tmp[x][j][c]) * qlinv[j]; in has nl=l+1 limbs per x, out has l
void rescale_combine(hipStream_t st, const ModC *mod, int N, const u64 *in, const u64 *tmp, u64 *out, int X, int l,
                     const ScaleSel &qlinv, int in_ls);

// ---- loop B of the HyDia sender: acc[g][3][nl][N] = sum_i rot[i] (x) db[g][i], fully reduced
void hydia_tensor_accumulate(hipStream_t st, const ModC *mod, int N, const u64 *rot, const u64 *db, u64 *acc, int G,
                             int dim, int nl, int bpp);
const char *hydia_tensor_kernel_name();

// ---- misc
void fill_uniform_hash(hipStream_t st, const ModC *mod, int N, u64 *dst, size_t n_limbpolys, int nl,
                       unsigned long long seed);  // bench filler: residues < q_limb from a counter hash
}  // namespace hk

// image_matching_amd/csrc/client_kernels.hip — gfx950 kernels of the receiver / enroller / key-generation side:
// ChaCha20-addressed samplers, the canonical-embedding FFT (CKKS encode/decode), public-key encryption, decryption
// and the diagonal packing of DiagonalEnroller.
//
// Replaces, on the GPU: MakeCKKSPackedPlaintext + Encrypt (/root/reference/src/openFHE_wrapper.cpp:74-77), Decrypt +
// GetRealPackedValue (:81-85), KeyGen/EvalMultKeyGen/EvalRotateKeyGen (/root/reference/src/main.cpp:184-206) and the
// packing loops of /root/reference/src/enroller/enroller_diag.cpp:57-156.
// Floating point here is compiled with -ffp-contract=off and uses the same operation order as the specification in
// DESIGN.md, so encodings are reproducible bit for bit.
#include "client_kernels.h"
#include "gauss_cdt.h"

namespace {

__device__ __constant__ u64 d_gauss_cdt[HYDIA_GAUSS_CDT_LEN] = HYDIA_GAUSS_CDT_VALUES;

DEV unsigned rotl32(unsigned v, int n) { return (v << n) | (v >> (32 - n)); }
#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

// ChaCha20 block with a 64-bit block counter (words 12,13) and a 64-bit stream id (words 14,15)
DEV void chacha_block(const ChaChaKey &key, u64 stream, u64 block, unsigned out[16]) {
    unsigned s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; i++) s[4 + i] = key.k[i];
    s[12] = (unsigned)block; s[13] = (unsigned)(block >> 32);
    s[14] = (unsigned)stream; s[15] = (unsigned)(stream >> 32);
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#pragma unroll
    for (int r = 0; r < 10; r++) {
        CHACHA_QR(x[0], x[4], x[8], x[12]) CHACHA_QR(x[1], x[5], x[9], x[13]) CHACHA_QR(x[2], x[6], x[10], x[14]) CHACHA_QR(x[3], x[7], x[11], x[15])
        CHACHA_QR(x[0], x[5], x[10], x[15]) CHACHA_QR(x[1], x[6], x[11], x[12]) CHACHA_QR(x[2], x[7], x[8], x[13]) CHACHA_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// uniform residues: thread = one ChaCha block = 4 coefficients (128 random bits each, reduced mod q).
// grid (N/4/256 rounded up, ny, nz): stream = base + y*step_y + z*step_z ; out = dst + y*stride_y + z*stride_z
__global__ __launch_bounds__(256) void k_sample_uniform(ChaChaKey key, const ModC *__restrict__ mod, int N, u64 sbase,
                                                        u64 step_y, u64 step_z, u64 *__restrict__ dst, size_t stride_y,
                                                        size_t stride_z, LimbSel ysel) {
    const int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk * 4 >= N) return;
    const int y = blockIdx.y, z = blockIdx.z;
    const ModC M = mod[ysel.mod[y]];
    unsigned w[16];
    chacha_block(key, sbase + (u64)y * step_y + (u64)z * step_z, (u64)blk, w);
    u64 *o = dst + (size_t)y * stride_y + (size_t)z * stride_z + (size_t)blk * 4;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const u64 lo = (u64)w[4 * t] | ((u64)w[4 * t + 1] << 32);
        const u64 hi = (u64)w[4 * t + 2] | ((u64)w[4 * t + 3] << 32);
        o[t] = reduce128(((u128)hi << 64) | lo, M);
    }
}
// small signed polynomials: MODE 0 ternary (16 per block), MODE 1 discrete Gaussian sigma 3.19 (8 per block).
// grid (ceil(N/per/256), X): stream = base + x*step ; out int32 [X][N]
template <int MODE>
__global__ __launch_bounds__(256) void k_sample_small(ChaChaKey key, int N, u64 sbase, u64 step, int *__restrict__ dst) {
    constexpr int PER = MODE == 0 ? 16 : 8;
    const int blk = blockIdx.x * 256 + threadIdx.x;
    if (blk * PER >= N) return;
    const int x = blockIdx.y;
    unsigned w[16];
    chacha_block(key, sbase + (u64)x * step, (u64)blk, w);
    int *o = dst + (size_t)x * N + (size_t)blk * PER;
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 16; t++) o[t] = (int)(((u64)w[t] * 3ull) >> 32) - 1;
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const u64 r = (u64)w[2 * t] | ((u64)w[2 * t + 1] << 32);
            const u64 u = r >> 1;
            int k = 0;
            while (u >= d_gauss_cdt[k]) k++;
            o[t] = (r & 1) ? -k : k;
        }
    }
}
// out[x][slot][c] = (a[x][c] (+ m[x][c])) mod q_slot ; a int32, m int64 (optional). grid (N/256, sel.n, X)
__global__ __launch_bounds__(256) void k_small_to_limbs(const ModC *__restrict__ mod, int N, const int *__restrict__ a,
                                                        const long long *__restrict__ m, u64 *__restrict__ out,
                                                        size_t out_x_stride, LimbSel sel) {
    const int slot = blockIdx.y, x = blockIdx.z;
    const ModC M = mod[sel.mod[slot]];
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const long long e = a ? (long long)a[(size_t)x * N + c] : 0;
    u64 r;
    if (m) {
        const long long mv = m[(size_t)x * N + c];
        const u64 rm = mv >= 0 ? reduce64((u64)mv, M) : negmod(reduce64((u64)(-(mv + 1)) + 1ull, M), M.q);
        const u64 re = e >= 0 ? (u64)e : M.q - (u64)(-e);
        r = addmod(rm, re, M.q);
    } else {
        r = e >= 0 ? (u64)e : M.q - (u64)(-e);
    }
    out[(size_t)x * out_x_stride + (size_t)slot * N + c] = r;
}
// evaluation-form automorphism on [n limbs][N]: out[j] = in[perm_g(j)]. grid (N/256, n)
__global__ __launch_bounds__(256) void k_automorph(int logN, const u64 *__restrict__ in, u64 *__restrict__ out, unsigned g) {
    const unsigned N = 1u << logN, co = blockIdx.x * 256 + threadIdx.x;
    const unsigned e = ((2u * (__brev(co) >> (32 - logN)) + 1u) * g) & (2u * N - 1u);
    const unsigned c = __brev((e - 1u) >> 1) >> (32 - logN);
    out[(size_t)blockIdx.y * N + co] = in[(size_t)blockIdx.y * N + c];
}
// o = a (*) b over [n limbs][N] with per-slot modulus. grid (N/256, n)
__global__ __launch_bounds__(256) void k_mul(const ModC *__restrict__ mod, int N, const u64 *__restrict__ a,
                                             const u64 *__restrict__ b, u64 *__restrict__ o, LimbSel sel) {
    const ModC M = mod[sel.mod[blockIdx.y]];
    const size_t i = (size_t)blockIdx.y * N + (size_t)blockIdx.x * 256 + threadIdx.x;
    o[i] = mulmod(a[i], b[i], M);
}
// public key: b[j] = e[j] - a[j] s[j]. grid (N/256, nQ)
__global__ __launch_bounds__(256) void k_pk_combine(const ModC *__restrict__ mod, int N, u64 *__restrict__ b,
                                                    const u64 *__restrict__ a, const u64 *__restrict__ s) {
    const ModC M = mod[blockIdx.y];
    const size_t i = (size_t)blockIdx.y * N + (size_t)blockIdx.x * 256 + threadIdx.x;
    b[i] = submod(b[i], mulmod(a[i], s[i], M), M.q);
}
// switching key: b[d][m] = e[d][m] - a[d][m] s_enc[m] + [m in digit d] P s_from[m].  key [dnum][2][nT][N], e [dnum][nT][N]
// grid (N/256, nT, dnum)
__global__ __launch_bounds__(256) void k_evk_combine(const ModC *__restrict__ mod, int N, int nT, int nQ, int alpha,
                                                     u64 *__restrict__ key, const u64 *__restrict__ e,
                                                     const u64 *__restrict__ s_enc, const u64 *__restrict__ s_from,
                                                     ScaleSel pmodq) {
    const int m = blockIdx.y, d = blockIdx.z;
    const ModC M = mod[m];
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64 a = key[(((size_t)d * 2 + 1) * nT + m) * N + c];
    u64 v = submod(e[((size_t)d * nT + m) * N + c], mulmod(a, s_enc[(size_t)m * N + c], M), M.q);
    if (m < nQ && m / alpha == d) v = addmod(v, mulmod(pmodq.s[m], s_from[(size_t)m * N + c], M), M.q);
    key[(((size_t)d * 2 + 0) * nT + m) * N + c] = v;
}
// c0 = b u + t0 ; c1 = a u + t1.  ct [X][2][nQ][N]; u,t0,t1 [X][nQ][N]; pk [2][nQ][N]. grid (N/256, nQ, X)
__global__ __launch_bounds__(256) void k_enc_combine(const ModC *__restrict__ mod, int N, int nQ, const u64 *__restrict__ pk,
                                                     const u64 *__restrict__ u, const u64 *__restrict__ t0,
                                                     const u64 *__restrict__ t1, u64 *__restrict__ ct) {
    const int j = blockIdx.y, x = blockIdx.z;
    const ModC M = mod[j];
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i = ((size_t)x * nQ + j) * N + c;
    const u64 uu = u[i];
    ct[(((size_t)x * 2 + 0) * nQ + j) * N + c] = addmod(mulmod(pk[(size_t)j * N + c], uu, M), t0[i], M.q);
    ct[(((size_t)x * 2 + 1) * nQ + j) * N + c] = addmod(mulmod(pk[((size_t)nQ + j) * N + c], uu, M), t1[i], M.q);
}
// decryption dot product on the first nu limbs: t[x][j] = c0 + s (c1 + s c2 ...). grid (N/256, nu, X)
__global__ __launch_bounds__(256) void k_dec_dot(const ModC *__restrict__ mod, int N, int npoly, int nl,
                                                 const u64 *__restrict__ ct, const u64 *__restrict__ s, u64 *__restrict__ t,
                                                 int nu) {
    const int j = blockIdx.y, x = blockIdx.z;
    const ModC M = mod[j];
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64 sv = s[(size_t)j * N + c];
    u64 acc = ct[(((size_t)x * npoly + (npoly - 1)) * nl + j) * N + c];
    for (int p = npoly - 2; p >= 0; p--) acc = addmod(mulmod(acc, sv, M), ct[(((size_t)x * npoly + p) * nl + j) * N + c], M.q);
    t[((size_t)x * nu + j) * N + c] = acc;
}

// ------------------------------------------------------------------------------------------------ canonical embedding
// one radix-2 stage of the slots->coefficients transform ("fftSpecialInv"); v [X][Nh] complex. grid (Nh/2/256, X)
__global__ __launch_bounds__(256) void k_fft_inv_stage(double2 *__restrict__ v, int Nh, int len, int M,
                                                       const unsigned *__restrict__ rot_group,
                                                       const double2 *__restrict__ ksi) {
    const int b = blockIdx.x * 256 + threadIdx.x;  // butterfly id < Nh/2
    const int lenh = len >> 1, lenq = len << 2, gap = M / lenq;
    const int j = b % lenh, i = (b / lenh) * len;
    double2 *p = v + (size_t)blockIdx.y * Nh;
    const int idx = (lenq - (int)(rot_group[j] % (unsigned)lenq)) * gap;
    const double2 w = ksi[idx];
    const double2 A = p[i + j], B = p[i + j + lenh];
    const double ur = A.x + B.x, ui = A.y + B.y, dr = A.x - B.x, di = A.y - B.y;
    const double m0 = dr * w.x, m1 = di * w.y, m2 = dr * w.y, m3 = di * w.x;
    p[i + j] = make_double2(ur, ui);
    p[i + j + lenh] = make_double2(m0 - m1, m2 + m3);
}
// one stage of the coefficients->slots transform ("fftSpecial")
__global__ __launch_bounds__(256) void k_fft_fwd_stage(double2 *__restrict__ v, int Nh, int len, int M,
                                                       const unsigned *__restrict__ rot_group,
                                                       const double2 *__restrict__ ksi) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int lenh = len >> 1, lenq = len << 2, gap = M / lenq;
    const int j = b % lenh, i = (b / lenh) * len;
    double2 *p = v + (size_t)blockIdx.y * Nh;
    const int idx = (int)(rot_group[j] % (unsigned)lenq) * gap;
    const double2 w = ksi[idx];
    const double2 A = p[i + j], B = p[i + j + lenh];
    const double m0 = B.x * w.x, m1 = B.y * w.y, m2 = B.x * w.y, m3 = B.y * w.x;
    const double tr = m0 - m1, ti = m2 + m3;
    p[i + j] = make_double2(A.x + tr, A.y + ti);
    p[i + j + lenh] = make_double2(A.x - tr, A.y - ti);
}
// real slot values -> complex work array. grid (Nh/256, X)
__global__ __launch_bounds__(256) void k_slots_to_complex(const double *__restrict__ slots, double2 *__restrict__ v, int Nh) {
    const size_t i = (size_t)blockIdx.y * Nh + (size_t)blockIdx.x * 256 + threadIdx.x;
    v[i] = make_double2(slots[i], 0.0);
}
// after the last inverse stage: bit-reverse, scale by 1/Nh, then round(re*scale), round(im*scale) -> int64 coeffs [X][N]
__global__ __launch_bounds__(256) void k_encode_finish(const double2 *__restrict__ v, long long *__restrict__ coeffs, int Nh,
                                                       int logNh, double scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(__brev((unsigned)i) >> (32 - logNh));
    const double2 a = v[(size_t)blockIdx.y * Nh + r];
    const double inv = 1.0 / (double)Nh;
    const double re = a.x * inv, im = a.y * inv;
    long long *o = coeffs + (size_t)blockIdx.y * 2 * Nh;
    o[i] = __double2ll_rn(re * scale);
    o[i + Nh] = __double2ll_rn(im * scale);
}
// decode front: centred CRT of the first nu (1 or 2) limbs (coefficient form, t [X][nu][N]) -> value/scale, written in
// BIT-REVERSED complex order ready for the forward stages. grid (Nh/256, X)
__global__ __launch_bounds__(256) void k_decode_front(const ModC *__restrict__ mod, const u64 *__restrict__ t, int nu, int N,
                                                      int logNh, double scale, u64 q0inv_mod_q1, double2 *__restrict__ v) {
    const int Nh = N >> 1;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const u64 *p = t + (size_t)blockIdx.y * nu * N;
    double val[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int c = i + h * Nh;
        double r;
        if (nu >= 2) {
            const ModC M0 = mod[0], M1 = mod[1];
            const u64 r0 = p[c], r1 = p[(size_t)N + c];
            const u64 d = submod(r1, reduce64(r0, M1), M1.q);
            const u64 tt = mulmod(d, q0inv_mod_q1, M1);
            const u128 Q = (u128)M0.q * M1.q, x = (u128)r0 + (u128)M0.q * tt;
            const bool neg = x > (Q >> 1);
            const u128 mag = neg ? Q - x : x;
            const double vv = (double)(u64)(mag >> 64) * 18446744073709551616.0 + (double)(u64)mag;
            r = (neg ? -vv : vv) / scale;
        } else {
            const u64 q0 = mod[0].q, r0 = p[c];
            const bool neg = r0 > (q0 >> 1);
            const double vv = (double)(neg ? q0 - r0 : r0);
            r = (neg ? -vv : vv) / scale;
        }
        val[h] = r;
    }
    const int rr = (int)(__brev((unsigned)i) >> (32 - logNh));
    v[(size_t)blockIdx.y * Nh + rr] = make_double2(val[0], val[1]);
}
__global__ __launch_bounds__(256) void k_complex_real(const double2 *__restrict__ v, double *__restrict__ out, int Nh) {
    const size_t i = (size_t)blockIdx.y * Nh + (size_t)blockIdx.x * 256 + threadIdx.x;
    out[i] = v[i].x;
}

// ------------------------------------------------------------------------------------------------ enroller packing
// DiagonalEnroller's splitIntoSquareMatrices + preprocessToDiagonalForm + concatenateRows for ONE ciphertext group g:
// slots[i][j*dim + r] = db[(g*per + j)*dim + r][(r + i) mod dim] (0 beyond n).  dbg points at row g*per*dim.
// grid (slots/256, dim)
// babies > 0 (baby-step / giant-step form of the mat-vec): diagonal i is rotated by -babies * (i / babies) slots in the clear, i.e. output
// slot s takes what the plain layout has in slot s - babies * (i / babies) (mod Nh)
__global__ __launch_bounds__(256) void k_diag_pack(const double *__restrict__ dbg, long long rows_left, int dim, int Nh,
                                                   double *__restrict__ slots, int babies) {
    const int so = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    const int s = babies > 0 ? (so + Nh - (babies * (i / babies)) % Nh) % Nh : so;
    const int r = s % dim;
    const long long v = s;  // row inside this group = j*dim + r = s
    slots[(size_t)i * Nh + so] = v < rows_left ? dbg[(size_t)v * dim + (r + i) % dim] : 0.0;
}

// HersEnroller::serializeDBThread (/root/reference/src/enroller/enroller_hers.cpp:108-113): slots[j][k] = db[m*S + k][j].
// grid (Nh/256, dim)
__global__ __launch_bounds__(256) void k_hers_pack(const double *__restrict__ dbg, long long rows_left, int dim, int Nh,
                                                   double *__restrict__ slots) {
    const int k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    slots[(size_t)j * Nh + k] = k < rows_left ? dbg[(size_t)k * dim + j] : 0.0;
}
// HersReceiver::encryptQueryThread (/root/reference/src/receiver/receiver_hers.cpp:58-63): every slot = coordinate i
__global__ __launch_bounds__(256) void k_broadcast_rows(const double *__restrict__ vals, int Nh, double *__restrict__ slots) {
    slots[(size_t)blockIdx.y * Nh + blockIdx.x * 256 + threadIdx.x] = vals[blockIdx.y];
}

}  // namespace

namespace hc {

void sample_uniform(hipStream_t st, const ChaChaKey &key, const ModC *mod, int N, u64 sbase, u64 step_y, u64 step_z,
                    u64 *dst, size_t stride_y, size_t stride_z, const LimbSel &ysel, int nz) {
    const int blocks = (N / 4 + 255) / 256;
    hipLaunchKernelGGL(k_sample_uniform, dim3(blocks, ysel.n, nz), dim3(256), 0, st, key, mod, N, sbase, step_y, step_z, dst,
                       stride_y, stride_z, ysel);
}
void sample_ternary(hipStream_t st, const ChaChaKey &key, int N, u64 sbase, u64 step, int *dst, int X) {
    hipLaunchKernelGGL(k_sample_small<0>, dim3((N / 16 + 255) / 256, X), dim3(256), 0, st, key, N, sbase, step, dst);
}
void sample_gauss(hipStream_t st, const ChaChaKey &key, int N, u64 sbase, u64 step, int *dst, int X) {
    hipLaunchKernelGGL(k_sample_small<1>, dim3((N / 8 + 255) / 256, X), dim3(256), 0, st, key, N, sbase, step, dst);
}
void small_to_limbs(hipStream_t st, const ModC *mod, int N, const int *a, const long long *m, u64 *out,
                    size_t out_x_stride, int X, const LimbSel &sel) {
    hipLaunchKernelGGL(k_small_to_limbs, dim3(N / 256, sel.n, X), dim3(256), 0, st, mod, N, a, m, out, out_x_stride, sel);
}
void automorph(hipStream_t st, int logN, const u64 *in, u64 *out, unsigned g, int nlimbs) {
    hipLaunchKernelGGL(k_automorph, dim3((1 << logN) / 256, nlimbs), dim3(256), 0, st, logN, in, out, g);
}
void mul(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, const LimbSel &sel) {
    hipLaunchKernelGGL(k_mul, dim3(N / 256, sel.n), dim3(256), 0, st, mod, N, a, b, o, sel);
}
void pk_combine(hipStream_t st, const ModC *mod, int N, int nQ, u64 *b, const u64 *a, const u64 *s) {
    hipLaunchKernelGGL(k_pk_combine, dim3(N / 256, nQ), dim3(256), 0, st, mod, N, b, a, s);
}
void evk_combine(hipStream_t st, const ModC *mod, int N, int nT, int nQ, int alpha, int dnum, u64 *key, const u64 *e,
                 const u64 *s_enc, const u64 *s_from, const ScaleSel &pmodq) {
    hipLaunchKernelGGL(k_evk_combine, dim3(N / 256, nT, dnum), dim3(256), 0, st, mod, N, nT, nQ, alpha, key, e, s_enc, s_from,
                       pmodq);
}
void enc_combine(hipStream_t st, const ModC *mod, int N, int nQ, const u64 *pk, const u64 *u, const u64 *t0, const u64 *t1,
                 u64 *ct, int X) {
    hipLaunchKernelGGL(k_enc_combine, dim3(N / 256, nQ, X), dim3(256), 0, st, mod, N, nQ, pk, u, t0, t1, ct);
}
void dec_dot(hipStream_t st, const ModC *mod, int N, int npoly, int nl, const u64 *ct, const u64 *s, u64 *t, int nu, int X) {
    hipLaunchKernelGGL(k_dec_dot, dim3(N / 256, nu, X), dim3(256), 0, st, mod, N, npoly, nl, ct, s, t, nu);
}
void encode(hipStream_t st, const double *slots, double2 *work, long long *coeffs, int N, int X, double scale,
            const unsigned *rot_group, const double2 *ksi) {
    const int Nh = N / 2, M = 2 * N;
    int logNh = 0;
    while ((1 << logNh) < Nh) logNh++;
    hipLaunchKernelGGL(k_slots_to_complex, dim3(Nh / 256, X), dim3(256), 0, st, slots, work, Nh);
    for (int len = Nh; len >= 2; len >>= 1)
        hipLaunchKernelGGL(k_fft_inv_stage, dim3(Nh / 2 / 256, X), dim3(256), 0, st, work, Nh, len, M, rot_group, ksi);
    hipLaunchKernelGGL(k_encode_finish, dim3(Nh / 256, X), dim3(256), 0, st, work, coeffs, Nh, logNh, scale);
}
void decode(hipStream_t st, const ModC *mod, const u64 *t, int nu, int N, int X, double scale, u64 q0inv_mod_q1,
            double2 *work, double *out, const unsigned *rot_group, const double2 *ksi) {
    const int Nh = N / 2, M = 2 * N;
    int logNh = 0;
    while ((1 << logNh) < Nh) logNh++;
    hipLaunchKernelGGL(k_decode_front, dim3(Nh / 256, X), dim3(256), 0, st, mod, t, nu, N, logNh, scale, q0inv_mod_q1, work);
    for (int len = 2; len <= Nh; len <<= 1)
        hipLaunchKernelGGL(k_fft_fwd_stage, dim3(Nh / 2 / 256, X), dim3(256), 0, st, work, Nh, len, M, rot_group, ksi);
    hipLaunchKernelGGL(k_complex_real, dim3(Nh / 256, X), dim3(256), 0, st, work, out, Nh);
}
void hers_pack(hipStream_t st, const double *dbg, long long rows_left, int dim, int Nh, double *slots) {
    hipLaunchKernelGGL(k_hers_pack, dim3(Nh / 256, dim), dim3(256), 0, st, dbg, rows_left, dim, Nh, slots);
}
void broadcast_rows(hipStream_t st, const double *vals, int dim, int Nh, double *slots) {
    hipLaunchKernelGGL(k_broadcast_rows, dim3(Nh / 256, dim), dim3(256), 0, st, vals, Nh, slots);
}
void diag_pack(hipStream_t st, const double *dbg, long long rows_left, int dim, int Nh, double *slots, int babies) {
    hipLaunchKernelGGL(k_diag_pack, dim3(Nh / 256, dim), dim3(256), 0, st, dbg, rows_left, dim, Nh, slots, babies);
}

}  // namespace hc

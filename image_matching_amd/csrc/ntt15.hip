// image_matching_amd/csrc/ntt15.hip — negacyclic NTT / INTT for the production ring N = 2^15 on gfx950.
//
// N = 128 rows x 256 columns.  Forward = pass 1 (stages 0-6, row strides 64..1: a workgroup owns 32 adjacent columns of
// all 128 rows) then pass 2 (stages 7-14 inside 256-coefficient blocks: a workgroup owns 2048 consecutive
// coefficients); the inverse runs pass 2' then pass 1' with Gentleman-Sande butterflies.  Butterflies run in REGISTERS
// in radix-16 / radix-8 / radix-4 groups; LDS is used only to transpose between register phases (one exchange in
// pass 1, two in pass 2, padded to be bank-conflict free for ds_read/write_b64).  Harvey lazy reduction keeps values in
// [0,4q) (forward) / [0,2q) (inverse) and corrects once at the end, so results equal the strict transform bit for bit.
// Pass-1 phase-A twiddles are workgroup-uniform (scalar loads); the remaining pass-1 twiddles sit in 2 KiB of LDS;
// pass-2 twiddles are 16-byte (w, w') pair loads shared by the TWO polynomials a workgroup transforms together.
#include "kernels.h"

namespace {

// lazy Cooley-Tukey butterfly: a, b in [0,4q) -> [0,4q)
DEV void ct_bfly(u64 &a, u64 &b, const ulonglong2 W, const u64 q, const u64 q2) {
    const u64 u = a >= q2 ? a - q2 : a;
    const u64 hi = __umul64hi(b, W.y);
    const u64 t = b * W.x - hi * q;  // [0,2q)
    a = u + t;
    b = u - t + q2;
}
// lazy Gentleman-Sande butterfly: a, b in [0,2q) -> [0,2q)
DEV void gs_bfly(u64 &a, u64 &b, const ulonglong2 W, const u64 q, const u64 q2) {
    u64 s = a + b;
    s = s >= q2 ? s - q2 : s;
    const u64 d = a - b + q2;
    const u64 hi = __umul64hi(d, W.y);
    b = d * W.x - hi * q;
    a = s;
}
DEV u64 fix4q(u64 x, const u64 q, const u64 q2) {
    x = x >= q2 ? x - q2 : x;
    return x >= q ? x - q : x;
}

// ------------------------------------------------------------------------------------------------ pass 1 (strided)
// grid (8 column tiles, X*sel.n), 256 threads: col = t&31, g = t>>5.  Phase A rows g+8k (k<16), phase B rows 8h+l.
template <bool INV>
__global__ __launch_bounds__(256) void k_ntt15_p1(NttTables T, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so,
                                                  size_t dso, LimbSel sel, ScaleSel scale) {
    constexpr int N = 32768;
    __shared__ u64 lds[128 * 32];
    __shared__ ulonglong2 ltw[128];
    const int y = blockIdx.y, x = y / sel.n, slot = y - x * sel.n, m = sel.mod[slot];
    const u64 q = T.mod[m].q, q2 = 2 * q;
    const ulonglong2 *__restrict__ tw = (INV ? T.itwp : T.twp) + (size_t)m * N;
    const u64 *s = src + (size_t)x * so + (size_t)slot * N + blockIdx.x * 32;
    u64 *d = dst + (size_t)x * dso + (size_t)slot * N + blockIdx.x * 32;
    const int t = threadIdx.x, col = t & 31, g = t >> 5;
    if (t < 128) ltw[t] = tw[t];
    u64 v[16];
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = s[(size_t)(g + 8 * k) * 256 + col];
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const int h = 8 >> st;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!(k & h)) ct_bfly(v[k], v[k + h], tw[(1 << st) + (k >> (4 - st))], q, q2);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) lds[(g + 8 * k) * 32 + col] = v[k];
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int h = g + 8 * hh;
            u64 w[8];
#pragma unroll
            for (int l = 0; l < 8; l++) w[l] = lds[(8 * h + l) * 32 + col];
#pragma unroll
            for (int l = 0; l < 4; l++) ct_bfly(w[l], w[l + 4], ltw[16 + h], q, q2);
#pragma unroll
            for (int l = 0; l < 8; l++)
                if (!(l & 2)) ct_bfly(w[l], w[l + 2], ltw[32 + 2 * h + (l >> 2)], q, q2);
#pragma unroll
            for (int l = 0; l < 8; l += 2) ct_bfly(w[l], w[l + 1], ltw[64 + 4 * h + (l >> 1)], q, q2);
#pragma unroll
            for (int l = 0; l < 8; l++) d[(size_t)(8 * h + l) * 256 + col] = w[l];  // lazy [0,4q): pass 2 finishes
        }
    } else {
        __syncthreads();  // ltw
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int h = g + 8 * hh;
            u64 w[8];
#pragma unroll
            for (int l = 0; l < 8; l++) w[l] = s[(size_t)(8 * h + l) * 256 + col];
#pragma unroll
            for (int l = 0; l < 8; l += 2) gs_bfly(w[l], w[l + 1], ltw[64 + 4 * h + (l >> 1)], q, q2);
#pragma unroll
            for (int l = 0; l < 8; l++)
                if (!(l & 2)) gs_bfly(w[l], w[l + 2], ltw[32 + 2 * h + (l >> 2)], q, q2);
#pragma unroll
            for (int l = 0; l < 4; l++) gs_bfly(w[l], w[l + 4], ltw[16 + h], q, q2);
#pragma unroll
            for (int l = 0; l < 8; l++) lds[(8 * h + l) * 32 + col] = w[l];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[(g + 8 * k) * 32 + col];
#pragma unroll
        for (int st = 3; st >= 0; st--) {
            const int h = 8 >> st;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (!(k & h)) gs_bfly(v[k], v[k + h], tw[(1 << st) + (k >> (4 - st))], q, q2);
        }
        const u64 sc = scale.s[slot], scs = scale.s_sh[slot];
#pragma unroll
        for (int k = 0; k < 16; k++) d[(size_t)(g + 8 * k) * 256 + col] = mulmod_shoup(v[k], sc, scs, q);
    }
}

// ------------------------------------------------------------------------------------------------ pass 2 (contiguous)
// grid (16 chunks of 2048, pairs), 256 threads: blk = t>>5, w = t&31.  NP polynomials (1 or 2, same modulus) share every
// twiddle load.  LDS rows of 32 coefficients are padded to 36 so phase B's (a, b) reads hit 64 distinct banks.
template <bool INV, int NP>
__global__ __launch_bounds__(256) void k_ntt15_p2(NttTables T, const u64 *__restrict__ src, u64 *__restrict__ dst, size_t so,
                                                  size_t dso, LimbSel sel) {
    constexpr int N = 32768, LROW = 36, LBLK = 8 * LROW;
    __shared__ u64 lds[NP][8 * LBLK];
    const int y = blockIdx.y;
    const int xp = y / sel.n, slot = y - xp * sel.n, m = sel.mod[slot];
    const u64 q = T.mod[m].q, q2 = 2 * q;
    const ulonglong2 *__restrict__ tw = (INV ? T.itwp : T.twp) + (size_t)m * N;
    const int B0 = blockIdx.x * 2048, t = threadIdx.x, blk = t >> 5, w = t & 31;
    const int bg = (B0 >> 8) + blk;
    const int a = w >> 2, b = w & 3;
    const u64 *s[NP];
    u64 *d[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        s[p] = src + (size_t)(xp * NP + p) * so + (size_t)slot * N + B0;
        d[p] = dst + (size_t)(xp * NP + p) * dso + (size_t)slot * N + B0;
    }
    u64 v[NP][8];
    if (!INV) {
        // phase A: coefficients blk*256 + 32k + w ; stages 7,8,9 (strides 128, 64, 32)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int k = 0; k < 8; k++) v[p][k] = s[p][blk * 256 + 32 * k + w];
        {
            const ulonglong2 W7 = tw[128 + bg];
            const ulonglong2 W8a = tw[256 + 2 * bg], W8b = tw[256 + 2 * bg + 1];
            ulonglong2 W9[4];
#pragma unroll
            for (int i = 0; i < 4; i++) W9[i] = tw[512 + 4 * bg + i];
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 4; k++) ct_bfly(v[p][k], v[p][k + 4], W7, q, q2);
                ct_bfly(v[p][0], v[p][2], W8a, q, q2);
                ct_bfly(v[p][1], v[p][3], W8a, q, q2);
                ct_bfly(v[p][4], v[p][6], W8b, q, q2);
                ct_bfly(v[p][5], v[p][7], W8b, q, q2);
#pragma unroll
                for (int k = 0; k < 8; k += 2) ct_bfly(v[p][k], v[p][k + 1], W9[k >> 1], q, q2);
#pragma unroll
                for (int k = 0; k < 8; k++) lds[p][blk * LBLK + k * LROW + w] = v[p][k];
            }
        }
        __syncthreads();
        // phase B: coefficients blk*256 + 32a + 4k + b ; stages 10,11,12 (strides 16, 8, 4)
        {
            const int ib = 8 * bg + a;
            const ulonglong2 W10 = tw[1024 + ib];
            const ulonglong2 W11a = tw[2048 + 2 * ib], W11b = tw[2048 + 2 * ib + 1];
            ulonglong2 W12[4];
#pragma unroll
            for (int i = 0; i < 4; i++) W12[i] = tw[4096 + 4 * ib + i];
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[p][k] = lds[p][blk * LBLK + a * LROW + 4 * k + b];
#pragma unroll
                for (int k = 0; k < 4; k++) ct_bfly(v[p][k], v[p][k + 4], W10, q, q2);
                ct_bfly(v[p][0], v[p][2], W11a, q, q2);
                ct_bfly(v[p][1], v[p][3], W11a, q, q2);
                ct_bfly(v[p][4], v[p][6], W11b, q, q2);
                ct_bfly(v[p][5], v[p][7], W11b, q, q2);
#pragma unroll
                for (int k = 0; k < 8; k += 2) ct_bfly(v[p][k], v[p][k + 1], W12[k >> 1], q, q2);
#pragma unroll
                for (int k = 0; k < 8; k++) lds[p][blk * LBLK + a * LROW + 4 * k + b] = v[p][k];
            }
        }
        __syncthreads();
        // phase C: two groups of 4 consecutive coefficients e = 4t + 1024*hh ; stages 13, 14 (strides 2, 1)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int e = 4 * t + 1024 * hh, u = e & 255, la = (e >> 8) * LBLK + (u >> 5) * LROW + (u & 31);
            const int gi = (B0 + e) >> 2;
            const ulonglong2 W13 = tw[8192 + gi], W14a = tw[16384 + 2 * gi], W14b = tw[16384 + 2 * gi + 1];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                u64 c0 = lds[p][la], c1 = lds[p][la + 1], c2 = lds[p][la + 2], c3 = lds[p][la + 3];
                ct_bfly(c0, c2, W13, q, q2);
                ct_bfly(c1, c3, W13, q, q2);
                ct_bfly(c0, c1, W14a, q, q2);
                ct_bfly(c2, c3, W14b, q, q2);
                ulonglong2 o0, o1;
                o0.x = fix4q(c0, q, q2); o0.y = fix4q(c1, q, q2);
                o1.x = fix4q(c2, q, q2); o1.y = fix4q(c3, q, q2);
                *reinterpret_cast<ulonglong2 *>(d[p] + e) = o0;
                *reinterpret_cast<ulonglong2 *>(d[p] + e + 2) = o1;
            }
        }
    } else {
        // phase C': strides 1, 2
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int e = 4 * t + 1024 * hh, u = e & 255, la = (e >> 8) * LBLK + (u >> 5) * LROW + (u & 31);
            const int gi = (B0 + e) >> 2;
            const ulonglong2 W13 = tw[8192 + gi], W14a = tw[16384 + 2 * gi], W14b = tw[16384 + 2 * gi + 1];
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const ulonglong2 i0 = *reinterpret_cast<const ulonglong2 *>(s[p] + e);
                const ulonglong2 i1 = *reinterpret_cast<const ulonglong2 *>(s[p] + e + 2);
                u64 c0 = i0.x, c1 = i0.y, c2 = i1.x, c3 = i1.y;
                gs_bfly(c0, c1, W14a, q, q2);
                gs_bfly(c2, c3, W14b, q, q2);
                gs_bfly(c0, c2, W13, q, q2);
                gs_bfly(c1, c3, W13, q, q2);
                lds[p][la] = c0; lds[p][la + 1] = c1; lds[p][la + 2] = c2; lds[p][la + 3] = c3;
            }
        }
        __syncthreads();
        // phase B': strides 4, 8, 16
        {
            const int ib = 8 * bg + a;
            const ulonglong2 W10 = tw[1024 + ib];
            const ulonglong2 W11a = tw[2048 + 2 * ib], W11b = tw[2048 + 2 * ib + 1];
            ulonglong2 W12[4];
#pragma unroll
            for (int i = 0; i < 4; i++) W12[i] = tw[4096 + 4 * ib + i];
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[p][k] = lds[p][blk * LBLK + a * LROW + 4 * k + b];
#pragma unroll
                for (int k = 0; k < 8; k += 2) gs_bfly(v[p][k], v[p][k + 1], W12[k >> 1], q, q2);
                gs_bfly(v[p][0], v[p][2], W11a, q, q2);
                gs_bfly(v[p][1], v[p][3], W11a, q, q2);
                gs_bfly(v[p][4], v[p][6], W11b, q, q2);
                gs_bfly(v[p][5], v[p][7], W11b, q, q2);
#pragma unroll
                for (int k = 0; k < 4; k++) gs_bfly(v[p][k], v[p][k + 4], W10, q, q2);
#pragma unroll
                for (int k = 0; k < 8; k++) lds[p][blk * LBLK + a * LROW + 4 * k + b] = v[p][k];
            }
        }
        __syncthreads();
        // phase A': strides 32, 64, 128
        {
            const ulonglong2 W7 = tw[128 + bg];
            const ulonglong2 W8a = tw[256 + 2 * bg], W8b = tw[256 + 2 * bg + 1];
            ulonglong2 W9[4];
#pragma unroll
            for (int i = 0; i < 4; i++) W9[i] = tw[512 + 4 * bg + i];
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[p][k] = lds[p][blk * LBLK + k * LROW + w];
#pragma unroll
                for (int k = 0; k < 8; k += 2) gs_bfly(v[p][k], v[p][k + 1], W9[k >> 1], q, q2);
                gs_bfly(v[p][0], v[p][2], W8a, q, q2);
                gs_bfly(v[p][1], v[p][3], W8a, q, q2);
                gs_bfly(v[p][4], v[p][6], W8b, q, q2);
                gs_bfly(v[p][5], v[p][7], W8b, q, q2);
#pragma unroll
                for (int k = 0; k < 4; k++) gs_bfly(v[p][k], v[p][k + 4], W7, q, q2);
#pragma unroll
                for (int k = 0; k < 8; k++) d[p][blk * 256 + 32 * k + w] = v[p][k];  // lazy [0,2q): pass 1' finishes
            }
        }
    }
}

}  // namespace

namespace hk {

// element (x, slot) at base + x*outer + slot*N.  When X is even, pass 2 transforms polynomials 2x', 2x'+1 together.
void ntt15_forward(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                   const LimbSel &sel) {
    ScaleSel dummy = {};
    hipLaunchKernelGGL(k_ntt15_p1<false>, dim3(8, X * sel.n), dim3(256), 0, st, T, src, dst, so, dso, sel, dummy);
    if (X % 2 == 0)
        hipLaunchKernelGGL((k_ntt15_p2<false, 2>), dim3(16, (X / 2) * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel);
    else
        hipLaunchKernelGGL((k_ntt15_p2<false, 1>), dim3(16, X * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel);
}
void ntt15_inverse(hipStream_t st, const NttTables &T, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                   const LimbSel &sel, const ScaleSel &scale) {
    if (X % 2 == 0)
        hipLaunchKernelGGL((k_ntt15_p2<true, 2>), dim3(16, (X / 2) * sel.n), dim3(256), 0, st, T, src, dst, so, dso, sel);
    else
        hipLaunchKernelGGL((k_ntt15_p2<true, 1>), dim3(16, X * sel.n), dim3(256), 0, st, T, src, dst, so, dso, sel);
    hipLaunchKernelGGL(k_ntt15_p1<true>, dim3(8, X * sel.n), dim3(256), 0, st, T, dst, dst, dso, dso, sel, scale);
}

}  // namespace hk

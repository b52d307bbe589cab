// image_matching_amd/csrc/kernels.hip — hand-written gfx950 kernels of the HyDia sender hot path.
//
// What each kernel replaces (reference call sites; the arithmetic itself is OpenFHE's, un-vendored):
//   k_ntt_*               every NTT/INTT inside EvalFastRotation / Relinearize / Rescale
//                         (/root/reference/src/sender/sender_diag.cpp:22-26, :79-80)
//   k_base_convert        ModUp / ModDown fast base conversion of hybrid key switching (same call sites)
//   k_inner_product       <digits, evk> of EvalFastRotation (sender_diag.cpp:25) and RelinearizeInPlace (:79)
//   k_moddown_combine     ModDown's (acc - conv) / P, + c0, + the evaluation-form automorphism of EvalFastRotation
//   k_hydia_tensor        512 x EvalMultNoRelin + 511 x EvalAddInPlace per block (sender_diag.cpp:70-77, :93)
//   k_rescale_*           RescaleInPlace (sender_diag.cpp:80)
//   k_tensor, k_lincomb*  ct x ct products and Chebyshev/f4 leaves of chebyshevCompare (src/openFHE_wrapper.cpp:143-185)
//   k_batch_sum           HERS: sum of the per-dimension products (src/sender/sender_hers.cpp:60-87)
// No MFMA: this is 64-bit integer modular arithmetic.  HBM-streaming kernels read 16 B per lane (1 KiB per wave
// instruction) of one limb, so modulus constants are wave-uniform.
#include <stdexcept>

#include "kernels.h"
#include "ntt_arith.h"  // FpA: exact FP64 products for the limbs below 2^47

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>

namespace {

DEV unsigned brev_n(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

// ------------------------------------------------------------------------------------------------ NTT
// N = 2^logN = R * 256.  Forward = strided pass (first logN-8 stages, stride >= 256, a workgroup owns 32 adjacent
// columns x all R rows in LDS) then contiguous pass (last 8 stages inside 256-blocks; a workgroup owns 2048
// consecutive coefficients).  Inverse runs the two passes in the opposite order with Gentleman-Sande butterflies.
template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_strided(NttTables T, int logN, const u64 *__restrict__ src,
                                                     u64 *__restrict__ dst, size_t so, size_t dso, LimbSel sel,
                                                     ScaleSel scale) {
    extern __shared__ u64 lds[];
    const int N = 1 << logN, logR = logN - 8, R = 1 << logR;
    const int y = blockIdx.y, x = y / sel.n, slot = y - x * sel.n, m = sel.mod[slot];
    const u64 q = T.mod[m].q;
    const u64 *s = src + (size_t)x * so + (size_t)slot * N;
    u64 *d = dst + (size_t)x * dso + (size_t)slot * N;
    const int c0 = blockIdx.x * 32, tid = threadIdx.x;
    for (int e = tid; e < R * 32; e += 256) lds[e] = s[(size_t)(e >> 5) * 256 + c0 + (e & 31)];
    __syncthreads();
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)m * N;
    const u64 *tws = (INV ? T.itw_sh : T.tw_sh) + (size_t)m * N;
    for (int st = 0; st < logR; st++) {
        const int lt = INV ? st : logR - 1 - st;           // log2 of the row stride
        const int base = INV ? (R >> (st + 1)) : (1 << st);  // twiddle block of this stage
        for (int e = tid; e < (R >> 1) * 32; e += 256) {
            const int c = e & 31, k = e >> 5;
            const int i = k >> lt, o = k & ((1 << lt) - 1);
            const int r0 = (i << (lt + 1)) + o, r1 = r0 + (1 << lt);
            const u64 W = tw[base + i], Ws = tws[base + i];
            const u64 U = lds[r0 * 32 + c], V = lds[r1 * 32 + c];
            if (!INV) {
                const u64 Vw = mulmod_shoup(V, W, Ws, q);
                lds[r0 * 32 + c] = addmod(U, Vw, q);
                lds[r1 * 32 + c] = submod(U, Vw, q);
            } else {
                lds[r0 * 32 + c] = addmod(U, V, q);
                lds[r1 * 32 + c] = mulmod_shoup(submod(U, V, q), W, Ws, q);
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < R * 32; e += 256) {
        u64 v = lds[e];
        if (INV) v = mulmod_shoup(v, scale.s[slot], scale.s_sh[slot], q);
        d[(size_t)(e >> 5) * 256 + c0 + (e & 31)] = v;
    }
}

template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_contig(NttTables T, int logN, const u64 *__restrict__ src,
                                                    u64 *__restrict__ dst, size_t so, size_t dso, LimbSel sel) {
    __shared__ u64 lds[2048];
    const int N = 1 << logN;
    const int y = blockIdx.y, x = y / sel.n, slot = y - x * sel.n, m = sel.mod[slot];
    const u64 q = T.mod[m].q;
    const int B0 = blockIdx.x * 2048, tid = threadIdx.x;
    const u64 *s = src + (size_t)x * so + (size_t)slot * N + B0;
    u64 *d = dst + (size_t)x * dso + (size_t)slot * N + B0;
    for (int k = 0; k < 4; k++) {
        const int idx = 2 * tid + 512 * k;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(s + idx);
        lds[idx] = v.x;
        lds[idx + 1] = v.y;
    }
    __syncthreads();
    const u64 *tw = (INV ? T.itw : T.tw) + (size_t)m * N;
    const u64 *tws = (INV ? T.itw_sh : T.tw_sh) + (size_t)m * N;
    for (int st = 0; st < 8; st++) {
        const int lt = INV ? st : 7 - st;  // log2 of the stride inside the chunk
        const int base = (INV ? (N >> (lt + 1)) : (N >> (lt + 1))) + (B0 >> (lt + 1));
        // forward stage with stride t uses block mm = N/(2t); inverse stage with stride t uses block h = N/(2t)
        for (int e = tid; e < 1024; e += 256) {
            const int i = e >> lt, o = e & ((1 << lt) - 1);
            const int l0 = (i << (lt + 1)) + o, l1 = l0 + (1 << lt);
            const u64 W = tw[base + i], Ws = tws[base + i];
            const u64 U = lds[l0], V = lds[l1];
            if (!INV) {
                const u64 Vw = mulmod_shoup(V, W, Ws, q);
                lds[l0] = addmod(U, Vw, q);
                lds[l1] = submod(U, Vw, q);
            } else {
                lds[l0] = addmod(U, V, q);
                lds[l1] = mulmod_shoup(submod(U, V, q), W, Ws, q);
            }
        }
        __syncthreads();
    }
    for (int k = 0; k < 4; k++) {
        const int idx = 2 * tid + 512 * k;
        ulonglong2 v;
        v.x = lds[idx];
        v.y = lds[idx + 1];
        *reinterpret_cast<ulonglong2 *>(d + idx) = v;
    }
}

// ------------------------------------------------------------------------------------------------ element-wise
// grid: (N/512, XP*sel.n) over XP polynomials; each thread 2 coefficients (16 B).  Operands may be limb-strided views
// (a dropped ciphertext keeps its allocation): polynomial xp of operand t starts at xp * t_ls * N.
template <int OP>
__global__ __launch_bounds__(256) void k_addsub(const ModC *__restrict__ mod, int N, const u64 *a, const u64 *b, u64 *o,
                                                LimbSel sel, int a_ls, int b_ls, int o_ls) {  // a, b, o may alias
    const int y = blockIdx.y, xp = y / sel.n, slot = y - xp * sel.n;
    const u64 q = mod[sel.mod[slot]].q;
    const size_t i = (size_t)slot * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 va = *reinterpret_cast<const ulonglong2 *>(a + (size_t)xp * a_ls * N + i);
    const ulonglong2 vb = *reinterpret_cast<const ulonglong2 *>(b + (size_t)xp * b_ls * N + i);
    ulonglong2 r;
    // OP 2: plain integer sum (the cross-shard membership reduction keeps residues unreduced until k_mod_reduce)
    r.x = OP == 0 ? addmod(va.x, vb.x, q) : OP == 1 ? submod(va.x, vb.x, q) : va.x + vb.x;
    r.y = OP == 0 ? addmod(va.y, vb.y, q) : OP == 1 ? submod(va.y, vb.y, q) : va.y + vb.y;
    *reinterpret_cast<ulonglong2 *>(o + (size_t)xp * o_ls * N + i) = r;
}
// any 64-bit value -> canonical residue of its limb (after an integer all-reduce of at most 16 residues < 2^60)
__global__ __launch_bounds__(256) void k_mod_reduce(const ModC *__restrict__ mod, int N, u64 *a, LimbSel sel, int a_ls) {
    const int y = blockIdx.y, xp = y / sel.n, slot = y - xp * sel.n;
    const ModC M = mod[sel.mod[slot]];
    u64 *p = a + (size_t)xp * a_ls * N + (size_t)slot * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
    *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(reduce64(v.x, M), reduce64(v.y, M));
}
__global__ __launch_bounds__(256) void k_mul_scalar(const ModC *__restrict__ mod, int N, const u64 *__restrict__ a,
                                                    u64 *__restrict__ o, LimbSel sel, ScaleSel c, int a_ls, int o_ls) {
    const int y = blockIdx.y, xp = y / sel.n, slot = y - xp * sel.n;
    const u64 q = mod[sel.mod[slot]].q;
    const size_t i = (size_t)slot * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 va = *reinterpret_cast<const ulonglong2 *>(a + (size_t)xp * a_ls * N + i);
    ulonglong2 r;
    r.x = mulmod_shoup(va.x, c.s[slot], c.s_sh[slot], q);
    r.y = mulmod_shoup(va.y, c.s[slot], c.s_sh[slot], q);
    *reinterpret_cast<ulonglong2 *>(o + (size_t)xp * o_ls * N + i) = r;
}
// grid (N/512, nl, X*npoly)
__global__ __launch_bounds__(256) void k_lincomb(const ModC *__restrict__ mod, int N, LinComb lc, u64 *__restrict__ o,
                                                 int npoly, int nl) {
    const int j = blockIdx.y, xp = blockIdx.z, p = xp % npoly;
    const ModC M = mod[j];
    const size_t i = (size_t)j * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    u64 ax = 0, ay = 0;  // each term < q < 2^60 and at most 8 terms (+ c0): no overflow
#pragma unroll
    for (int t = 0; t < HY_LC_TERMS; t++)
        if (t < lc.nterms) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(lc.src[t] + (size_t)xp * lc.ls[t] * N + i);
            ax += mulmod_shoup(v.x, lc.c[t][j], lc.cs[t][j], M.q);
            ay += mulmod_shoup(v.y, lc.c[t][j], lc.cs[t][j], M.q);
        }
    if (p == 0) {
        ax += lc.c0[j];
        ay += lc.c0[j];
    }
    ulonglong2 r;
    r.x = reduce64(ax, M);
    r.y = reduce64(ay, M);
    *reinterpret_cast<ulonglong2 *>(o + (size_t)xp * nl * N + i) = r;
}
// grid (N/512, nl, X*npoly): the terms are loaded once into registers, then K outputs with wave-uniform constants
// (128-bit lazy sums: 8 terms * 2^120 + c0 < 2^124)
__global__ __launch_bounds__(256) void k_lincomb_multi(const ModC *__restrict__ mod, int N, LinCombMulti lc, u64 *__restrict__ o,
                                                       int XP, int npoly, int nl) {
    const int j = blockIdx.y, xp = blockIdx.z, p = xp % npoly;
    const ModC M = mod[j];
    const size_t i = (size_t)j * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v[HY_LC_TERMS];
#pragma unroll
    for (int t = 0; t < HY_LC_TERMS; t++)
        v[t] = t < lc.nterms ? *reinterpret_cast<const ulonglong2 *>(lc.src[t] + (size_t)xp * lc.ls[t] * N + i) : make_ulonglong2(0, 0);
    if (lc.fp && M.ks + 2 <= 47) {
        // limbs below 2^47 (round 4): exact FP64 products, v c - rint(v c / q) q (six instructions, no carries) instead of 128-bit
        // multiply-accumulates and a Barrett reduction — the same canonical residues (|sum| < 8 x 0.8 q + q, exact in a double)
        const FpA ar(M);
        double vx[HY_LC_TERMS], vy[HY_LC_TERMS];
#pragma unroll
        for (int t = 0; t < HY_LC_TERMS; t++) {
            vx[t] = FpA::u2d(v[t].x);
            vy[t] = FpA::u2d(v[t].y);
        }
        for (int k = 0; k < lc.K; k++) {
            const u64 *tb = lc.tab + (size_t)k * HY_LCM_BLOCK;
            double ax = 0, ay = 0;
#pragma unroll
            for (int t = 0; t < HY_LC_TERMS; t++)
                if (t < lc.nterms) {
                    const FpA::TW W = ar.tw8(FpA::u2d(tb[t * HY_LC_LIMBS + j]));
                    ax += ar.mulmod(vx[t], W);
                    ay += ar.mulmod(vy[t], W);
                }
            if (p == 0) {
                const double c0 = FpA::u2d(tb[HY_LC_TERMS * HY_LC_LIMBS + j]);
                ax += c0;
                ay += c0;
            }
            *reinterpret_cast<ulonglong2 *>(o + ((size_t)k * XP + xp) * nl * N + i) = make_ulonglong2(ar.fin_fwd(ax), ar.fin_fwd(ay));
        }
        return;
    }
    for (int k = 0; k < lc.K; k++) {
        const u64 *tb = lc.tab + (size_t)k * HY_LCM_BLOCK;
        u128 ax = 0, ay = 0;
#pragma unroll
        for (int t = 0; t < HY_LC_TERMS; t++)
            if (t < lc.nterms) {
                const u64 c = tb[t * HY_LC_LIMBS + j];
                ax += (u128)v[t].x * c;
                ay += (u128)v[t].y * c;
            }
        if (p == 0) {
            const u64 c0 = tb[HY_LC_TERMS * HY_LC_LIMBS + j];
            ax += c0;
            ay += c0;
        }
        ulonglong2 r;
        r.x = reduce_lazy(ax, M, lc.nterms + 1);
        r.y = reduce_lazy(ay, M, lc.nterms + 1);
        *reinterpret_cast<ulonglong2 *>(o + ((size_t)k * XP + xp) * nl * N + i) = r;
    }
}
// grid (N/512, nl, npoly): serial sum over the batch with 128-bit accumulators (X * 2^60 fits)
// output m (grid.z = npoly * nout) = sum over x of ciphertext m + x * stride of `in` (stride 1, nout 1: a plain batch sum)
__global__ __launch_bounds__(256) void k_batch_sum(const ModC *__restrict__ mod, int N, const u64 *__restrict__ in,
                                                   u64 *__restrict__ o, int X, int npoly, int nl, int stride) {
    const int j = blockIdx.y, p = blockIdx.z % npoly, m = blockIdx.z / npoly;
    const ModC M = mod[j];
    const size_t i = ((size_t)p * nl + j) * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t cs = (size_t)npoly * nl * N, xs = cs * (size_t)stride;
    u128 ax = 0, ay = 0;
    for (int x = 0; x < X; x++) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(in + (size_t)m * cs + (size_t)x * xs + i);
        ax += v.x;
        ay += v.y;
    }
    ulonglong2 r;
    r.x = reduce128(ax, M);
    r.y = reduce128(ay, M);
    *reinterpret_cast<ulonglong2 *>(o + (size_t)m * cs + i) = r;
}
__global__ __launch_bounds__(256) void k_add_scalar(const ModC *__restrict__ mod, int N, u64 *__restrict__ a,
                                                    size_t outer, LimbSel sel, ScaleSel c) {
    const int y = blockIdx.y, x = y / sel.n, slot = y - x * sel.n;
    const u64 q = mod[sel.mod[slot]].q;
    u64 *p = a + (size_t)x * outer + (size_t)slot * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v = *reinterpret_cast<ulonglong2 *>(p);
    v.x = addmod(v.x, c.s[slot], q);
    v.y = addmod(v.y, c.s[slot], q);
    *reinterpret_cast<ulonglong2 *>(p) = v;
}
__global__ __launch_bounds__(256) void k_copy_limbs(int N, const u64 *__restrict__ src, u64 *__restrict__ dst,
                                                    size_t so, size_t dso, int nlimbs) {
    const int y = blockIdx.y, x = y / nlimbs, slot = y - x * nlimbs;
    const size_t i = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    *reinterpret_cast<ulonglong2 *>(dst + (size_t)x * dso + (size_t)slot * N + i) =
        *reinterpret_cast<const ulonglong2 *>(src + (size_t)x * so + (size_t)slot * N + i);
}
// EvalMultNoRelin on X pairs: grid (N/512, nl, X); inputs may be limb-strided views, output compact.
// SUB: d0 -= kap_j*c0, d1 -= kap_j*c1 for a 2-component c (the comparator's  2ab - K*c  with kap = K/2 mod q_j, applied
// ahead of the doubling relinearisation)
template <bool SUB>
__global__ __launch_bounds__(256) void k_tensor(const ModC *__restrict__ mod, int N, const u64 *__restrict__ a,
                                                const u64 *__restrict__ b, u64 *__restrict__ o, int nl, int a_ls, int b_ls,
                                                const u64 *__restrict__ c, int c_ls, ScaleSel kap) {
    const int j = blockIdx.y, x = blockIdx.z;
    const ModC M = mod[j];
    const size_t i = (size_t)j * N + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const size_t pa = (size_t)x * 2 * a_ls * N + i, pb = (size_t)x * 2 * b_ls * N + i, ps = (size_t)nl * N;
    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(a + pa), a1 = *reinterpret_cast<const ulonglong2 *>(a + pa + (size_t)a_ls * N);
    const ulonglong2 b0 = *reinterpret_cast<const ulonglong2 *>(b + pb), b1 = *reinterpret_cast<const ulonglong2 *>(b + pb + (size_t)b_ls * N);
    ulonglong2 d0, d1, d2;
    d0.x = mulmod(a0.x, b0.x, M);
    d0.y = mulmod(a0.y, b0.y, M);
    d1.x = reduce128k((u128)a0.x * b1.x + (u128)a1.x * b0.x, M);
    d1.y = reduce128k((u128)a0.y * b1.y + (u128)a1.y * b0.y, M);
    d2.x = mulmod(a1.x, b1.x, M);
    d2.y = mulmod(a1.y, b1.y, M);
    if (SUB) {
        const size_t pc = (size_t)x * 2 * c_ls * N + i;
        const ulonglong2 c0 = *reinterpret_cast<const ulonglong2 *>(c + pc), c1 = *reinterpret_cast<const ulonglong2 *>(c + pc + (size_t)c_ls * N);
        const u64 k = kap.s[j], ks = kap.s_sh[j];
        d0.x = submod(d0.x, mulmod_shoup(c0.x, k, ks, M.q), M.q);
        d0.y = submod(d0.y, mulmod_shoup(c0.y, k, ks, M.q), M.q);
        d1.x = submod(d1.x, mulmod_shoup(c1.x, k, ks, M.q), M.q);
        d1.y = submod(d1.y, mulmod_shoup(c1.y, k, ks, M.q), M.q);
    }
    const size_t po = (size_t)x * 3 * ps + i;
    *reinterpret_cast<ulonglong2 *>(o + po) = d0;
    *reinterpret_cast<ulonglong2 *>(o + po + ps) = d1;
    *reinterpret_cast<ulonglong2 *>(o + po + 2 * ps) = d2;
}

// ------------------------------------------------------------------------------------------------ key switching
// grid (N/512, X): each thread reads its ns source residues ONCE (2 coefficients, 16 B loads) and produces all nt
// targets — (ns + nt) limb-polys of traffic instead of nt*(ns + 1)
__global__ __launch_bounds__(256) void k_base_convert(const ModC *__restrict__ mod, int N, const u64 *__restrict__ y,
                                                      size_t yo, u64 *__restrict__ out, size_t oo, ConvTab tab,
                                                      LimbSel dsel, int tz) {
    // grid.z > 1 (few polynomials: a query's fixed-cost tail): targets [z*tz, (z+1)*tz) per slice, sources re-read from L2
    const int x = blockIdx.y;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v[HY_MAX_DIGIT];
#pragma unroll
    for (int s = 0; s < HY_MAX_DIGIT; s++)
        if (s < tab.ns) v[s] = *reinterpret_cast<const ulonglong2 *>(y + (size_t)x * yo + (size_t)s * N + c);
    const int t_lo = blockIdx.z * tz, t_hi = min(tab.nt, t_lo + tz);
    for (int t = t_lo; t < t_hi; t++) {
        if (t >= tab.skip_lo && t < tab.skip_hi) continue;
        const ModC M = mod[dsel.mod[t]];
        u128 ax = 0, ay = 0;
#pragma unroll
        for (int s = 0; s < HY_MAX_DIGIT; s++)
            if (s < tab.ns) {
                ax += (u128)v[s].x * tab.f[s][t];
                ay += (u128)v[s].y * tab.f[s][t];
            }
        // sources are residues of OTHER moduli (< 2^60), constants < q_t: up to four terms stay below 2^(k+62)
        ulonglong2 r;
        r.x = tab.ns <= 4 ? reduce128k(ax, M) : reduce128(ax, M);
        r.y = tab.ns <= 4 ? reduce128k(ay, M) : reduce128(ay, M);
        *reinterpret_cast<ulonglong2 *>(out + (size_t)x * oo + (size_t)t * N + c) = r;
    }
}
// ModUp of ALL digits of a key switch in one launch: grid (N/512, X, nd * slices).  Digit d = z / slices converts its own limbs
// (rows [skip_lo, skip_hi) of y [X][nl][N], coefficient form) into rows of out [X][nd][nE][N]; tabs[d] lives in device memory
__global__ __launch_bounds__(256) void k_base_convert_digits(const ModC *__restrict__ mod, int N, const u64 *__restrict__ y, size_t yo,
                                                             u64 *__restrict__ out, size_t oo, const ConvTab *__restrict__ tabs,
                                                             LimbSel dsel, int slices, int tz, int nE) {
    const int x = blockIdx.y, d = blockIdx.z / slices, zs = blockIdx.z - d * slices;
    const ConvTab &tab = tabs[d];
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const int ns = tab.ns, lo = tab.skip_lo, hi = tab.skip_hi;
    ulonglong2 v[HY_MAX_DIGIT];
#pragma unroll
    for (int s = 0; s < HY_MAX_DIGIT; s++)
        if (s < ns) v[s] = *reinterpret_cast<const ulonglong2 *>(y + (size_t)x * yo + (size_t)(lo + s) * N + c);
    u64 *o = out + (size_t)x * oo + (size_t)d * nE * N;
    const int t_lo = zs * tz, t_hi = min(tab.nt, t_lo + tz);
    for (int t = t_lo; t < t_hi; t++) {
        if (t >= lo && t < hi) continue;
        const ModC M = mod[dsel.mod[t]];
        u128 ax = 0, ay = 0;
#pragma unroll
        for (int s = 0; s < HY_MAX_DIGIT; s++)
            if (s < ns) {
                const u64 f = tab.f[s][t];
                ax += (u128)v[s].x * f;
                ay += (u128)v[s].y * f;
            }
        ulonglong2 r;
        r.x = ns <= 4 ? reduce128k(ax, M) : reduce128(ax, M);
        r.y = ns <= 4 ? reduce128k(ay, M) : reduce128(ay, M);
        *reinterpret_cast<ulonglong2 *>(o + (size_t)t * N + c) = r;
    }
}
// two consecutive residues of an 8-byte (PK = false) or 6-byte (PK = true) row; NT: non-temporal load
// grid (N/512, nE, X); 2 coefficients per thread, both key polys.  PK: keys[x] points at a packed key (see above)
template <bool PK>
__global__ __launch_bounds__(256) void k_inner_product(const ModC *__restrict__ mod, int N, const u64 *__restrict__ dig,
                                                       size_t dxs, int nd, const u64 *const *__restrict__ keys,
                                                       int same_key, int nT, u64 *__restrict__ acc, LimbSel esel,
                                                       const u64 *__restrict__ own, size_t own_xs, int alpha, int nl, int acc_rows,
                                                       int nQ, int dig_rows, int dig_t0) {
    // acc row t <-> modulus esel.mod[t] <-> digit row dig_t0 + t (dig_t0 > 0: only the special-prime limbs are accumulated here,
    // the Q limbs' inner product lives in the ModDown transform's epilogue — NttStore mode 5)
    const int t = blockIdx.y, x = blockIdx.z, nE = acc_rows, m = esel.mod[t];
    const ModC M = mod[m];
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const u64 *key = keys[same_key ? 0 : x];
    const bool six = PK && m > 0 && m < nQ;
    const unsigned char *kbytes = reinterpret_cast<const unsigned char *>(key) + (PK ? key_limb_offset(N, nQ, m) : 0) + c * (six ? 6 : 8);
    const size_t set_bytes = PK ? key_set_bytes(N, nQ, nT) : 0;
    u128 a0x = 0, a0y = 0, a1x = 0, a1y = 0;
    for (int d = 0; d < nd; d++) {
        // a digit's own limbs are the input itself (evaluation form): read them in place when the caller did not copy them
        const u64 *src = (own && t < nl && t / alpha == d) ? own + (size_t)x * own_xs + (size_t)t * N + c
                                                           : dig + (size_t)x * dxs + ((size_t)d * dig_rows + dig_t0 + t) * N + c;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src);
        ulonglong2 kb, ka;
        if (PK) {
            const unsigned char *pb = kbytes + (size_t)(d * 2) * set_bytes, *pa = pb + set_bytes;
            if (six) {
                kb = db_load2<true, false>(pb);
                ka = db_load2<true, false>(pa);
            } else {
                kb = db_load2<false, false>(pb);
                ka = db_load2<false, false>(pa);
            }
        } else {
            kb = *reinterpret_cast<const ulonglong2 *>(key + (((size_t)d * 2 + 0) * nT + m) * N + c);
            ka = *reinterpret_cast<const ulonglong2 *>(key + (((size_t)d * 2 + 1) * nT + m) * N + c);
        }
        a0x += (u128)v.x * kb.x;
        a0y += (u128)v.y * kb.y;
        a1x += (u128)v.x * ka.x;
        a1y += (u128)v.y * ka.y;
    }
    ulonglong2 r0, r1;
    r0.x = reduce_lazy(a0x, M, nd);
    r0.y = reduce_lazy(a0y, M, nd);
    r1.x = reduce_lazy(a1x, M, nd);
    r1.y = reduce_lazy(a1y, M, nd);
    *reinterpret_cast<ulonglong2 *>(acc + (((size_t)x * 2 + 0) * nE + t) * N + c) = r0;
    *reinterpret_cast<ulonglong2 *>(acc + (((size_t)x * 2 + 1) * nE + t) * N + c) = r1;
}
// [nd][2][nT][N] u64 -> packed key.  grid (N/512, nT, nd*2)
// premul: the Q-limb rows are stored multiplied by P^{-1} mod q_j (mul.s[j]) — loop A's fused ModDown epilogue then needs no
// multiplication at all: (acc - conv) P^{-1} = acc' - conv' with both operands pre-scaled (NttStore mode 5, LoopAIp::premul)
__global__ __launch_bounds__(256) void k_key_pack(const ModC *__restrict__ mod, int N, int nQ, int nT, const u64 *__restrict__ key,
                                                  unsigned char *__restrict__ out, int premul, ScaleSel mul) {
    const int m = blockIdx.y, dp = blockIdx.z;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(key + ((size_t)dp * nT + m) * N + c);
    if (premul && m < nQ) {
        const u64 q = mod[m].q;
        v.x = mulmod_shoup(v.x, mul.s[m], mul.s_sh[m], q);
        v.y = mulmod_shoup(v.y, mul.s[m], mul.s_sh[m], q);
    }
    const bool six = m > 0 && m < nQ;
    unsigned char *d = out + (size_t)dp * key_set_bytes(N, nQ, nT) + key_limb_offset(N, nQ, m) + c * (six ? 6 : 8);
    typedef unsigned int u3 __attribute__((ext_vector_type(3), aligned(4)));
    if (six) {
        u3 w;
        w.x = (unsigned)v.x;
        w.y = (unsigned)(v.x >> 32) | ((unsigned)v.y << 16);
        w.z = (unsigned)(v.y >> 16);
        *reinterpret_cast<u3 *>(d) = w;
    } else {
        *reinterpret_cast<ulonglong2 *>(d) = v;
    }
}
// grid (N/256, nl, X*2)
__global__ __launch_bounds__(256) void k_moddown_combine(const ModC *__restrict__ mod, int logN,
                                                         const u64 *__restrict__ acc, int acc_limbs,
                                                         const u64 *__restrict__ conv, const u64 *__restrict__ addend,
                                                         size_t axs, size_t aps, int add_polys, u64 *__restrict__ out, int nl,
                                                         ScaleSel pinv, const unsigned *__restrict__ galois,
                                                         int same_g) {
    const int N = 1 << logN;
    const int j = blockIdx.y, xp = blockIdx.z, x = xp >> 1, p = xp & 1;
    const u64 q = mod[j].q;
    const unsigned co = blockIdx.x * 256 + threadIdx.x;
    unsigned c = co;
    if (galois) {
        const unsigned g = galois[same_g ? 0 : x];
        if (g != 1u) {
            const unsigned e = ((2u * brev_n(co, logN) + 1u) * g) & (2u * N - 1u);
            c = brev_n((e - 1u) >> 1, logN);
        }
    }
    u64 v = submod(acc[((size_t)xp * acc_limbs + j) * N + c], conv[((size_t)xp * nl + j) * N + c], q);
    v = mulmod_shoup(v, pinv.s[j], pinv.s_sh[j], q);
    if (addend && p < add_polys) v = addmod(v, addend[(size_t)x * axs + (size_t)p * aps + (size_t)j * N + c], q);
    out[((size_t)xp * nl + j) * N + co] = v;
}
// grid (N/512, XP): see moddown_rescale_conv in kernels.h.  Two coefficients per thread.  (Sources are P-limb residues < 2^60:
// up to four terms stay below 2^(k+62), the range of reduce128k.)  The conversion constants arrive with
// P^{-1} (and the doubling) already folded in: tab.f[s][j] = (P/p_s mod q_j) * P^{-1} (* 2) mod q_j, so a target costs nP lazy
// multiply-accumulates and ONE reduction.
__global__ __launch_bounds__(256) void k_moddown_rescale_conv(const ModC *__restrict__ mod, int N, const u64 *__restrict__ y,
                                                              size_t yo, const u64 *__restrict__ u, size_t uo, u64 *__restrict__ w,
                                                              int l, int nP, ConvTab tab, int tz) {
    const int xp = blockIdx.y;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v[HY_MAX_DIGIT];
#pragma unroll
    for (int s = 0; s < HY_MAX_DIGIT; s++)
        if (s < nP) v[s] = *reinterpret_cast<const ulonglong2 *>(y + (size_t)xp * yo + (size_t)s * N + c);
    // the dropped limb of the ModDown output, coefficient form: y_l = u - conv_l P^{-1} (doubled when dbl)
    const ModC Ml = mod[l];
    u128 ax = 0, ay = 0;
#pragma unroll
    for (int s = 0; s < HY_MAX_DIGIT; s++)
        if (s < nP) {
            ax += (u128)v[s].x * tab.f[s][l];
            ay += (u128)v[s].y * tab.f[s][l];
        }
    const ulonglong2 uu = *reinterpret_cast<const ulonglong2 *>(u + (size_t)xp * uo + c);
    const u64 ylx = submod(uu.x, (nP <= 4 ? reduce128k(ax, Ml) : reduce128(ax, Ml)), Ml.q), yly = submod(uu.y, (nP <= 4 ? reduce128k(ay, Ml) : reduce128(ay, Ml)), Ml.q);
    const u64 half = Ml.q >> 1;
    const bool negx = ylx > half, negy = yly > half;
    const u64 magx = negx ? Ml.q - ylx : ylx, magy = negy ? Ml.q - yly : yly;  // |centred residue|
    const int j_lo = blockIdx.z * tz, j_hi = min(l, j_lo + tz);  // grid.z slices the targets of small launches
    for (int j = j_lo; j < j_hi; j++) {
        const ModC M = mod[j];
        u128 bx = 0, by = 0;
#pragma unroll
        for (int s = 0; s < HY_MAX_DIGIT; s++)
            if (s < nP) {
                bx += (u128)v[s].x * tab.f[s][j];
                by += (u128)v[s].y * tab.f[s][j];
            }
        const u64 rx = reduce64(magx, M), ry = reduce64(magy, M);
        ulonglong2 o;
        o.x = addmod((nP <= 4 ? reduce128k(bx, M) : reduce128(bx, M)), negx ? negmod(rx, M.q) : rx, M.q);
        o.y = addmod((nP <= 4 ? reduce128k(by, M) : reduce128(by, M)), negy ? negmod(ry, M.q) : ry, M.q);
        *reinterpret_cast<ulonglong2 *>(w + ((size_t)xp * l + j) * N + c) = o;
    }
}
// grid (N/512, XP)
__global__ __launch_bounds__(256) void k_moddown_last_limb(const ModC *__restrict__ mod, int N, const u64 *acc,  // u may be acc's row l
                                                           int acc_limbs, const u64 *__restrict__ addend, size_t add_x,
                                                           size_t add_p, u64 *u, size_t uo, int l, u64 pinv, u64 pinv_sh,
                                                           int dbl) {
    const int xp = blockIdx.y, x = xp >> 1, p = xp & 1;
    const u64 q = mod[l].q;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(acc + ((size_t)xp * acc_limbs + l) * N + c);
    const ulonglong2 d = *reinterpret_cast<const ulonglong2 *>(addend + (size_t)x * add_x + (size_t)p * add_p + (size_t)l * N + c);
    ulonglong2 r;
    r.x = addmod(mulmod_shoup(a.x, pinv, pinv_sh, q), d.x, q);
    r.y = addmod(mulmod_shoup(a.y, pinv, pinv_sh, q), d.y, q);
    if (dbl) {
        r.x = addmod(r.x, r.x, q);
        r.y = addmod(r.y, r.y, q);
    }
    *reinterpret_cast<ulonglong2 *>(u + (size_t)xp * uo + c) = r;
}
// grid (N/256, l, X)
__global__ __launch_bounds__(256) void k_rescale_spread(const ModC *__restrict__ mod, int N, const u64 *__restrict__ t,
                                                        u64 *__restrict__ tmp, int l) {
    const int j = blockIdx.y, x = blockIdx.z;
    const ModC M = mod[j];
    const u64 ql = mod[l].q, half = ql >> 1;
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64 v = t[(size_t)x * N + c];
    tmp[((size_t)x * l + j) * N + c] = v > half ? negmod(reduce64(ql - v, M), M.q) : reduce64(v, M);
}
__global__ __launch_bounds__(256) void k_rescale_combine(const ModC *__restrict__ mod, int N, const u64 *__restrict__ in,
                                                         const u64 *__restrict__ tmp, u64 *__restrict__ out, int l,
                                                         ScaleSel qlinv, int in_ls) {
    const int j = blockIdx.y, x = blockIdx.z;
    const u64 q = mod[j].q;
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u64 v = submod(in[((size_t)x * in_ls + j) * N + c], tmp[((size_t)x * l + j) * N + c], q);
    out[((size_t)x * l + j) * N + c] = mulmod_shoup(v, qlinv.s[j], qlinv.s_sh[j], q);
}

// ------------------------------------------------------------------------------------------------ loop B
// acc[g][{d0,d1,d2}][j][c] = sum_{i<dim} rot[i] (x) db[g][i] with 128-bit lazy accumulation: one double-word
// Barrett per output instead of 4*dim reductions.  45/46-bit limbs never overflow (dim * 2^93 < 2^128); the 60-bit
// limb folds its accumulators every 64 diagonals.  16 B per lane per operand (1 KiB per wave instruction).
//
// Work split (HBM must see the 3 GiB of rotated queries ONCE, not once per block): a workgroup owns one 128-coefficient
// tile of one limb and NW*BPP database blocks — each of its NW waves serves BPP blocks with one register copy of the
// rot operands, and the NW waves read the SAME rot addresses in step (one barrier per diagonal), so the per-CU vector
// cache serves NW-1 of them.  The database operands are streamed with non-temporal loads so they do not evict rot.
// grid (256 tiles * G/(NW*BPP), nl), block group fastest.
// Database residues of the 45/46-bit limbs are stored as 48-bit integers (two per 12-byte load): -23 % HBM bytes on the
// operand that dominates loop B.  Limb 0 (60 bit) and the rotated queries stay 8-byte.
// what a loop-B wave adds to its operand pointer: first byte of (block g0, diagonal 0, polynomial 0, its two residues) and the strides
// to the next block of the wave, the next diagonal, the other polynomial.  g0 = first block of the wave, grp = its workgroup's group
struct DbWalk {
    size_t base, su, si, sp;
};
DEV DbWalk db_walk(const DbLayout &L, int N, int dim, int j, int tile, int lane, int g0, int grp, int u0) {
    const size_t es = (L.packed && j > 0) ? 6 : 8;
    DbWalk w;
    if (!L.seq) {
        w.base = (size_t)g0 * dim * L.ct_bytes + db_limb_offset(L, N, j) + ((size_t)tile * 128 + lane * 2) * es;
        w.su = (size_t)dim * L.ct_bytes;
        w.si = L.ct_bytes;
        w.sp = L.poly_bytes;
    } else {
        const size_t groups = L.blocks / L.seq, ub = db_unit_bytes(L, j);
        // (bits46: the lane's two residues start at bit 92 lane of the unit; it loads 16 bytes from the dword that holds that bit)
        const size_t in_unit = (L.bits46 && j > 0) ? db_lane_load46(lane) : (size_t)lane * 2 * es;
        w.base = (size_t)L.blocks * L.bd * 2 * db_limb_offset(L, N, j) + ((((size_t)tile * groups + grp) * L.bd) * L.seq + u0) * 2 * ub + in_unit;
        w.su = 2 * ub;
        w.si = (size_t)L.seq * 2 * ub;
        w.sp = ub;
    }
    return w;
}
// One Karatsuba step per coefficient: d0 += a0 b0, d2 += a1 b1, dk += (a0+a1)(b0+b1); d1 = dk - d0 - d2 at the end.
// Three 64x64->128 products per coefficient instead of four — loop B is co-bound by the integer multiplier, not only
// by HBM (gfx950 builds a 128-bit product from four v_mad_u64_u32).
template <int BPP, int NW, bool NT, bool PK>
__global__ __launch_bounds__(64 * NW, 2) void k_hydia_tensor(const ModC *__restrict__ mod, int N, const u64 *__restrict__ rot,
                                                             const unsigned char *__restrict__ db, u64 *__restrict__ acc,
                                                             int dim, int nl, int Gq, int xcd_map, DbLayout L, int j0, int ng, int nblk) {
    const int j = blockIdx.y + j0;
    // consecutive workgroup ids are dealt round-robin over the 8 XCDs: give every XCD its own tiles and let the Gq block
    // groups of one tile follow each other ON THAT XCD, so they find the tile's rot lines in its L2
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int gq = xcd_map ? k % Gq : blockIdx.x % Gq;
    const int tile = xcd_map ? xcd + 8 * (k / Gq) : blockIdx.x / Gq;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ModC M = mod[j];
    const size_t c = (size_t)tile * 128 + lane * 2;
    const size_t ps = (size_t)nl * N, cs = 2 * ps;  // rot / acc poly stride, ciphertext stride (elements)
    const int g0 = (gq * NW + wv) * BPP;
    const u64 *ra = rot + (size_t)j * N + c;
    const DbWalk dw = db_walk(L, N, dim, j, tile, lane, g0, gq, wv * BPP);
    const unsigned char *da = db + dw.base;
    const size_t db_cs = dw.si, db_ps = dw.sp, db_bs = dw.su;
    const int kbits = M.ks + 2;
    const int chunk = (125 - 2 * kbits >= 30) ? dim : (1 << (125 - 2 * kbits));
    u128 d0x[BPP], d0y[BPP], dkx[BPP], dky[BPP], d2x[BPP], d2y[BPP];
#pragma unroll
    for (int u = 0; u < BPP; u++) d0x[u] = d0y[u] = dkx[u] = dky[u] = d2x[u] = d2y[u] = 0;
    // One diagonal's operands are fetched while the previous one's products run: a wave always has 2 + 4 BPP loads in flight instead
    // of none during its ~230 multiply-accumulate instructions (188 registers, two waves per SIMD; 27.2 -> 26.65 ms at 64 blocks on the
    // same GPU.  Forcing three waves per SIMD spills and loses it: 28.0 ms).
    struct Operands {
        ulonglong2 a0, a1;
        DbRaw<PK> b0[BPP], b1[BPP];
    };
    auto fetch = [&](Operands &o, int i) {
        o.a0 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs);
        o.a1 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs + ps);
#pragma unroll
        for (int u = 0; u < BPP; u++) {
            o.b0[u].template load<NT>(da + u * db_bs + (size_t)i * db_cs);
            o.b1[u].template load<NT>(da + u * db_bs + (size_t)i * db_cs + db_ps);
        }
    };
    auto accumulate = [&](const Operands &o) {
        const u64 sax = o.a0.x + o.a1.x, say = o.a0.y + o.a1.y;
#pragma unroll
        for (int u = 0; u < BPP; u++) {
            const ulonglong2 b0 = o.b0[u].get(), b1 = o.b1[u].get();
            d0x[u] += (u128)o.a0.x * b0.x;
            d0y[u] += (u128)o.a0.y * b0.y;
            d2x[u] += (u128)o.a1.x * b1.x;
            d2y[u] += (u128)o.a1.y * b1.y;
            dkx[u] += (u128)sax * (b0.x + b1.x);
            dky[u] += (u128)say * (b0.y + b1.y);
        }
    };
    Operands cur, nxt;
    fetch(cur, 0);
    for (int i0 = 0; i0 < dim; i0 += chunk) {
        const int i1 = i0 + chunk < dim ? i0 + chunk : dim;
        // dim (>= 2, checked at context creation) and every chunk are powers of two: even.  No branch inside the loop: at a join the
        // compiler waits for every outstanding load, the prefetched ones included
        for (int i = i0; i < i1; i += 2) {
            fetch(nxt, i + 1);
            accumulate(cur);
            if (NW > 1) __builtin_amdgcn_s_barrier();  // keep the waves on the same diagonal (no memory wait implied)
            fetch(cur, i + 2 < dim ? i + 2 : i + 1);   // the last one re-reads a line that is in flight: never used
            accumulate(nxt);
            if (NW > 1) __builtin_amdgcn_s_barrier();
        }
        if (i1 < dim) {
#pragma unroll
            for (int u = 0; u < BPP; u++) {
                d0x[u] = reduce128(d0x[u], M); d0y[u] = reduce128(d0y[u], M);
                dkx[u] = reduce128(dkx[u], M); dky[u] = reduce128(dky[u], M);
                d2x[u] = reduce128(d2x[u], M); d2y[u] = reduce128(d2y[u], M);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < BPP; u++) {
        ulonglong2 r0, r1, r2;
        r0.x = reduce128(d0x[u], M); r0.y = reduce128(d0y[u], M);
        r2.x = reduce128(d2x[u], M); r2.y = reduce128(d2y[u], M);
        r1.x = submod(submod(reduce128(dkx[u], M), r0.x, M.q), r2.x, M.q);
        r1.y = submod(submod(reduce128(dky[u], M), r0.y, M.q), r2.y, M.q);
        // ng > 0 (baby-step / giant-step split): accumulator of "block" gi = (database block, giant g) goes to slot g * blocks + block,
        // so that the partial sums of one giant step over all database blocks are one contiguous batch
        const int gi = g0 + u, go = ng > 0 ? (gi % ng) * nblk + gi / ng : gi;
        u64 *o = acc + ((size_t)go * 3 * nl + j) * N + c;
        *reinterpret_cast<ulonglong2 *>(o) = r0;
        *reinterpret_cast<ulonglong2 *>(o + ps) = r1;
        *reinterpret_cast<ulonglong2 *>(o + 2 * ps) = r2;
    }
}

// Loop B on the packed limbs of a group-sequential database (residues and rotated-query residues below 2^48).  There HBM delivers
// 7 TB/s and the 128-bit multiply-accumulates above (94 % of the vector issue slots at 6 TB/s) would be the limit, so the products are
// taken on 24-bit halves, a = ah 2^24 + al, b = bh 2^24 + bl: the partial sums  ll = sum al bl,  mid = sum (al bh + ah bl),
// hh = sum ah bh  stay below 2^63 for up to 4096 diagonals (Karatsuba's operand sums included: halves below 2^25), so every
// multiply-accumulate is ONE v_mad_u64_u32 with no carry — 12 per coefficient instead of three 128-bit ones of ~9 instructions each;
// the 6-byte residues are cut into halves straight from the three loaded dwords.  Same sums, same final reduction.
// (The halves pass through an empty asm statement: knowing an operand has 24 bits the compiler (ROCm 7.2) forms 24-bit multiplies,
// drops the masks they make redundant, and then fuses some of them back into v_mad_u64_u32 on the UNMASKED registers — wrong
// products; tools/ubench/tensor_check.cpp found it.)
DEV unsigned hide24(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}
struct Acc24 {
    u64 ll, mid, hh;
    DEV void mac(unsigned al, unsigned ah, unsigned bl, unsigned bh) {
        ll += (u64)al * bl;
        mid += (u64)al * bh;
        mid += (u64)ah * bl;
        hh += (u64)ah * bh;
    }
    DEV u128 wide() const { return (u128)ll + ((u128)mid << 24) + ((u128)hh << 48); }
};
// two 46-bit residues at bit `s` (a multiple of 4 below 32) of four dwords (bits46 layout): fetched as one 4-byte-aligned 16-byte load
struct DbRaw46 {
    typedef unsigned int u4a __attribute__((ext_vector_type(4), aligned(4)));
    u4a w;
    template <bool NT>
    DEV void load(const unsigned char *p) {
        w = NT ? __builtin_nontemporal_load(reinterpret_cast<const u4a *>(p)) : *reinterpret_cast<const u4a *>(p);
    }
};
template <int BPP, int NW, bool B46, bool D2 = B46>
__global__ __launch_bounds__(64 * NW, 2) void k_hydia_tensor24(const ModC *__restrict__ mod, int N, const u64 *__restrict__ rot,
                                                               const unsigned char *__restrict__ db, u64 *__restrict__ acc,
                                                               int dim, int nl, int Gq, int xcd_map, DbLayout L, int j0, int ng, int nblk) {
    const int j = blockIdx.y + j0;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int gq = xcd_map ? k % Gq : blockIdx.x % Gq;
    const int tile = xcd_map ? xcd + 8 * (k / Gq) : blockIdx.x / Gq;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ModC M = mod[j];
    const size_t c = (size_t)tile * 128 + lane * 2;
    const size_t ps = (size_t)nl * N, cs = 2 * ps;
    const int g0 = (gq * NW + wv) * BPP;
    const u64 *ra = rot + (size_t)j * N + c;
    const DbWalk dw = db_walk(L, N, dim, j, tile, lane, g0, gq, wv * BPP);
    const unsigned char *da = db + dw.base;
    const size_t db_cs = dw.si, db_ps = dw.sp, db_bs = dw.su;
    Acc24 d0[BPP][2], dk[BPP][2], d2[BPP][2];  // [block][coefficient of the lane's pair]
#pragma unroll
    for (int u = 0; u < BPP; u++)
#pragma unroll
        for (int e = 0; e < 2; e++) d0[u][e] = dk[u][e] = d2[u][e] = Acc24{0, 0, 0};
    typedef typename std::conditional<B46, DbRaw46, DbRaw<true>>::type Raw;
    const unsigned s46 = (unsigned)(lane * 92) & 31u;  // B46: bit of the lane's first residue inside its first dword
    struct Operands {
        ulonglong2 a0, a1;
        Raw b0[BPP], b1[BPP];
    };
    auto fetch = [&](Operands &o, int i) {
        o.a0 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs);
        o.a1 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs + ps);
#pragma unroll
        for (int u = 0; u < BPP; u++) {
            o.b0[u].template load<true>(da + u * db_bs + (size_t)i * db_cs);
            o.b1[u].template load<true>(da + u * db_bs + (size_t)i * db_cs + db_ps);
        }
    };
    auto accumulate = [&](const Operands &o) {
        // rotated-query residues (8 bytes each, below 2^48): halves of both polynomials and of their sum, shared by the BPP blocks
        const u64 av[2][2] = {{o.a0.x, o.a0.y}, {o.a1.x, o.a1.y}};
        unsigned al[2][2], ah[2][2], sl[2], sh[2];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                al[p][e] = hide24((unsigned)av[p][e] & 0xFFFFFFu);
                ah[p][e] = __builtin_amdgcn_alignbit((unsigned)(av[p][e] >> 32), (unsigned)av[p][e], 24);
            }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            sl[e] = al[0][e] + al[1][e];
            sh[e] = ah[0][e] + ah[1][e];
        }
#pragma unroll
        for (int u = 0; u < BPP; u++) {
            // database residues: two 48-bit integers in three dwords -> four 24-bit halves per polynomial
            unsigned bl[2][2], bh[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const auto w = p == 0 ? o.b0[u].w : o.b1[u].w;
                if constexpr (B46) {  // T = (w3:w2:w1:w0) >> s: residue 0 = T[0, 46), residue 1 = T[46, 92); halves of 24 and 22 bits
                    const unsigned t0 = __builtin_amdgcn_alignbit(w[1], w[0], s46), t1 = __builtin_amdgcn_alignbit(w[2], w[1], s46),
                                   t2 = __builtin_amdgcn_alignbit(w[3], w[2], s46);
                    bl[p][0] = hide24(t0 & 0xFFFFFFu);
                    bh[p][0] = hide24(__builtin_amdgcn_alignbit(t1, t0, 24) & 0x3FFFFFu);
                    bl[p][1] = hide24(__builtin_amdgcn_alignbit(t2, t1, 14) & 0xFFFFFFu);
                    bh[p][1] = hide24((t2 >> 6) & 0x3FFFFFu);
                } else {
                    bl[p][0] = hide24(w[0] & 0xFFFFFFu);
                    bh[p][0] = hide24(__builtin_amdgcn_alignbit(w[1], w[0], 24) & 0xFFFFFFu);
                    bl[p][1] = hide24(__builtin_amdgcn_alignbit(w[2], w[1], 16) & 0xFFFFFFu);
                    bh[p][1] = hide24(w[2] >> 8);
                }
            }
#pragma unroll
            for (int e = 0; e < 2; e++) {
                d0[u][e].mac(al[0][e], ah[0][e], bl[0][e], bh[0][e]);
                d2[u][e].mac(al[1][e], ah[1][e], bl[1][e], bh[1][e]);
                dk[u][e].mac(sl[e], sh[e], bl[0][e] + bl[1][e], bh[0][e] + bh[1][e]);
            }
        }
    };
    if constexpr (D2) {
        // 46-bit units: 4 % fewer bytes per diagonal, and with one diagonal in flight per wave the launch did not get shorter — it is
        // bound by what a CU keeps in flight (3 workgroups x one diagonal = 35 KB; 1.4 us of latency), not by HBM.  So the database
        // operands run TWO diagonals ahead here (three rotating sets), the rotated-query lines (L2) one ahead as before.
        Operands A, B, C;
        fetch(A, 0);
        fetch(B, 1);
        int i = 0;
        for (; i + 2 < dim; i += 3) {  // branch-free inside: clamped re-fetches of the last diagonal are never accumulated
            fetch(C, i + 2);
            accumulate(A);
            if (NW > 1) __builtin_amdgcn_s_barrier();
            fetch(A, i + 3 < dim ? i + 3 : dim - 1);
            accumulate(B);
            if (NW > 1) __builtin_amdgcn_s_barrier();
            fetch(B, i + 4 < dim ? i + 4 : dim - 1);
            accumulate(C);
            if (NW > 1) __builtin_amdgcn_s_barrier();
        }
        if (i < dim) accumulate(A);      // the one or two diagonals the groups of three leave over (workgroup-uniform)
        if (i + 1 < dim) accumulate(B);
    } else {
        Operands cur, nxt;
        fetch(cur, 0);
        for (int i = 0; i < dim; i += 2) {  // dim is a power of two >= 2; no branch inside (see k_hydia_tensor)
            fetch(nxt, i + 1);
            accumulate(cur);
            if (NW > 1) __builtin_amdgcn_s_barrier();
            fetch(cur, i + 2 < dim ? i + 2 : i + 1);
            accumulate(nxt);
            if (NW > 1) __builtin_amdgcn_s_barrier();
        }
    }
#pragma unroll
    for (int u = 0; u < BPP; u++) {
        ulonglong2 r0, r1, r2;
        r0.x = reduce128(d0[u][0].wide(), M); r0.y = reduce128(d0[u][1].wide(), M);
        r2.x = reduce128(d2[u][0].wide(), M); r2.y = reduce128(d2[u][1].wide(), M);
        r1.x = submod(submod(reduce128(dk[u][0].wide(), M), r0.x, M.q), r2.x, M.q);
        r1.y = submod(submod(reduce128(dk[u][1].wide(), M), r0.y, M.q), r2.y, M.q);
        const int gi = g0 + u, go = ng > 0 ? (gi % ng) * nblk + gi / ng : gi;
        u64 *o = acc + ((size_t)go * 3 * nl + j) * N + c;
        *reinterpret_cast<ulonglong2 *>(o) = r0;
        *reinterpret_cast<ulonglong2 *>(o + ps) = r1;
        *reinterpret_cast<ulonglong2 *>(o + 2 * ps) = r2;
    }
}

// Loop B for SMALL databases (at most 8 blocks on this GPU): the limb-0 launch of k_hydia_tensor has only 256 x G waves, each
// walking all `dim` diagonals — latency-bound (0.6 ms at G = 1 for 0.5 GB).  Here KS waves of a workgroup share one
// (block, tile) and take every KS-th diagonal; the partial sums are reduced modulo q_j through LDS.  Same residues as the
// one-wave kernel (a sum modulo q does not depend on how it is split).
template <int KS, bool PK>
__global__ __launch_bounds__(64 * KS) void k_hydia_tensor_sk(const ModC *__restrict__ mod, int N, const u64 *__restrict__ rot,
                                                             const unsigned char *__restrict__ db, u64 *__restrict__ acc, int dim,
                                                             int nl, DbLayout L, int j0, int ng, int nblk) {
    __shared__ u64 part[KS][6][64];
    const int j = blockIdx.y + j0, tiles = N / 128;
    const int tile = blockIdx.x % tiles, g = blockIdx.x / tiles;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ModC M = mod[j];
    const size_t c = (size_t)tile * 128 + lane * 2;
    const size_t ps = (size_t)nl * N, cs = 2 * ps;
    const u64 *ra = rot + (size_t)j * N + c;
    const unsigned char *da = db + (size_t)g * dim * L.ct_bytes + db_limb_offset(L, N, j) + c * (PK ? 6 : 8);
    const int kbits = M.ks + 2;
    const int chunk = (125 - 2 * kbits >= 30) ? dim : (1 << (125 - 2 * kbits));  // products a lazy 128-bit sum can take
    u128 d0x = 0, d0y = 0, dkx = 0, dky = 0, d2x = 0, d2y = 0;
    int since = 0;
    for (int i = wv; i < dim; i += KS) {
        const ulonglong2 a0 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs);
        const ulonglong2 a1 = *reinterpret_cast<const ulonglong2 *>(ra + (size_t)i * cs + ps);
        const ulonglong2 b0 = db_load2<PK, true>(da + (size_t)i * L.ct_bytes);
        const ulonglong2 b1 = db_load2<PK, true>(da + (size_t)i * L.ct_bytes + L.poly_bytes);
        d0x += (u128)a0.x * b0.x;
        d0y += (u128)a0.y * b0.y;
        d2x += (u128)a1.x * b1.x;
        d2y += (u128)a1.y * b1.y;
        dkx += (u128)(a0.x + a1.x) * (b0.x + b1.x);
        dky += (u128)(a0.y + a1.y) * (b0.y + b1.y);
        if (++since == chunk) {
            since = 0;
            d0x = reduce128(d0x, M); d0y = reduce128(d0y, M);
            dkx = reduce128(dkx, M); dky = reduce128(dky, M);
            d2x = reduce128(d2x, M); d2y = reduce128(d2y, M);
        }
    }
    part[wv][0][lane] = reduce128(d0x, M); part[wv][1][lane] = reduce128(d0y, M);
    part[wv][2][lane] = reduce128(dkx, M); part[wv][3][lane] = reduce128(dky, M);
    part[wv][4][lane] = reduce128(d2x, M); part[wv][5][lane] = reduce128(d2y, M);
    __syncthreads();
    if (wv == 0) {
        u64 r[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            u64 t = part[0][k][lane];
            for (int w = 1; w < KS; w++) t = addmod(t, part[w][k][lane], M.q);
            r[k] = t;
        }
        const int go = ng > 0 ? (g % ng) * nblk + g / ng : g;  // giant-major order (see k_hydia_tensor)
        u64 *o = acc + ((size_t)go * 3 * nl + j) * N + c;
        *reinterpret_cast<ulonglong2 *>(o) = make_ulonglong2(r[0], r[1]);
        *reinterpret_cast<ulonglong2 *>(o + ps) = make_ulonglong2(submod(submod(r[2], r[0], M.q), r[4], M.q), submod(submod(r[3], r[1], M.q), r[5], M.q));
        *reinterpret_cast<ulonglong2 *>(o + 2 * ps) = make_ulonglong2(r[4], r[5]);
    }
}


// unpacked [X][2][nQ][N] u64  <->  database layout.  grid (N/512, nQ, X*2)
template <bool PACK>
__global__ __launch_bounds__(256) void k_db_repack(int N, int nQ, u64 *__restrict__ plain, unsigned char *__restrict__ db,
                                                   DbLayout L, size_t t0) {
    const int j = blockIdx.y, xp = blockIdx.z, x = xp >> 1, p = xp & 1;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 2;
    u64 *pl = plain + ((size_t)xp * nQ + j) * N + c;
    const bool pk = L.packed && j > 0;
    unsigned char *d = db + db_offset(L, N, t0 + x, p, j, c);
    typedef unsigned int u3 __attribute__((ext_vector_type(3), aligned(4)));
    if (PACK) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(pl);
        if (pk) {
            u3 w;
            w.x = (unsigned)v.x;
            w.y = (unsigned)(v.x >> 32) | ((unsigned)v.y << 16);
            w.z = (unsigned)(v.y >> 16);
            *reinterpret_cast<u3 *>(d) = w;
        } else {
            *reinterpret_cast<ulonglong2 *>(d) = v;
        }
    } else {
        *reinterpret_cast<ulonglong2 *>(pl) = pk ? db_load2<true, false>(d) : db_load2<false, false>(d);
    }
}

// the 46-bit limbs of a bits46 layout: a thread moves SIXTEEN residues = 23 dwords (the granule that starts on a dword).
// grid (N/4096, nQ - 1, X*2): limb j = blockIdx.y + 1
template <bool PACK>
__global__ __launch_bounds__(256) void k_db_repack46(int N, int nQ, u64 *__restrict__ plain, unsigned char *__restrict__ db, DbLayout L, size_t t0) {
    const int j = blockIdx.y + 1, xp = blockIdx.z, x = xp >> 1, p = xp & 1;
    const size_t c = (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
    if (c >= (size_t)N) return;
    u64 *pl = plain + ((size_t)xp * nQ + j) * N + c;
    unsigned *d = reinterpret_cast<unsigned *>(db + db_offset(L, N, t0 + x, p, j, c));
    unsigned w[24];
    if (PACK) {
#pragma unroll
        for (int k = 0; k < 24; k++) w[k] = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u64 v = pl[r] & ((1ull << 46) - 1);
            const int bit = 46 * r, di = bit >> 5, sh = bit & 31;
            w[di] |= (unsigned)(v << sh);
            w[di + 1] |= (unsigned)(sh ? v >> (32 - sh) : v >> 32);
            if (sh > 18) w[di + 2] |= (unsigned)(v >> (64 - sh));  // 46 + sh > 64: the field reaches a third dword
        }
#pragma unroll
        for (int k = 0; k < 23; k++) d[k] = w[k];
    } else {
#pragma unroll
        for (int k = 0; k < 23; k++) w[k] = d[k];
        w[23] = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int bit = 46 * r, di = bit >> 5, sh = bit & 31;
            u64 v = ((u64)w[di] >> sh) | ((u64)w[di + 1] << (32 - sh));
            if (sh > 18) v |= (u64)w[di + 2] << (64 - sh);
            pl[r] = v & ((1ull << 46) - 1);
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_uniform_hash(const ModC *__restrict__ mod, int N, u64 *__restrict__ dst,
                                                           int nl, u64 seed) {
    const size_t lp = blockIdx.y + (size_t)blockIdx.z * gridDim.y;
    const ModC M = mod[lp % nl];
    const size_t idx = lp * N + (size_t)blockIdx.x * 256 + threadIdx.x;
    u64 z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    dst[idx] = reduce64(z, M);
}

}  // namespace

// ================================================================================================ launchers
namespace hk {

// ---- byte ledger
namespace {
bool g_ledger_on = false;
std::mutex g_ledger_mu;
std::map<std::string, std::pair<long, double>> g_ledger;
}  // namespace
void ledger_enable(bool on) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    g_ledger_on = on;
    g_ledger.clear();
}
void ledger_add(const char *kernel, double bytes) {
    if (!g_ledger_on) return;
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    auto &e = g_ledger[kernel];
    e.first++;
    e.second += bytes;
}
size_t ledger_dump(char *out, size_t cap) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    std::string t;
    char line[256];
    for (auto &kv : g_ledger) {
        snprintf(line, sizeof line, "%s\t%ld\t%.0f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        t += line;
    }
    if (out && cap) {
        const size_t n = std::min(cap - 1, t.size());
        memcpy(out, t.data(), n);
        out[n] = 0;
    }
    return t.size() + 1;
}
#define LP_BYTES(N) ((double)(N) * 8.0)

void ntt_forward(hipStream_t st, const NttTables &T, int logN, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                 const LimbSel &sel) {
    if (logN == 15 && !T.generic) return ntt15_forward(st, T, src, dst, so, dso, X, sel);
    const int N = 1 << logN, R = N >> 8;
    ScaleSel dummy = {};
    hipLaunchKernelGGL(k_ntt_strided<false>, dim3(8, X * sel.n), dim3(256), (size_t)R * 32 * 8, st, T, logN, src, dst, so,
                       dso, sel, dummy);
    hipLaunchKernelGGL(k_ntt_contig<false>, dim3(N / 2048, X * sel.n), dim3(256), 0, st, T, logN, dst, dst, dso, dso, sel);
}
void ntt_inverse(hipStream_t st, const NttTables &T, int logN, const u64 *src, u64 *dst, size_t so, size_t dso, int X,
                 const LimbSel &sel, const ScaleSel &scale) {
    if (logN == 15 && !T.generic) return ntt15_inverse(st, T, src, dst, so, dso, X, sel, scale);
    const int N = 1 << logN, R = N >> 8;
    hipLaunchKernelGGL(k_ntt_contig<true>, dim3(N / 2048, X * sel.n), dim3(256), 0, st, T, logN, src, dst, so, dso, sel);
    hipLaunchKernelGGL(k_ntt_strided<true>, dim3(8, X * sel.n), dim3(256), (size_t)R * 32 * 8, st, T, logN, dst, dst, dso,
                       dso, sel, scale);
}
void add(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int XP, const LimbSel &sel, int a_ls,
         int b_ls, int o_ls) {
    ledger_add("k_addsub", 3.0 * XP * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_addsub<0>, dim3(N / 512, XP * sel.n), dim3(256), 0, st, mod, N, a, b, o, sel, a_ls, b_ls, o_ls);
}
void sub(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int XP, const LimbSel &sel, int a_ls,
         int b_ls, int o_ls) {
    ledger_add("k_addsub", 3.0 * XP * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_addsub<1>, dim3(N / 512, XP * sel.n), dim3(256), 0, st, mod, N, a, b, o, sel, a_ls, b_ls, o_ls);
}
void add_raw(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int XP, const LimbSel &sel, int a_ls,
             int b_ls, int o_ls) {
    ledger_add("k_addsub", 3.0 * XP * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_addsub<2>, dim3(N / 512, XP * sel.n), dim3(256), 0, st, mod, N, a, b, o, sel, a_ls, b_ls, o_ls);
}
void mod_reduce(hipStream_t st, const ModC *mod, int N, u64 *a, int XP, const LimbSel &sel, int a_ls) {
    ledger_add("k_mod_reduce", 2.0 * XP * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_mod_reduce, dim3(N / 512, XP * sel.n), dim3(256), 0, st, mod, N, a, sel, a_ls);
}
void mul_scalar(hipStream_t st, const ModC *mod, int N, const u64 *a, u64 *o, int XP, const LimbSel &sel,
                const ScaleSel &c, int a_ls, int o_ls) {
    ledger_add("k_mul_scalar", 2.0 * XP * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_mul_scalar, dim3(N / 512, XP * sel.n), dim3(256), 0, st, mod, N, a, o, sel, c, a_ls, o_ls);
}
void lincomb(hipStream_t st, const ModC *mod, int N, const LinComb &lc, u64 *o, int X, int npoly, int nl) {
    ledger_add("k_lincomb", (lc.nterms + 1.0) * X * npoly * nl * LP_BYTES(N));
    hipLaunchKernelGGL(k_lincomb, dim3(N / 512, nl, X * npoly), dim3(256), 0, st, mod, N, lc, o, npoly, nl);
}
void lincomb_multi(hipStream_t st, const ModC *mod, int N, const LinCombMulti &lc, u64 *o, int X, int npoly, int nl) {
    ledger_add("k_lincomb_multi", (lc.nterms + (double)lc.K) * X * npoly * nl * LP_BYTES(N));
    hipLaunchKernelGGL(k_lincomb_multi, dim3(N / 512, nl, X * npoly), dim3(256), 0, st, mod, N, lc, o, X * npoly, npoly, nl);
}
void batch_sum(hipStream_t st, const ModC *mod, int N, const u64 *in, u64 *o, int X, int npoly, int nl, int stride, int nout) {
    ledger_add("k_batch_sum", (X + 1.0) * nout * npoly * nl * LP_BYTES(N));
    hipLaunchKernelGGL(k_batch_sum, dim3(N / 512, nl, npoly * nout), dim3(256), 0, st, mod, N, in, o, X, npoly, nl, stride);
}
void add_scalar(hipStream_t st, const ModC *mod, int N, u64 *a, size_t outer, int X, const LimbSel &sel,
                const ScaleSel &c) {
    ledger_add("k_add_scalar", 2.0 * X * sel.n * LP_BYTES(N));
    hipLaunchKernelGGL(k_add_scalar, dim3(N / 512, X * sel.n), dim3(256), 0, st, mod, N, a, outer, sel, c);
}
void copy_limbs(hipStream_t st, int N, const u64 *src, u64 *dst, size_t so, size_t dso, int X, int nlimbs) {
    ledger_add("k_copy_limbs", 2.0 * X * nlimbs * LP_BYTES(N));
    hipLaunchKernelGGL(k_copy_limbs, dim3(N / 512, X * nlimbs), dim3(256), 0, st, N, src, dst, so, dso, nlimbs);
}
void tensor(hipStream_t st, const ModC *mod, int N, const u64 *a, const u64 *b, u64 *o, int X, int nl, int a_ls, int b_ls,
            const u64 *c, int c_ls, const ScaleSel *kap) {
    ledger_add(c ? "k_tensor<true>" : "k_tensor<false>", (c ? 9.0 : 7.0) * X * nl * LP_BYTES(N));  // a0 a1 b0 b1 (c0 c1) in, d0 d1 d2 out
    if (c)
        hipLaunchKernelGGL(k_tensor<true>, dim3(N / 512, nl, X), dim3(256), 0, st, mod, N, a, b, o, nl, a_ls, b_ls, c, c_ls, *kap);
    else
        hipLaunchKernelGGL(k_tensor<false>, dim3(N / 512, nl, X), dim3(256), 0, st, mod, N, a, b, o, nl, a_ls, b_ls,
                           (const u64 *)nullptr, 0, ScaleSel{});
}
// conversion kernels let one thread produce every target limb (sources read once).  With few polynomials that is N/512 * X
// workgroups, each a long serial chain: below ~2 workgroups per CU the targets are sliced over grid.z instead (sources come from L2)
static int small_launch_targets(int N, int X, int nt) {
    const int wgs = (N / 512) * X;
    if (nt <= 1 || wgs >= 512) return nt > 0 ? nt : 1;
    const int slices = std::min(nt, (512 + wgs - 1) / wgs);
    return (nt + slices - 1) / slices;
}
void base_convert(hipStream_t st, const ModC *mod, int N, const u64 *y, size_t yo, u64 *out, size_t oo, int X,
                  const ConvTab &tab, const LimbSel &dsel) {
    ledger_add("k_base_convert", (double)X * (tab.ns + tab.nt - (tab.skip_hi - tab.skip_lo)) * LP_BYTES(N));  // sources once, every target once
    const int tz = small_launch_targets(N, X, tab.nt);
    hipLaunchKernelGGL(k_base_convert, dim3(N / 512, X, (tab.nt + tz - 1) / tz), dim3(256), 0, st, mod, N, y, yo, out, oo, tab, dsel, tz);
}
void base_convert_digits(hipStream_t st, const ModC *mod, int N, const u64 *y, size_t yo, u64 *out, size_t oo, int X,
                          const ConvTab *d_tabs, int nd, int nl, int nE, const LimbSel &esel) {
    ledger_add("k_base_convert_digits", (double)X * nd * nE * LP_BYTES(N));  // nl sources once + (nd nE - nl) targets once
    const int tz = small_launch_targets(N, X * nd, nE);
    const int slices = (nE + tz - 1) / tz;
    hipLaunchKernelGGL(k_base_convert_digits, dim3(N / 512, X, nd * slices), dim3(256), 0, st, mod, N, y, yo, out, oo, d_tabs, esel, slices, tz, nE);
}
void inner_product(hipStream_t st, const ModC *mod, int N, const u64 *dig, size_t dxs, int nd, const u64 *const *keys,
                   int same_key, int nT, u64 *acc, int X, const LimbSel &esel, const u64 *own, size_t own_xs, int alpha, int nl,
                   int acc_rows, int packed_nQ, int dig_rows, int dig_t0) {
    const int rows = acc_rows > 0 ? acc_rows : esel.n, drows = dig_rows > 0 ? dig_rows : rows;
    {   // keys of X rotations streamed once (one shared key: once in all), digits / own limbs once per x unless shared (dxs == 0), acc out
        double rowb = 0;  // bytes of one (digit, poly) key row set restricted to the limbs of esel
        for (int t = 0; t < esel.n; t++) rowb += (packed_nQ > 0 && esel.mod[t] > 0 && esel.mod[t] < packed_nQ) ? N * 6.0 : N * 8.0;
        const double keyb = nd * 2 * rowb;
        const double digb = (double)nd * esel.n * LP_BYTES(N);
        ledger_add(packed_nQ > 0 ? "k_inner_product<true>" : "k_inner_product<false>",
                   (same_key ? keyb : keyb * X) + (dxs ? digb * X : digb) + 2.0 * X * esel.n * LP_BYTES(N));
    }
    if (packed_nQ > 0)
        hipLaunchKernelGGL(k_inner_product<true>, dim3(N / 512, esel.n, X), dim3(256), 0, st, mod, N, dig, dxs, nd, keys, same_key,
                           nT, acc, esel, own, own_xs, alpha, nl, rows, packed_nQ, drows, dig_t0);
    else
        hipLaunchKernelGGL(k_inner_product<false>, dim3(N / 512, esel.n, X), dim3(256), 0, st, mod, N, dig, dxs, nd, keys, same_key,
                           nT, acc, esel, own, own_xs, alpha, nl, rows, 0, drows, dig_t0);
}
size_t key_packed_bytes(int N, int nQ, int nT, int nd) { return (size_t)nd * 2 * key_set_bytes(N, nQ, nT); }
void key_pack(hipStream_t st, const ModC *mod, int N, int nQ, int nT, int nd, const u64 *key, void *out, const ScaleSel *premul) {
    ledger_add("k_key_pack", (double)nd * 2 * nT * LP_BYTES(N) + (double)key_packed_bytes(N, nQ, nT, nd));
    hipLaunchKernelGGL(k_key_pack, dim3(N / 512, nT, nd * 2), dim3(256), 0, st, mod, N, nQ, nT, key, (unsigned char *)out, premul ? 1 : 0,
                       premul ? *premul : ScaleSel{});
}
void moddown_combine(hipStream_t st, const ModC *mod, int logN, const u64 *acc, int acc_limbs, const u64 *conv,
                     const u64 *addend, size_t axs, size_t aps, int add_polys, u64 *out, int X, int nl,
                     const ScaleSel &pinv, const unsigned *galois, int same_g) {
    ledger_add("k_moddown_combine", (3.0 + (addend ? 0.5 * add_polys : 0.0)) * X * 2 * nl * LP_BYTES(1 << logN));
    hipLaunchKernelGGL(k_moddown_combine, dim3((1 << logN) / 256, nl, X * 2), dim3(256), 0, st, mod, logN, acc, acc_limbs,
                       conv, addend, axs, aps, add_polys, out, nl, pinv, galois, same_g);
}
void moddown_rescale_conv(hipStream_t st, const ModC *mod, int N, const u64 *y, size_t yo, const u64 *u, size_t uo, u64 *w, int XP, int l,
                          int nP, const ConvTab &tab) {
    ledger_add("k_moddown_rescale_conv", (double)XP * (nP + 1 + l) * LP_BYTES(N));  // y (nP limbs) + u in, l limbs out
    const int tz = small_launch_targets(N, XP, l);
    hipLaunchKernelGGL(k_moddown_rescale_conv, dim3(N / 512, XP, (l + tz - 1) / tz), dim3(256), 0, st, mod, N, y, yo, u, uo, w, l, nP, tab, tz);
}
void moddown_last_limb(hipStream_t st, const ModC *mod, int N, const u64 *acc, int acc_limbs, const u64 *addend, size_t add_x,
                       size_t add_p, u64 *u, size_t uo, int XP, int l, u64 pinv, u64 pinv_sh, int dbl) {
    ledger_add("k_moddown_last_limb", 3.0 * XP * LP_BYTES(N));
    hipLaunchKernelGGL(k_moddown_last_limb, dim3(N / 512, XP), dim3(256), 0, st, mod, N, acc, acc_limbs, addend, add_x, add_p, u, uo, l,
                       pinv, pinv_sh, dbl);
}
void rescale_spread(hipStream_t st, const ModC *mod, int N, const u64 *t, u64 *tmp, int X, int l) {
    ledger_add("k_rescale_spread", (1.0 + l) * X * LP_BYTES(N));
    hipLaunchKernelGGL(k_rescale_spread, dim3(N / 256, l, X), dim3(256), 0, st, mod, N, t, tmp, l);
}
void rescale_combine(hipStream_t st, const ModC *mod, int N, const u64 *in, const u64 *tmp, u64 *out, int X, int l,
                     const ScaleSel &qlinv, int in_ls) {
    ledger_add("k_rescale_combine", 3.0 * X * l * LP_BYTES(N));
    hipLaunchKernelGGL(k_rescale_combine, dim3(N / 256, l, X), dim3(256), 0, st, mod, N, in, tmp, out, l, qlinv, in_ls);
}
template <int BPP, int NW>
static void launch_tensor(hipStream_t st, const ModC *mod, int N, const u64 *rot, const void *db, u64 *acc, int G, int dim,
                          int nl, const DbLayout &L, int ng) {
    const int nblk = ng > 0 ? G / ng : 0;
    const int Gq = G / (BPP * NW);
    const int xm = (N / 128) % 8 == 0 ? 1 : 0;  // XCD-aware tile -> workgroup map
    const unsigned char *dbb = (const unsigned char *)db;
    const dim3 blk(64 * NW);
    const bool h24 = L.packed && L.seq && dim <= 4096;  // the 24-bit-halves kernel: its partial sums hold 4096 diagonals
    {   // resident database (6- or 8-byte residues) + rotated queries once + accumulators, split limb 0 / other limbs like the launches
        const double per_lp6 = (double)N * (L.bits46 ? 5.75 : 6.0), per_lp8 = LP_BYTES(N);
        const double rot_acc = (double)dim * 2 * per_lp8 + (double)G * 3 * per_lp8;
        char n0[64], n1[64];
        snprintf(n0, sizeof n0, "k_hydia_tensor<%d, %d, true, false>", BPP, NW);
        snprintf(n1, sizeof n1, h24 ? (L.bits46 ? "k_hydia_tensor24<%d, %d, true, true>" : "k_hydia_tensor24<%d, %d, false, false>") : "k_hydia_tensor<%d, %d, true, true>", BPP, NW);  // as rocprofv3 prints the instantiation
        if (L.packed && G <= 8) snprintf(n0, sizeof n0, "k_hydia_tensor_sk<%d, false>", G <= 2 ? 8 : 4);
        if (L.packed) {
            ledger_add(n0, (double)G * dim * 2 * per_lp8 + rot_acc);
            if (nl > 1) ledger_add(n1, (nl - 1) * ((double)G * dim * 2 * per_lp6 + rot_acc));
        } else {
            ledger_add(n0, nl * ((double)G * dim * 2 * per_lp8 + rot_acc));
        }
    }
    if (L.packed) {  // limb 0 (8-byte residues) and limbs 1.. (6-byte residues) as two launches: no shared register budget
        if (G <= 2)  // few blocks: 256 x G one-wave workgroups cannot hide the latency of 512 dependent steps -> split the diagonals
            // (eight waves: sixteen hold a lane to 128 registers and the kernel spilled 69 of them — round 5)
            hipLaunchKernelGGL((k_hydia_tensor_sk<8, false>), dim3((N / 128) * G, 1), dim3(64 * 8), 0, st, mod, N, rot, dbb, acc, dim, nl, L, 0, ng, nblk);
        else if (G <= 8)
            hipLaunchKernelGGL((k_hydia_tensor_sk<4, false>), dim3((N / 128) * G, 1), dim3(64 * 4), 0, st, mod, N, rot, dbb, acc, dim, nl, L, 0, ng, nblk);
        else
            hipLaunchKernelGGL((k_hydia_tensor<BPP, NW, true, false>), dim3((N / 128) * Gq, 1), blk, 0, st, mod, N, rot, dbb, acc, dim, nl,
                               Gq, xm, L, 0, ng, nblk);
        if (nl > 1 && h24)
            if (L.bits46)
                hipLaunchKernelGGL((k_hydia_tensor24<BPP, NW, true>), dim3((N / 128) * Gq, nl - 1), blk, 0, st, mod, N, rot, dbb, acc, dim, nl, Gq, xm, L,
                                   1, ng, nblk);
            else
                hipLaunchKernelGGL((k_hydia_tensor24<BPP, NW, false>), dim3((N / 128) * Gq, nl - 1), blk, 0, st, mod, N, rot, dbb, acc, dim, nl, Gq, xm, L,
                                   1, ng, nblk);
        else if (nl > 1)
            hipLaunchKernelGGL((k_hydia_tensor<BPP, NW, true, true>), dim3((N / 128) * Gq, nl - 1), blk, 0, st, mod, N, rot, dbb, acc,
                               dim, nl, Gq, xm, L, 1, ng, nblk);
    } else {
        hipLaunchKernelGGL((k_hydia_tensor<BPP, NW, true, false>), dim3((N / 128) * Gq, nl), blk, 0, st, mod, N, rot, dbb, acc, dim, nl,
                           Gq, xm, L, 0, ng, nblk);
    }
}
// bpp = database blocks per wave (1 or 2), nw = max waves per workgroup (1, 2 or 4; 0 = 4); both must divide G.  Round 5: four blocks
// per wave and eight / sixteen waves per workgroup are gone — they spilled (863 registers at <4,16>, 237 at <4,8>) and were never
// faster (profiles/r04/experiments.txt: BPP=4 140.5 ms, NW=8 inside the run-to-run spread); larger values are clamped.
void tensor_split(int G, int bpp, int nw, int *Bo, int *Wo) {
    const int B = (bpp >= 2 && G % 2 == 0) ? 2 : 1;
    const int rest = G / B;
    if (nw == 0 || nw > 4) nw = 4;
    int W = 1;
    for (int cand : {4, 2})
        if ((nw == 0 || cand <= nw) && rest % cand == 0) {
            W = cand;
            break;
        }
    *Bo = B;
    *Wo = W;
}
void hydia_tensor_accumulate(hipStream_t st, const ModC *mod, int N, const u64 *rot, const void *db, u64 *acc, int G,
                             int dim, int nl, int bpp, int nw, const DbLayout &L, int ng) {
    int B, W;
    tensor_split(G, bpp, nw, &B, &W);
    if (L.seq) {  // the layout fixes the workgroup's share: a group of the database is what one workgroup walks
        if (L.bits46 && !(L.packed && dim <= 4096)) throw std::logic_error("hydia: 46-bit database outside the 24-bit-halves loop B");
        if (G != L.blocks || dim != L.bd || L.seq % L.seq_bpp || G % L.seq || G <= 8)
            throw std::logic_error("hydia: loop B launched against a group-sequential database with another shape");
        B = L.seq_bpp;
        W = L.seq / L.seq_bpp;
    }
#define HY_TENSOR_CASE(b, w) \
    if (B == b && W == w) return launch_tensor<b, w>(st, mod, N, rot, db, acc, G, dim, nl, L, ng);
    HY_TENSOR_CASE(2, 4) HY_TENSOR_CASE(2, 2) HY_TENSOR_CASE(2, 1)
    HY_TENSOR_CASE(1, 4) HY_TENSOR_CASE(1, 2) HY_TENSOR_CASE(1, 1)
#undef HY_TENSOR_CASE
    throw std::logic_error("hydia: no loop B kernel for this split");
}
DbLayout db_layout(int N, int nQ, int packed) {
    DbLayout L{};
    L.packed = packed;
    L.poly_bytes = packed ? (unsigned long long)N * 8 + (unsigned long long)(nQ - 1) * N * 6 : (unsigned long long)nQ * N * 8;
    L.ct_bytes = 2 * L.poly_bytes;
    return L;
}
// group-sequential for `blocks` blocks of bd ciphertexts: only where loop B is a stream worth shaping (more than 8 blocks, whole
// 128-residue tiles) — otherwise the ciphertext-major layout comes back
DbLayout db_layout_seq(int N, int nQ, int packed, int bd, int blocks, int bpp, int nw, bool bits46) {
    DbLayout L = db_layout(N, nQ, packed);
    if (blocks <= 8 || N % 128) return L;
    int B, W;
    tensor_split(blocks, bpp, nw, &B, &W);
    L.seq = B * W;
    L.seq_bpp = B;
    L.bd = bd;
    L.blocks = blocks;
    if (bits46 && packed && bd <= 4096) {  // 46-bit residues for the packed limbs (the caller vouches for the moduli)
        L.bits46 = 1;
        L.poly_bytes = (unsigned long long)N * 8 + (unsigned long long)(nQ - 1) * (N / 128) * 736;
        L.ct_bytes = 2 * L.poly_bytes;
    }
    return L;
}
void db_pack(hipStream_t st, int N, int nQ, const u64 *plain, void *db, size_t t0, int X, const DbLayout &L) {
    const bool b46 = L.bits46 && L.seq && L.packed && nQ > 1;  // limb 0 through the pair kernel, the 46-bit limbs through the granule kernel
    hipLaunchKernelGGL(k_db_repack<true>, dim3(N / 512, b46 ? 1 : nQ, X * 2), dim3(256), 0, st, N, nQ, const_cast<u64 *>(plain),
                       (unsigned char *)db, L, t0);
    if (b46)
        hipLaunchKernelGGL(k_db_repack46<true>, dim3((N / 16 + 255) / 256, nQ - 1, X * 2), dim3(256), 0, st, N, nQ, const_cast<u64 *>(plain), (unsigned char *)db, L, t0);
}
void db_unpack(hipStream_t st, int N, int nQ, u64 *plain, const void *db, size_t t0, int X, const DbLayout &L) {
    const bool b46 = L.bits46 && L.seq && L.packed && nQ > 1;
    hipLaunchKernelGGL(k_db_repack<false>, dim3(N / 512, b46 ? 1 : nQ, X * 2), dim3(256), 0, st, N, nQ, plain, (unsigned char *)db, L, t0);
    if (b46) hipLaunchKernelGGL(k_db_repack46<false>, dim3((N / 16 + 255) / 256, nQ - 1, X * 2), dim3(256), 0, st, N, nQ, plain, (unsigned char *)db, L, t0);
}
const char *hydia_tensor_kernel_name() { return "k_hydia_tensor"; }
void fill_uniform_hash(hipStream_t st, const ModC *mod, int N, u64 *dst, size_t n_limbpolys, int nl,
                       unsigned long long seed) {
    // slabs of at most 32768 limb-polys (grid.y limit), each starting on a limb-0 boundary
    const size_t slab = 32768 - 32768 % nl;
    for (size_t done = 0; done < n_limbpolys; done += slab) {
        const size_t cnt = n_limbpolys - done < slab ? n_limbpolys - done : slab;
        hipLaunchKernelGGL(k_fill_uniform_hash, dim3(N / 256, (unsigned)cnt, 1), dim3(256), 0, st, mod, N,
                           dst + done * N, nl, seed + done * 0x51ED27ull);
    }
}

}  // namespace hk

// tools/ubench/int_rates.hip — issue rate of the integer vector instructions the 60-bit butterflies are made of (gfx950), and of the two
// whole butterflies (Shoup / Harvey against the pseudo-Mersenne fold for q = 2^60 - c).  Result = lane-operations per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/int_rates.hip -o tools/ubench/int_rates && tools/ubench/int_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned __int128 u128;

__device__ __forceinline__ u64 mulfold(u64 b, u64 w, unsigned c) {
    const u128 P = (u128)b * w;
    const u64 MASK = (1ull << 60) - 1;
    const u64 Pl = (u64)P & MASK, Ph = (u64)(P >> 60);
    const u128 r1 = (u128)Ph * c + Pl;
    const u64 r1l = (u64)r1 & MASK;
    const unsigned r1h = (unsigned)(r1 >> 60);
    return (u64)r1h * c + r1l;
}

template <int OP, int ILP>
__global__ __launch_bounds__(256) void k(u64 *out, int iters, u64 a, u64 b, u64 q, unsigned c) {
    u64 x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = a + threadIdx.x * 977 + i, y[i] = b + threadIdx.x + 3 * i;
    const u64 q2 = 2 * q;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) x[i] = (unsigned)x[i] * (unsigned)a + 1u;                                         // v_mul_lo_u32 (+ add)
            else if (OP == 1) x[i] = __umulhi((unsigned)x[i], (unsigned)a) + 1u;                          // v_mul_hi_u32 (+ add)
            else if (OP == 2) x[i] = (u64)(unsigned)x[i] * (unsigned)a + y[i];                             // v_mad_u64_u32
            else if (OP == 3) x[i] = x[i] + y[i];                                                          // 64-bit add
            else if (OP == 4) x[i] = x[i] >= q2 ? x[i] - q2 : x[i];                                        // conditional subtraction
            else if (OP == 5) x[i] = __umul64hi(x[i], a);                                                  // 64 x 64 -> high 64
            else if (OP == 6) x[i] = x[i] * a;                                                             // 64 x 64 -> low 64
            else if (OP == 7) {                                                                            // Harvey butterfly, Shoup multiply
                const u64 u = x[i] >= q2 ? x[i] - q2 : x[i];
                const u64 hi = __umul64hi(y[i], b);
                const u64 t = y[i] * a - hi * q;
                x[i] = u + t;
                y[i] = u - t + q2;
            } else if (OP == 8) {                                                                          // same, pseudo-Mersenne fold
                const u64 u = x[i] >= q2 ? x[i] - q2 : x[i];
                const u64 t = mulfold(y[i], a, c);
                x[i] = u + t;
                y[i] = u - t + q2;
            } else if (OP == 9) {                                                                          // fold, `a` reduced lazily (not here)
                const u64 t = mulfold(y[i], a, c);
                const u64 u = x[i];
                x[i] = u + t;
                y[i] = u - t + q2;
            } else if (OP == 10) x[i] = (u64)(((unsigned)x[i] & 0xffffffu) * ((unsigned)a & 0xffffffu)) + 1u;   // v_mul_u32_u24 (the compiler sees both operands masked to 24 bits)
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i] ^ y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name) {
    const int iters = 2048, ILP = 4, threads = 256, blocks = 256 * 8;
    u64 *out;
    hipMalloc(&out, sizeof(u64) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const u64 q = 0xffffffffffc0001ull;
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(threads), 0, 0, out, 16, 0x123456789abcdefull, 0xfedcba987654321ull, q, 0x3ffffu);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0x123456789abcdefull, 0xfedcba987654321ull, q, 0x3ffffu);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * threads * iters * ILP;
    printf("%-46s %8.3f ms  %7.2f per clk per CU  (%6.1f SIMD-cycles per wave-op)\n", name, ms, n / (ms * 1e-3) / 2.4e9 / 256,
           (ms * 1e-3) * 2.4e9 * 256 * 4 / (n / 64));
    hipFree(out);
}
int main() {
    run<0>("v_mul_lo_u32 (+ v_add)");
    run<1>("v_mul_hi_u32 (+ v_add)");
    run<2>("v_mad_u64_u32");
    run<3>("64-bit add");
    run<4>("conditional subtraction (64-bit)");
    run<5>("__umul64hi");
    run<6>("64 x 64 -> low 64");
    run<7>("butterfly: Harvey + Shoup");
    run<8>("butterfly: Harvey + fold (q = 2^60 - c)");
    run<9>("butterfly: fold, no conditional subtraction");
    run<10>("v_mul_u32_u24 (+ v_add)");
    return 0;
}

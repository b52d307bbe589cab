// tools/ubench/fp64_rates.hip — issue rate of the FP64 vector instructions the NTT butterflies are made of (gfx950).
// Each wave runs ILP independent dependency chains of one instruction; result = lane-operations per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/fp64_rates.hip -o tools/ubench/fp64_rates && tools/ubench/fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int ILP>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = a + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) x[i] = __fma_rn(x[i], a, b);
            else if (OP == 1) x[i] = x[i] * a;
            else if (OP == 2) x[i] = x[i] + b;
            else if (OP == 3) x[i] = rint(x[i]) + 0.25;  // rndne + add (add rate known from OP 2)
            else if (OP == 4) x[i] = (x[i] + 6755399441055744.0) - 6755399441055744.0;  // magic-number rounding: 2 adds
            else if (OP == 5) {  // the butterfly's mulmod: mul, fma, mul, rint, fma, add
                const double h = x[i] * a, l = __fma_rn(x[i], a, -h), c = rint(x[i] * b);
                x[i] = __fma_rn(-c, 1234567.0, h) + l;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int ops_per_iter, int threads) {
    const int iters = 4096, ILP = 8, blocks = 256 * (1024 / threads) * 1;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(threads), 0, 0, out, 16, 1.0000001, 0.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, ILP>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double laneops = (double)blocks * threads * iters * ILP * ops_per_iter;
    printf("%-34s threads/WG %4d: %7.3f ms  %7.1f lane-instr/clk/CU (at 2.4 GHz, 256 CUs)\n", name, threads, ms, laneops / (ms * 1e-3) / 2.4e9 / 256);
    hipFree(out);
}
int main() {
    for (int threads : {256, 1024}) {
        run<0>("v_fma_f64", 1, threads);
        run<1>("v_mul_f64", 1, threads);
        run<2>("v_add_f64", 1, threads);
        run<3>("v_rndne_f64 + v_add_f64", 2, threads);
        run<4>("magic round (2 x v_add_f64)", 2, threads);
        run<5>("mulmod (mul fma mul rint fma add)", 6, threads);
    }
    return 0;
}

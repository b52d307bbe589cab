// Micro-benchmark: what bandwidth does the access pattern of NTT pass 1 allow, independent of arithmetic?
// A limb-poly is 128 rows x 256 columns of u64; a workgroup owns a column tile (32 or 64 columns) of all 128 rows and copies it.
//   hipcc --offload-arch=gfx950 -O3 -o p1_pattern p1_pattern.hip && ./p1_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 32-column tiles, 8 bytes per lane (the current pass-1 pattern): grid (8, polys), 256 threads
__global__ __launch_bounds__(256) void k_tile32(const u64 *__restrict__ s, u64 *__restrict__ d) {
    const size_t base = (size_t)blockIdx.y * 32768 + blockIdx.x * 32;
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = s[base + (size_t)(g + 8 * k) * 256 + col];
#pragma unroll
    for (int k = 0; k < 16; k++) d[base + (size_t)(g + 8 * k) * 256 + col] = v[k] + 1;
}
// 64-column tiles, 16 bytes per lane: grid (4, polys), 256 threads, 32 values per thread
__global__ __launch_bounds__(256) void k_tile64(const u64 *__restrict__ s, u64 *__restrict__ d) {
    const size_t base = (size_t)blockIdx.y * 32768 + blockIdx.x * 64;
    const int col = (threadIdx.x & 31) * 2, g = threadIdx.x >> 5;
    ulonglong2 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = *reinterpret_cast<const ulonglong2 *>(s + base + (size_t)(g + 8 * k) * 256 + col);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k].x += 1;
        *reinterpret_cast<ulonglong2 *>(d + base + (size_t)(g + 8 * k) * 256 + col) = v[k];
    }
}
// 32-column tiles but 16 bytes per lane (16 lanes per row, 4 rows per wave instruction): grid (8, polys), 256 threads, 8 x 2 values
__global__ __launch_bounds__(256) void k_tile32w(const u64 *__restrict__ s, u64 *__restrict__ d) {
    const size_t base = (size_t)blockIdx.y * 32768 + blockIdx.x * 32;
    const int col = (threadIdx.x & 15) * 2, g = threadIdx.x >> 4;  // 16 row groups
    ulonglong2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const ulonglong2 *>(s + base + (size_t)(g + 16 * k) * 256 + col);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        v[k].x += 1;
        *reinterpret_cast<ulonglong2 *>(d + base + (size_t)(g + 16 * k) * 256 + col) = v[k];
    }
}
// contiguous copy, 16 bytes per lane: grid (16, polys), 256 threads x 8
__global__ __launch_bounds__(256) void k_contig(const u64 *__restrict__ s, u64 *__restrict__ d) {
    const size_t base = (size_t)blockIdx.y * 32768 + (size_t)blockIdx.x * 2048;
    ulonglong2 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const ulonglong2 *>(s + base + k * 512 + threadIdx.x * 2);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k].x += 1;
        *reinterpret_cast<ulonglong2 *>(d + base + k * 512 + threadIdx.x * 2) = v[k];
    }
}
int main() {
    const int polys = 12288;  // 3 GiB in, 3 GiB out
    const size_t n = (size_t)polys * 32768;
    u64 *a, *b;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 1, n * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.3f ms per 6 GiB  -> %.2f TB/s\n", name, ms / 5, 2.0 * n * 8 / (ms / 5) / 1e9);
    };
    run("tile 32 cols, 8 B/lane (current)", [&] { hipLaunchKernelGGL(k_tile32, dim3(8, polys), dim3(256), 0, 0, a, b); });
    run("tile 32 cols, 16 B/lane", [&] { hipLaunchKernelGGL(k_tile32w, dim3(8, polys), dim3(256), 0, 0, a, b); });
    run("tile 64 cols, 16 B/lane", [&] { hipLaunchKernelGGL(k_tile64, dim3(4, polys), dim3(256), 0, 0, a, b); });
    run("contiguous, 16 B/lane", [&] { hipLaunchKernelGGL(k_contig, dim3(16, polys), dim3(256), 0, 0, a, b); });
    return 0;
}

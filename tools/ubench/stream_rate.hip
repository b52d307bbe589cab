// tools/ubench/stream_rate.hip — what HBM gives a read-once stream on gfx950, by how the loads are issued:
//   mode 0  global_load_dwordx4 nt into registers, 8 loads (8 KiB) per wave in flight, software-pipelined one group ahead
//   mode 1  global_load_lds_dwordx4 (LDS-DMA) into a per-wave two-stage LDS ring, nobody reads the data (the DMA ceiling)
//   mode 2  the same with the wave reading its stage back (ds_read_b128) and folding it
// and by access pattern: seq = every wave sweeps its own contiguous region; chunks = eight sub-streams per wave (a group's pieces far
// apart); sweep = all waves walk the buffer together, 1 KiB each per step (loop B: one tile per workgroup, one diagonal per step).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_rate.hip -o tools/ubench/stream_rate && tools/ubench/stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((__attribute__((address_space(1))) void *)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void *)(p))

// piece k of group `it` of this wave: seq -> base + (it*8 + k) KiB; chunks -> 8 pieces `stride` apart, the next group 1 KiB further on
__device__ inline const u4 *piece(const u4 *base, long it, int k, long stride16, int pattern) {
    return pattern == 0 ? base + (it * 8 + k) * 64 : pattern == 1 ? base + (long)k * stride16 + it * 64 : base + (it * 8 + k) * stride16;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const u4 *__restrict__ src, long groups, long wave_stride16, long stride16, int pattern,
                                                unsigned *__restrict__ out) {
    extern __shared__ u4 lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 4 + w;
    const u4 *base = src + gw * wave_stride16 + lane;
    u4 acc = {0, 0, 0, 0};
    if (MODE == 3) {  // mode 0 with 12-byte loads: pieces of 768 B (the packed residues' width), same addresses otherwise
        typedef unsigned int v3 __attribute__((ext_vector_type(3), aligned(4)));
        const unsigned char *b3 = (const unsigned char *)(base - lane) + lane * 12;
        v3 cur[8], nxt[8], a3 = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load((const v3 *)(b3 + ((const unsigned char *)piece(base, 0, k, stride16, pattern) - (const unsigned char *)base) / 4 * 3));
        for (long it = 0; it < groups; it += 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) nxt[k] = __builtin_nontemporal_load((const v3 *)(b3 + ((const unsigned char *)piece(base, it + 1, k, stride16, pattern) - (const unsigned char *)base) / 4 * 3));
#pragma unroll
            for (int k = 0; k < 8; k++) a3 ^= cur[k];
            const long it2 = it + 2 < groups ? it + 2 : it + 1;
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load((const v3 *)(b3 + ((const unsigned char *)piece(base, it2, k, stride16, pattern) - (const unsigned char *)base) / 4 * 3));
#pragma unroll
            for (int k = 0; k < 8; k++) a3 ^= nxt[k];
        }
        acc.x = a3.x; acc.y = a3.y; acc.z = a3.z;
    } else if (MODE == 0) {
        u4 cur[8], nxt[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load(piece(base, 0, k, stride16, pattern));
        for (long it = 0; it < groups; it += 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) nxt[k] = __builtin_nontemporal_load(piece(base, it + 1, k, stride16, pattern));
#pragma unroll
            for (int k = 0; k < 8; k++) acc ^= cur[k];
            const long it2 = it + 2 < groups ? it + 2 : it + 1;
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load(piece(base, it2, k, stride16, pattern));
#pragma unroll
            for (int k = 0; k < 8; k++) acc ^= nxt[k];
        }
    } else {
        u4 *my = lds + w * 2 * 512;  // two stages of 8 KiB per wave
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_amdgcn_global_load_lds(GPTR(piece(base, 0, k, stride16, pattern)), LPTR(my + k * 64), 16, 0, 2);
        for (long it = 0; it < groups; it++) {
            const int st = (int)(it & 1);
            const long nx = it + 1 < groups ? it + 1 : it;
#pragma unroll
            for (int k = 0; k < 8; k++)
                __builtin_amdgcn_global_load_lds(GPTR(piece(base, nx, k, stride16, pattern)), LPTR(my + (st ^ 1) * 512 + k * 64), 16, 0, 2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage `st` has landed
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= my[st * 512 + k * 64 + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the stage is refilled two iterations on
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[gw] = 1;  // keeps the loads alive
}

template <int MODE>
double run(const u4 *src, size_t bytes, int pattern, unsigned *out) {
    const int waves = 256 * 2 * 4;
    const long per_wave = (long)(bytes / waves), groups = per_wave / 8192;
    long wave_stride16 = per_wave / 16, stride16 = 64;
    if (pattern == 1) stride16 = per_wave / 8 / 16;  // the wave's region cut into 8 sub-streams: pieces of one group are far apart
    if (pattern == 2) wave_stride16 = 64, stride16 = (long)waves * 64;  // all waves sweep the buffer together, 1 KiB each per step (loop B today)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_stream<MODE>), dim3(waves / 4), dim3(256), 4 * 2 * 8192, 0, src, groups, wave_stride16,
                           stride16, pattern, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double moved = (double)groups * (MODE == 3 ? 6144 : 8192) * waves;
    return moved / (best * 1e-3) / 1e12;
}

// loop B's own address pattern, no arithmetic: workgroup = (tile of 128 coefficients, packed limb, group of 8 blocks), 4 waves x 2 blocks,
// per diagonal each wave reads 2 blocks x 2 polynomials x 768 B (dwordx3 nt) one diagonal ahead; layout as resident: ciphertext t at
// t*ct_bytes, polynomial at +poly_bytes, limb j >= 1 at N*8 + (j-1)*N*6.  LAYOUT 1 = the same bytes stored tile-major instead:
// [block][limb][tile][diagonal][polynomial][128 x 6 B], one contiguous 786 KiB run per workgroup and block.
typedef unsigned int u3 __attribute__((ext_vector_type(3), aligned(4)));
template <int LAYOUT, bool BARRIER, int D = 1>
__global__ __launch_bounds__(256, 2) void k_loopb(const unsigned char *__restrict__ db, int G, int dim, unsigned *__restrict__ out) {
    constexpr long N = 32768, poly_bytes = N * 8 + 11 * N * 6, ct_bytes = 2 * poly_bytes;
    const int Gq = G / 8, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int gq = k % Gq, tile = xcd + 8 * (k / Gq), j = blockIdx.y + 1;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g0 = (gq * 4 + wv) * 2;
    const unsigned char *p[2][2];
    long step;
    if (LAYOUT == 0) {
        for (int u = 0; u < 2; u++)
            for (int q = 0; q < 2; q++) p[u][q] = db + (long)(g0 + u) * dim * ct_bytes + q * poly_bytes + N * 8 + (long)(j - 1) * N * 6 + ((long)tile * 128 + lane * 2) * 6;
        step = ct_bytes;
    } else if (LAYOUT == 1) {
        for (int u = 0; u < 2; u++)
            for (int q = 0; q < 2; q++) p[u][q] = db + ((((long)(g0 + u) * 11 + (j - 1)) * 256 + tile) * dim * 2 + q) * 768 + lane * 12;
        step = 2 * 768;
    } else {  // [limb][tile][group of 8 blocks][diagonal][block in group][polynomial]: ONE sequential run of 6 MiB per workgroup
        for (int u = 0; u < 2; u++)
            for (int q = 0; q < 2; q++) p[u][q] = db + ((((((long)(j - 1) * 256 + tile) * Gq + gq) * dim * 8) + (wv * 2 + u)) * 2 + q) * 768 + lane * 12;
        step = 8 * 2 * 768;
    }
    if (LAYOUT == 3) {  // layout 2 with 46-bit residues: units of 736 B, lane l's two residues at bit 92 l -> a 16-byte load from dword floor(2.875 l)
        for (int u = 0; u < 2; u++)
            for (int q = 0; q < 2; q++) p[u][q] = db + ((((((long)(j - 1) * 256 + tile) * Gq + gq) * dim * 8) + (wv * 2 + u)) * 2 + q) * 736 + ((lane * 92) >> 5) * 4;
        step = 8 * 2 * 736;
        typedef unsigned int u4a __attribute__((ext_vector_type(4), aligned(4)));
        u4a c4[4], n4[4], a4 = {0, 0, 0, 0};
        auto fetch4 = [&](u4a *o, int i) {
#pragma unroll
            for (int x = 0; x < 4; x++) o[x] = __builtin_nontemporal_load((const u4a *)(p[x >> 1][x & 1] + (long)i * step));
        };
        fetch4(c4, 0);
        for (int i = 0; i < dim; i += 2) {
            fetch4(n4, i + 1);
#pragma unroll
            for (int x = 0; x < 4; x++) a4 ^= c4[x];
            if (BARRIER) __builtin_amdgcn_s_barrier();
            fetch4(c4, i + 2 < dim ? i + 2 : i + 1);
#pragma unroll
            for (int x = 0; x < 4; x++) a4 ^= n4[x];
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
        if ((a4.x ^ a4.y ^ a4.z ^ a4.w) == 0x12345678u) out[blockIdx.x] = 1;
        return;
    }
    u3 cur[4 * D], nxt[4 * D], acc = {0, 0, 0};
    auto fetch = [&](u3 *o, int i) {  // D consecutive diagonals from i
#pragma unroll
        for (int dd = 0; dd < D; dd++)
#pragma unroll
            for (int x = 0; x < 4; x++) o[dd * 4 + x] = __builtin_nontemporal_load((const u3 *)(p[x >> 1][x & 1] + (long)(i + dd) * step));
    };
    fetch(cur, 0);
    for (int i = 0; i < dim; i += 2 * D) {
        fetch(nxt, i + D);
#pragma unroll
        for (int x = 0; x < 4 * D; x++) acc ^= cur[x];
        if (BARRIER) __builtin_amdgcn_s_barrier();
        fetch(cur, i + 2 * D < dim ? i + 2 * D : i + D);
#pragma unroll
        for (int x = 0; x < 4 * D; x++) acc ^= nxt[x];
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    if ((acc.x ^ acc.y ^ acc.z) == 0x12345678u) out[blockIdx.x] = 1;
}
template <int LAYOUT, bool BARRIER, int D = 1>
double run_loopb(const unsigned char *db, int G, unsigned *out, int wgs_per_cu = 8) {
    const int dim = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_loopb<LAYOUT, BARRIER, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_loopb<LAYOUT, BARRIER, D>), dim3(256 * (G / 8), 11), dim3(256), 160 * 1024 / wgs_per_cu - 512, 0, db, G, dim, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (LAYOUT == 3) printf("    [46-bit layout: %.3f ms per pass]\n", best);
    else if (LAYOUT == 2) printf("    [48-bit workgroup-sequential layout: %.3f ms per pass]\n", best);
    return (double)G * dim * 2 * 11 * 32768 * (LAYOUT == 3 ? 5.75 : 6) / (best * 1e-3) / 1e12;
}

int main() {
    const size_t bytes = (size_t)16 << 30;
    u4 *src;
    unsigned *out;
    if (hipMalloc((void **)&src, bytes) != hipSuccess) return 1;
    hipMalloc((void **)&out, 1 << 20);
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    for (int pattern = 0; pattern < 3; pattern++) {
        printf("%-7s registers nt %.2f TB/s (12-byte loads %.2f) | LDS-DMA, unread %.2f TB/s | LDS-DMA + ds_read %.2f TB/s\n", pattern == 0 ? "seq" : pattern == 1 ? "chunks" : "sweep",
               run<0>(src, bytes, pattern, out), run<3>(src, bytes, pattern, out), run<1>(src, bytes, pattern, out), run<2>(src, bytes, pattern, out));
    }
    (void)hipFree(src);
    // loop B's pattern on G blocks of the resident layout (packed limbs only: 2.16 of the 2.31 GiB of a block are read)
    const int G = 48;
    unsigned char *db;
    const size_t db_bytes = (size_t)G * 512 * 2 * (32768 * 8 + 11 * 32768 * 6);
    if (hipMalloc((void **)&db, db_bytes) != hipSuccess) return 1;
    (void)hipMemset(db, 1, db_bytes);
    (void)hipDeviceSynchronize();
    printf("loop B pattern, %d blocks: resident layout %.2f TB/s (no barrier %.2f) | tile-major layout %.2f TB/s (no barrier %.2f)\n", G,
           run_loopb<0, true>(db, G, out), run_loopb<0, false>(db, G, out), run_loopb<1, true>(db, G, out), run_loopb<1, false>(db, G, out));
    printf("  diagonals per step (one step ahead): 2: resident %.2f tile-major %.2f | 4: resident %.2f tile-major %.2f | 8: tile-major %.2f TB/s\n",
           run_loopb<0, true, 2>(db, G, out), run_loopb<1, true, 2>(db, G, out), run_loopb<0, true, 4>(db, G, out), run_loopb<1, true, 4>(db, G, out),
           run_loopb<1, true, 8>(db, G, out));
    printf("  workgroup-sequential layout: %.2f TB/s (no barrier %.2f; 2 diagonals per step %.2f)\n", run_loopb<2, true>(db, G, out, 3), run_loopb<2, false>(db, G, out, 3), run_loopb<2, true, 2>(db, G, out, 3));
    printf("  the same with 46-bit residues (units of 736 B, 16-byte loads at 11.5-byte lane stride): %.2f TB/s of distinct bytes (no barrier %.2f)\n",
           run_loopb<3, true>(db, G, out, 3), run_loopb<3, false>(db, G, out, 3));
    for (int w : {2, 3, 4, 6})
        printf("  at most %d workgroups per CU: resident %.2f TB/s | tile-major %.2f TB/s\n", w, run_loopb<0, true>(db, G, out, w), run_loopb<1, true>(db, G, out, w));
    return 0;
}

// tools/ubench/tensor_check.cpp — loop B's kernels against a host recomputation on random residues, small ring (debugging aid:
// it isolated a ROCm 7.2 miscompile of 24-bit multiply-accumulates in an experimental kernel).  Usage: tensor_check <blocks> <dim> [1 = group-sequential layout]
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I image_matching_amd/csrc -I include tools/ubench/tensor_check.cpp -L image_matching_amd -lhydia -Wl,-rpath,$PWD/image_matching_amd -o tools/ubench/tensor_check
#include <cstdio>
#include <vector>
#include "hydia_core.h"
using namespace hydia;
typedef unsigned __int128 u128_t;
int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2, dim = argc > 2 ? atoi(argv[2]) : 8;
    Params p;
    p.logN = 11;
    p.dim = dim;
    Context cx(p, 0);
    const int N = cx.N, nl = cx.nQ;
    const size_t ct = (size_t)2 * nl * N;
    u64 *rot, *plain, *acc;
    void *db;
    const bool seq = argc > 3 && atoi(argv[3]) != 0;  // third argument 1: the group-sequential layout (more than 8 blocks)
    const DbLayout L = seq ? hk::db_layout_seq(N, nl, 1, dim, G, 2, 4) : hk::db_layout(N, nl, 1);
    printf("layout: %s (group of %d blocks)\n", L.seq ? "group-sequential" : "ciphertext-major", L.seq);
    hipMalloc((void **)&rot, ct * dim * 8);
    hipMalloc((void **)&plain, ct * dim * G * 8);
    hipMalloc(&db, (size_t)L.ct_bytes * dim * G);
    hipMalloc((void **)&acc, (size_t)G * 3 * nl * N * 8);
    hk::fill_uniform_hash(cx.stream, cx.d_mod, N, rot, (size_t)2 * nl * dim, nl, 11);
    hk::fill_uniform_hash(cx.stream, cx.d_mod, N, plain, (size_t)2 * nl * dim * G, nl, 12);
    hk::db_pack(cx.stream, N, nl, plain, db, 0, dim * G, L);
    hk::hydia_tensor_accumulate(cx.stream, cx.d_mod, N, rot, db, acc, G, dim, nl, 2, 4, L, 0);
    hipStreamSynchronize(cx.stream);
    std::vector<u64> hr(ct * dim), hp(ct * dim * G), ha((size_t)G * 3 * nl * N);
    hipMemcpy(hr.data(), rot, hr.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hp.data(), plain, hp.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), acc, ha.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int g = 0; g < G; g++)
        for (int j = 0; j < nl; j++)
            for (int c = 0; c < N; c++) {
                const u64 q = cx.q[j];
                u128_t d0 = 0, d1 = 0, d2 = 0;
                for (int i = 0; i < dim; i++) {
                    const u64 a0 = hr[((size_t)i * 2 + 0) * nl * N + (size_t)j * N + c], a1 = hr[((size_t)i * 2 + 1) * nl * N + (size_t)j * N + c];
                    const u64 b0 = hp[(((size_t)g * dim + i) * 2 + 0) * nl * N + (size_t)j * N + c], b1 = hp[(((size_t)g * dim + i) * 2 + 1) * nl * N + (size_t)j * N + c];
                    d0 = (d0 + (u128_t)a0 * b0 % q) % q;
                    d1 = (d1 + (u128_t)a0 * b1 % q + (u128_t)a1 * b0 % q) % q;
                    d2 = (d2 + (u128_t)a1 * b1 % q) % q;
                }
                const u64 g0 = ha[((size_t)g * 3 + 0) * nl * N + (size_t)j * N + c], g1 = ha[((size_t)g * 3 + 1) * nl * N + (size_t)j * N + c], g2 = ha[((size_t)g * 3 + 2) * nl * N + (size_t)j * N + c];
                if (g0 != (u64)d0 || g1 != (u64)d1 || g2 != (u64)d2) {
                    if (bad < 6) printf("mismatch g %d limb %d c %d: got %llu %llu %llu want %llu %llu %llu\n", g, j, c, (unsigned long long)g0, (unsigned long long)g1, (unsigned long long)g2, (unsigned long long)d0, (unsigned long long)d1, (unsigned long long)d2);
                    bad++;
                }
            }
    printf("G %d dim %d: %ld mismatches of %ld\n", G, dim, bad, (long)G * nl * N);
    return bad != 0;
}

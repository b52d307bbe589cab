// tests/csrc/loop_b_check.cpp — loop B exactly as a query runs it (Context::similarity_accumulate_rot: the production layout choice,
// kernel choice and launch shape) against a host recomputation with unsigned __int128, on pseudo-random residues at the FULL ring.
// Test infrastructure (tests/test_gpu_full_ring.py builds and runs it; never part of the product).  Why it exists: ROCm 7.2 once
// miscompiled the 24-bit-halves kernel (k_hydia_tensor24) — it dropped the operand masks and fused unmasked registers back into
// v_mad_u64_u32 — and only a host recomputation shows that; a compiler bump could bring it back at dim 512.  What it replaces:
// 512 x EvalMultNoRelin + 511 x EvalAddInPlace per block, /root/reference/src/sender/sender_diag.cpp:70-77,:93.
// Usage: loop_b_check <blocks> <dim> <logN> <0 = ciphertext-major | 1 = the layout the context picks (group-sequential above 8 blocks,
//        46-bit residues) | 2 = group-sequential with 48-bit residues>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hydia_core.h"
using namespace hydia;
typedef unsigned __int128 u128_t;

static inline u64 hash_residue(u64 seed, size_t idx, u64 q) {  // k_fill_uniform_hash on the host
    u64 z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z % q;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 2, dim = argc > 2 ? atoi(argv[2]) : 8, logN = argc > 3 ? atoi(argv[3]) : 11;
    const int mode = argc > 4 ? atoi(argv[4]) : 1;
    const bool pick = mode != 0;
    Params p;
    p.logN = logN;
    p.dim = dim;
    Context cx(p, 0);
    const int N = cx.N, nl = cx.nQ;
    if (!pick) cx.db_seq_ok = false;
    if (mode == 2) cx.db_bits46_ok = false;
    const size_t cts = (size_t)G * dim, e = (size_t)2 * nl * N;
    if ((size_t)dim * 2 * nl > 32768 - 32768 % (size_t)nl) {
        printf("block too large for one fill launch\n");
        return 2;
    }
    cx.db_resize((size_t)G * cx.slots, cts, dim);
    cx.db_kind = 5;
    cx.db_babies = dim;
    printf("N = 2^%d, dim %d, %d blocks: %s layout (groups of %d blocks, %d-bit packed residues), %.2f GiB resident\n", logN, dim, G,
           cx.db_lay.seq ? "group-sequential" : "ciphertext-major", cx.db_lay.seq, cx.db_lay.bits46 ? 46 : 48, (double)cts * cx.db_lay.ct_bytes / (1 << 30));
    const u64 seed_rot = 11, seed_db = 1200;
    Ct rot(&cx, dim, 2, nl, cx.delta);
    hk::fill_uniform_hash(cx.stream, cx.d_mod, N, rot.d, (size_t)dim * 2 * nl, nl, seed_rot);
    u64 *tmp = cx.pool.get((size_t)dim * e * sizeof(u64));
    std::vector<u64> probe(4096);
    for (int g = 0; g < G; g++) {
        hk::fill_uniform_hash(cx.stream, cx.d_mod, N, tmp, (size_t)dim * 2 * nl, nl, seed_db + g);
        cx.db_store((size_t)g * dim, tmp, dim);
        if (g == G - 1) {  // the host mirror of the fill must be the device's: a stretch of the last block's last ciphertext, limb nl - 1
            cx.sync();
            const size_t off = (size_t)dim * e - probe.size();
            HIP_CHECK(hipMemcpy(probe.data(), tmp + off, probe.size() * 8, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < probe.size(); k++)
                if (probe[k] != hash_residue(seed_db + g, off + k, cx.q[nl - 1])) {
                    printf("host mirror of the fill differs from the device at %zu\n", k);
                    return 2;
                }
        }
    }
    cx.sync();
    cx.pool.put(tmp);
    Ct acc = cx.similarity_accumulate_rot(rot);
    cx.sync();
    std::vector<u64> ha((size_t)G * 3 * nl * N);
    HIP_CHECK(hipMemcpy(ha.data(), acc.d, ha.size() * 8, hipMemcpyDeviceToHost));
    // rotated queries on the host, once: [i][poly][limb][c]
    std::vector<u64> hr((size_t)dim * e);
    const unsigned T = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (size_t lp = t; lp < (size_t)dim * 2 * nl; lp += T)
                    for (int c = 0; c < N; c++) hr[lp * N + c] = hash_residue(seed_rot, lp * N + c, cx.q[lp % nl]);
            });
        for (auto &x : th) x.join();
    }
    std::atomic<long> bad{0};
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&] {
            for (;;) {
                const int w = next.fetch_add(1);  // one (block, limb) at a time
                if (w >= G * nl) break;
                const int g = w / nl, j = w % nl;
                const u64 q = cx.q[j];
                const bool wide = (q >> 50) != 0;  // the 60-bit limb: products reduced one by one (lazy sums would pass 2^128)
                for (int c = 0; c < N; c++) {
                    u128_t d0 = 0, d1 = 0, d2 = 0;
                    for (int i = 0; i < dim; i++) {
                        const u64 a0 = hr[((size_t)i * 2 + 0) * nl * N + (size_t)j * N + c], a1 = hr[((size_t)i * 2 + 1) * nl * N + (size_t)j * N + c];
                        const u64 b0 = hash_residue(seed_db + g, (((size_t)i * 2 + 0) * nl + j) * N + c, q);
                        const u64 b1 = hash_residue(seed_db + g, (((size_t)i * 2 + 1) * nl + j) * N + c, q);
                        if (wide) {
                            d0 += (u128_t)a0 * b0 % q;
                            d1 += (u128_t)a0 * b1 % q + (u128_t)a1 * b0 % q;
                            d2 += (u128_t)a1 * b1 % q;
                        } else {
                            d0 += (u128_t)a0 * b0;
                            d1 += (u128_t)a0 * b1 + (u128_t)a1 * b0;
                            d2 += (u128_t)a1 * b1;
                        }
                    }
                    const u64 w0 = (u64)(d0 % q), w1 = (u64)(d1 % q), w2 = (u64)(d2 % q);
                    const u64 g0 = ha[((size_t)g * 3 + 0) * nl * N + (size_t)j * N + c], g1 = ha[((size_t)g * 3 + 1) * nl * N + (size_t)j * N + c],
                              g2 = ha[((size_t)g * 3 + 2) * nl * N + (size_t)j * N + c];
                    if (g0 != w0 || g1 != w1 || g2 != w2) {
                        if (bad.fetch_add(1) < 6)
                            printf("mismatch block %d limb %d coefficient %d: got %llu %llu %llu want %llu %llu %llu\n", g, j, c, (unsigned long long)g0,
                                   (unsigned long long)g1, (unsigned long long)g2, (unsigned long long)w0, (unsigned long long)w1, (unsigned long long)w2);
                    }
                }
            }
        });
    for (auto &x : th) x.join();
    printf("loop B, %d blocks x %d diagonals at N = 2^%d: %ld mismatches of %ld (block, limb, coefficient) triples\n", G, dim, logN, bad.load(),
           (long)G * nl * N);
    return bad.load() != 0;
}

// Host check of the resident database's address map (image_matching_amd/csrc/db_layout.h): in both layouts every residue pair of
// every (ciphertext, polynomial, limb) gets its own bytes inside the allocation, a pair is contiguous, and in the group-sequential
// layout the bytes a loop-B workgroup reads (one 128-residue tile of one limb of one group of blocks, every diagonal, both
// polynomials) are ONE contiguous run.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "db_layout.h"

static DbLayout make(int N, int nQ, int packed, int bd, int blocks, int gs, int bpp) {
    DbLayout L{};
    L.packed = packed;
    L.poly_bytes = packed ? (unsigned long long)N * 8 + (unsigned long long)(nQ - 1) * N * 6 : (unsigned long long)nQ * N * 8;
    L.ct_bytes = 2 * L.poly_bytes;
    L.seq = gs;
    L.seq_bpp = bpp;
    L.bd = bd;
    L.blocks = blocks;
    return L;
}

static int check(int N, int nQ, int packed, int bd, int blocks, int gs) {
    const DbLayout L = make(N, nQ, packed, bd, blocks, gs, 1);
    const size_t cts = (size_t)bd * blocks, total = cts * L.ct_bytes;
    std::vector<unsigned char> used(total, 0);
    for (size_t t = 0; t < cts; t++)
        for (int p = 0; p < 2; p++)
            for (int j = 0; j < nQ; j++)
                for (size_t c = 0; c < (size_t)N; c += 2) {
                    const size_t es = (packed && j > 0) ? 6 : 8, o = db_offset(L, N, t, p, j, c);
                    if (o + 2 * es > total) return printf("out of range: t %zu p %d j %d c %zu\n", t, p, j, c), 1;
                    if (db_offset(L, N, t, p, j, c) + es != o + es) return 1;
                    for (size_t k = 0; k < 2 * es; k++) {
                        if (used[o + k]) return printf("overlap at byte %zu (t %zu p %d j %d c %zu)\n", o + k, t, p, j, c), 1;
                        used[o + k] = 1;
                    }
                }
    for (size_t k = 0; k < total; k++)
        if (!used[k]) return printf("hole at byte %zu\n", k), 1;
    if (gs) {  // one workgroup's bytes: limb j, tile, group -> [diagonal][block in group][polynomial][128 residues] back to back
        for (int j = 0; j < nQ; j++)
            for (int tile = 0; tile < N / 128; tile++)
                for (int grp = 0; grp < blocks / gs; grp++) {
                    const size_t es = (packed && j > 0) ? 6 : 8;
                    size_t expect = db_offset(L, N, (size_t)grp * gs * bd, 0, j, (size_t)tile * 128);
                    for (int i = 0; i < bd; i++)
                        for (int u = 0; u < gs; u++)
                            for (int p = 0; p < 2; p++) {
                                const size_t o = db_offset(L, N, ((size_t)grp * gs + u) * bd + i, p, j, (size_t)tile * 128);
                                if (o != expect) return printf("run broken: j %d tile %d grp %d i %d u %d p %d\n", j, tile, grp, i, u, p), 1;
                                expect += 128 * es;
                            }
                }
    }
    return 0;
}

int main() {
    int bad = 0;
    bad |= check(256, 3, 1, 4, 12, 0);  // ciphertext-major, packed
    bad |= check(256, 3, 0, 4, 12, 0);  // ciphertext-major, 8-byte
    bad |= check(256, 3, 1, 4, 12, 4);  // group-sequential, groups of 4
    bad |= check(256, 4, 1, 8, 16, 8);
    bad |= check(512, 2, 1, 2, 9, 1);   // degenerate groups of one block
    bad |= check(256, 3, 1, 4, 20, 2);
    if (!bad) printf("db layout ok\n");
    return bad;
}

// Host check of the resident database's address map (image_matching_amd/csrc/db_layout.h): in both layouts every residue pair of
// every (ciphertext, polynomial, limb) gets its own bytes inside the allocation, a pair is contiguous, and in the group-sequential
// layout the bytes a loop-B workgroup reads (one 128-residue tile of one limb of one group of blocks, every diagonal, both
// polynomials) are ONE contiguous run.
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "db_layout.h"

static DbLayout make(int N, int nQ, int packed, int bd, int blocks, int gs, int bpp, int bits46) {
    DbLayout L{};
    L.packed = packed;
    L.bits46 = (bits46 && gs && packed) ? 1 : 0;
    L.poly_bytes = packed ? (unsigned long long)N * 8 + (unsigned long long)(nQ - 1) * (L.bits46 ? (N / 128) * 736 : N * 6) : (unsigned long long)nQ * N * 8;
    L.ct_bytes = 2 * L.poly_bytes;
    L.seq = gs;
    L.seq_bpp = bpp;
    L.bd = bd;
    L.blocks = blocks;
    return L;
}

static int check(int N, int nQ, int packed, int bd, int blocks, int gs, int bits46 = 0) {
    const DbLayout L = make(N, nQ, packed, bd, blocks, gs, 1, bits46);
    const size_t cts = (size_t)bd * blocks, total = cts * L.ct_bytes;
    std::vector<unsigned char> used(total, 0);
    for (size_t t = 0; t < cts; t++)
        for (int p = 0; p < 2; p++)
            for (int j = 0; j < nQ; j++)
                for (size_t c = 0; c < (size_t)N; c += (L.bits46 && j > 0) ? 16 : 2) {
                    // granule: a pair of 6- / 8-byte residues, or sixteen 46-bit residues = 92 bytes
                    const size_t es = (L.bits46 && j > 0) ? 46 : (packed && j > 0) ? 6 : 8, o = db_offset(L, N, t, p, j, c);
                    if (o + 2 * es > total) return printf("out of range: t %zu p %d j %d c %zu\n", t, p, j, c), 1;
                    if (o % 4) return printf("granule not on a dword: t %zu p %d j %d c %zu\n", t, p, j, c), 1;
                    for (size_t k = 0; k < 2 * es; k++) {
                        if (used[o + k]) return printf("overlap at byte %zu (t %zu p %d j %d c %zu)\n", o + k, t, p, j, c), 1;
                        used[o + k] = 1;
                    }
                }
    for (size_t k = 0; k < total; k++)
        if (!used[k]) return printf("hole at byte %zu\n", k), 1;
    if (L.bits46) {  // the widest access: lane 63's 16-byte load inside the LAST unit of the allocation must end inside db_alloc_size
        size_t last = 0;
        for (size_t t = 0; t < cts; t++)
            for (int p = 0; p < 2; p++)
                for (int j = 1; j < nQ; j++)
                    for (size_t c = 0; c < (size_t)N; c += 128) last = std::max(last, db_offset(L, N, t, p, j, c));
        if (last + db_lane_load46(63) + 16 <= total) return printf("lane 63 stays inside the unit: the tail would be dead weight\n"), 1;
        if (last + db_lane_load46(63) + 16 > db_alloc_size(L, cts)) return printf("lane 63's load ends past the allocation\n"), 1;
    }
    if (gs) {  // one workgroup's bytes: limb j, tile, group -> [diagonal][block in group][polynomial][128 residues] back to back
        for (int j = 0; j < nQ; j++)
            for (int tile = 0; tile < N / 128; tile++)
                for (int grp = 0; grp < blocks / gs; grp++) {
                    size_t expect = db_offset(L, N, (size_t)grp * gs * bd, 0, j, (size_t)tile * 128);
                    for (int i = 0; i < bd; i++)
                        for (int u = 0; u < gs; u++)
                            for (int p = 0; p < 2; p++) {
                                const size_t o = db_offset(L, N, ((size_t)grp * gs + u) * bd + i, p, j, (size_t)tile * 128);
                                if (o != expect) return printf("run broken: j %d tile %d grp %d i %d u %d p %d\n", j, tile, grp, i, u, p), 1;
                                expect += db_unit_bytes(L, j);
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
    bad |= check(256, 3, 1, 4, 12, 4, 1);  // group-sequential with 46-bit residues in 736-byte units
    bad |= check(512, 4, 1, 8, 16, 8, 1);
    bad |= check(256, 2, 1, 2, 10, 2, 1);
    if (!bad) printf("db layout ok\n");
    return bad;
}

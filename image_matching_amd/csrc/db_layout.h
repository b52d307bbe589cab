// image_matching_amd/csrc/db_layout.h — where a residue of the resident database lies (shared by the kernels and, on the host, by
// tests/csrc/db_layout_check.cpp, which checks that the address map is a bijection onto the allocation).
#pragma once
#include <cstddef>

#include "devmath.h"

// Resident database layouts.  Residues: limb 0 as 8-byte integers, (packed) limbs 1.. as 6-byte integers — or 8 bytes everywhere.
//   ciphertext-major (seq = 0): ciphertext t at t*ct_bytes, polynomial p at + p*poly_bytes, limb j at + db_limb_offset(j), coefficients in order.
//   group-sequential (seq = gs > 0), for databases of many blocks in the hoisted form: the bytes ONE loop-B workgroup reads — a
//     128-coefficient tile of one limb of gs blocks (gs = waves x blocks per wave of the launch) — form ONE sequential run:
//     [limb][tile][group of gs blocks][diagonal][block in group][polynomial][128 residues].  HBM serves that pattern at 7.0 TB/s
//     where the ciphertext-major one (768-byte pieces 4.6 MB apart) gets 6.05 (tools/ubench/stream_rate.hip).
// Both hold ct_bytes per ciphertext; a ciphertext's address is db_offset() in either.
struct DbLayout {
    unsigned long long ct_bytes, poly_bytes;
    int packed;
    int seq, seq_bpp;  // group size gs (0 = ciphertext-major) and the blocks per wave it was chosen with (waves = gs / seq_bpp)
    int bd, blocks;    // seq: ciphertexts per block (the diagonal count), blocks resident
};

HD size_t db_limb_offset(const DbLayout &L, int N, int j) {
    return L.packed ? (j == 0 ? 0 : (size_t)N * 8 + (size_t)(j - 1) * N * 6) : (size_t)j * N * 8;
}
// byte offset of residue c (even: residues travel in pairs) of limb j, polynomial p, ciphertext t
HD size_t db_offset(const DbLayout &L, int N, size_t t, int p, int j, size_t c) {
    const size_t es = (L.packed && j > 0) ? 6 : 8;
    if (!L.seq) return t * L.ct_bytes + (size_t)p * L.poly_bytes + db_limb_offset(L, N, j) + c * es;
    const size_t g = t / L.bd, i = t % L.bd, grp = g / L.seq, u = g % L.seq, tile = c >> 7, cc = c & 127, groups = L.blocks / L.seq;
    return (size_t)L.blocks * L.bd * 2 * db_limb_offset(L, N, j) + ((((tile * groups + grp) * L.bd + i) * L.seq + u) * 2 + p) * 128 * es + cc * es;
}

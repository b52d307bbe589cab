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
//     bits46 (round 4; group-sequential only): the packed limbs' residues — below 2^46, every scaling prime of the default chain is —
//     are stored as 46-BIT integers, 128 of them in a unit of 736 bytes instead of 768: lane l of a loop-B wave finds its two residues
//     at bit 92 l of the unit and fetches them with one 16-byte load from dword (92 l) >> 5 (measured: the stream takes 5.5 % less
//     time, stream_rate.hip).  Sixteen residues = 92 bytes = 23 dwords is the granule everything else (enrolment, export, files) uses.
// Both hold ct_bytes per ciphertext; a ciphertext's address is db_offset() in either.
struct DbLayout {
    unsigned long long ct_bytes, poly_bytes;
    int packed;
    int seq, seq_bpp;  // group size gs (0 = ciphertext-major) and the blocks per wave it was chosen with (waves = gs / seq_bpp)
    int bd, blocks;    // seq: ciphertexts per block (the diagonal count), blocks resident
    int bits46;        // seq && packed: 46-bit residues in 736-byte units (else 48-bit in 768)
};

// bytes of 128 consecutive residues of limb j
HD size_t db_unit_bytes(const DbLayout &L, int j) { return (L.packed && j > 0) ? (L.bits46 ? 736 : 768) : 1024; }
// bytes of limbs 0 .. j-1 of one polynomial
HD size_t db_limb_offset(const DbLayout &L, int N, int j) {
    if (!L.packed) return (size_t)j * N * 8;
    if (j == 0) return 0;
    return L.bits46 ? (size_t)N * 8 + (size_t)(j - 1) * (N / 128) * 736 : (size_t)N * 8 + (size_t)(j - 1) * N * 6;
}
// byte offset of residue c of limb j, polynomial p, ciphertext t.  c even (residues travel in pairs); for the 46-bit limbs of a bits46
// layout c is a multiple of 16 (the 92-byte granule)
HD size_t db_offset(const DbLayout &L, int N, size_t t, int p, int j, size_t c) {
    const size_t es = (L.packed && j > 0) ? 6 : 8;
    if (!L.seq) return t * L.ct_bytes + (size_t)p * L.poly_bytes + db_limb_offset(L, N, j) + c * es;
    const size_t g = t / L.bd, i = t % L.bd, grp = g / L.seq, u = g % L.seq, tile = c >> 7, cc = c & 127, groups = L.blocks / L.seq;
    const size_t unit = ((((tile * groups + grp) * L.bd + i) * L.seq + u) * 2 + p) * db_unit_bytes(L, j);
    const size_t in_unit = (L.bits46 && L.packed && j > 0) ? (cc >> 4) * 92 : cc * es;
    return (size_t)L.blocks * L.bd * 2 * db_limb_offset(L, N, j) + unit + in_unit;
}

// first byte of the 16-byte load lane `lane` of a loop-B wave issues inside a 46-bit unit (its two residues start at bit 92 lane)
HD size_t db_lane_load46(int lane) { return (size_t)((lane * 92) >> 5) * 4; }
// bytes EVERY allocator of a resident database asks for: the layout's bytes plus a tail — lane 63's load ends 4 bytes past its 736-byte
// unit, i.e. past the allocation when the unit is the last one (tests/csrc/db_layout_check.cpp checks the bound)
constexpr size_t DB_ALLOC_TAIL = 64;
HD size_t db_alloc_size(const DbLayout &L, size_t cts) { return cts * L.ct_bytes + DB_ALLOC_TAIL; }

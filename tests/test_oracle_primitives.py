"""CPU oracle: arithmetic primitives against independent known answers (pure-Python big-int arithmetic)."""
import numpy as np
import pytest

import oracle_lib as O


def br(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def test_parameter_shape_matches_reference_context(full_params):
    # src/main.cpp:169-173 -> N = 2^15, depth 11 -> 12 Q limbs; SURVEY §8d: 60 + 11x45 bits, 4x60-bit P, dnum 3
    P = full_params
    assert (P.N, P.slots, P.nQ, P.nP, P.dnum, P.alpha, P.dim) == (32768, 16384, 12, 4, 3, 4, 512)
    bits = [int(q).bit_length() for q in P.moduli]
    assert bits[0] == 60 and all(b in (45, 46) for b in bits[1:12]) and bits[12:] == [60] * 4
    assert len(set(int(q) for q in P.moduli)) == P.nT
    for q, psi in zip(P.moduli, P.roots):
        q, psi = int(q), int(psi)
        assert q % (2 * P.N) == 1
        assert pow(psi, P.N, q) == q - 1  # primitive 2N-th root
    # scaling primes hug 2^45 (FIXEDMANUAL treats them as Delta)
    assert max(abs(int(q) / 2.0 ** 45 - 1) for q in P.moduli[1:12]) < 1e-6


@pytest.mark.parametrize("m", [0, 1, 11, 12, 15])
def test_ntt_matches_direct_evaluation_and_roundtrip(small_params, m):
    P = small_params
    q, psi = int(P.moduli[m]), int(P.roots[m])
    rng = np.random.default_rng(m)
    a = rng.integers(0, q, size=P.N, dtype=np.uint64)
    A = P.ntt_fwd(a, m)
    assert np.array_equal(P.ntt_inv(A, m), a)
    al = [int(v) for v in a]
    for j in [0, 1, 2, 77, P.N - 1]:
        x = pow(psi, 2 * br(j, P.log_n) + 1, q)
        v = 0
        for c in reversed(al):
            v = (v * x + c) % q
        assert v == int(A[j])


def test_negacyclic_known_answer(small_params):
    # X * X^(N-1) = X^N = -1 in Z_q[X]/(X^N+1)
    P = small_params
    for m in (0, 3):
        q = int(P.moduli[m])
        x = np.zeros(P.N, dtype=np.uint64); x[1] = 1
        y = np.zeros(P.N, dtype=np.uint64); y[P.N - 1] = 1
        X, Y = P.ntt_fwd(x, m), P.ntt_fwd(y, m)
        prod = np.array([(int(a) * int(b)) % q for a, b in zip(X, Y)], dtype=np.uint64)
        z = P.ntt_inv(prod, m)
        assert int(z[0]) == q - 1 and not z[1:].any()


def test_ntt_edge_values(small_params):
    P = small_params
    q = int(P.moduli[0])
    for fill in (0, q - 1):
        a = np.full(P.N, fill, dtype=np.uint64)
        assert np.array_equal(P.ntt_inv(P.ntt_fwd(a, 0), 0), a)


def test_automorphism_eval_is_ntt_of_coefficient_automorphism(small_params):
    P = small_params
    rng = np.random.default_rng(5)
    for m in (0, 2):
        q = int(P.moduli[m])
        a = rng.integers(0, q, size=P.N, dtype=np.uint64)
        for rot in (1, 3, 63, 512, -1):
            g = P.galois(rot)
            assert g == pow(5, rot % P.slots, 2 * P.N)
            lhs = P.automorph_eval(P.ntt_fwd(a, m), g)
            rhs = P.ntt_fwd(P.automorph_coeff(a, g, q), m)
            assert np.array_equal(lhs, rhs)


def test_chacha20_known_answer():
    # RFC 7539 section 2.3.2 block function vector, mapped onto this build's (64-bit counter, 64-bit stream) words:
    # counter word 12 = 1, words 13..15 = 0x09000000, 0x4a000000, 0
    key = bytes(range(32))
    out = O.chacha_block(key, stream=0x4A000000, block=1 | (0x09000000 << 32))
    expect = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
              0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert [int(v) for v in out] == expect


def test_samplers_are_deterministic_and_well_distributed():
    q = (1 << 45) + 0x8001
    u = O.sample_uniform(3, 42, q, 1 << 16)
    assert np.array_equal(u, O.sample_uniform(3, 42, q, 1 << 16))
    assert not np.array_equal(u, O.sample_uniform(3, 43, q, 1 << 16))
    assert u.max() < q and abs(u.astype(np.float64).mean() / q - 0.5) < 0.01
    # prefix property: sample i depends only on (seed, stream, i)
    assert np.array_equal(u[:1000], O.sample_uniform(3, 42, q, 1000))
    t = O.sample_ternary(3, 1, 1 << 16)
    assert set(np.unique(t)) == {-1, 0, 1}
    assert all(abs((t == v).mean() - 1 / 3) < 0.01 for v in (-1, 0, 1))
    g = O.sample_gauss(3, 2, 1 << 18)
    assert abs(g.mean()) < 0.05 and abs(g.std() - 3.19) < 0.03 and np.abs(g).max() <= 29
    # exact law at 0: rho(0)/S with S = sum_k exp(-k^2/(2 sigma^2))
    ks = np.arange(-60, 61)
    p0 = 1.0 / np.exp(-ks ** 2 / (2 * 3.19 ** 2)).sum()
    assert abs((g == 0).mean() - p0) < 0.004


def test_encode_is_the_canonical_embedding(small_params):
    """m(zeta^{5^j}) = Delta * z_j for the encoded polynomial (checked by direct complex evaluation)."""
    P = small_params
    rng = np.random.default_rng(9)
    z = rng.uniform(-1, 1, P.slots)
    co = P.encode_coeffs(z).astype(np.float64)
    M = 2 * P.N
    for j in (0, 1, 17, P.slots - 1):
        x = np.exp(2j * np.pi * pow(5, j, M) / M)
        val = np.polyval(co[::-1], x) / P.delta
        assert abs(val.real - z[j]) < 1e-9 and abs(val.imag) < 1e-9
    poly = np.stack([np.mod(P.encode_coeffs(z), int(P.moduli[j])).astype(np.uint64) for j in range(2)])
    assert np.abs(P.decode(poly) - z).max() < 1e-10
    assert np.abs(P.decode(poly[:1]) - z).max() < 1e-10
    # short input is zero padded
    co2 = P.encode_coeffs(z[:10])
    zz = np.zeros(P.slots); zz[:10] = z[:10]
    assert np.array_equal(co2, P.encode_coeffs(zz))


def test_encrypt_decrypt_rotate_multiply(small_params, small_keys):
    P, Or = small_params, O.Oracle(small_params, small_keys)
    rng = np.random.default_rng(11)
    z, w = rng.uniform(-1, 1, P.slots), rng.uniform(-1, 1, P.slots)
    a, b = Or.encrypt(z, 1, 1), Or.encrypt(w, 1, 2)
    assert (a.npoly, a.nl) == (2, P.nQ)
    assert np.abs(Or.decrypt(a) - z).max() < 1e-7
    # same (seed, nonce) -> same ciphertext; different nonce -> different randomness
    assert np.array_equal(a.data(), Or.encrypt(z, 1, 1).data())
    assert not np.array_equal(a.data()[1], Or.encrypt(z, 1, 3).data()[1])
    for r in (1, 5, 63, 64, 512):
        assert np.abs(Or.decrypt(Or.rotate(a, r)) - np.roll(z, -r)).max() < 1e-7
    m = Or.mult(a, b)
    assert (m.npoly, m.nl) == (2, P.nQ - 1)
    assert np.abs(Or.decrypt(m) - z * w).max() < 1e-7
    d = Or.mult_norelin(a, b)
    assert d.npoly == 3 and np.abs(Or.decrypt(d) - z * w).max() < 1e-7  # 3-component decrypt


def test_keyswitch_at_every_level(small_params, small_keys):
    """Relinearise/rescale chain down to one limb: exercises partial digits (nl not a multiple of alpha)."""
    P, Or = small_params, O.Oracle(small_params, small_keys)
    z = np.full(P.slots, 0.9)
    a = Or.encrypt(z, 2, 1)
    cur, val = a, z.copy()
    while cur.nl > 1:
        cur = Or.mult(cur, cur)
        val = val * val
        assert np.abs(Or.decrypt(cur) - val).max() < 1e-5, cur.nl
    assert cur.nl == 1
    r = Or.rotate(cur, 8)
    assert np.abs(Or.decrypt(r) - val).max() < 1e-5

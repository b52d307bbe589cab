"""CPU oracle: the HyDia roles on a reduced ring, against plaintext cosine (the reference's own check,
src/main_accuracy.cpp:354-364, tolerance 1e-4) and the >= 1.0 decision rule (src/receiver/receiver_hers.cpp:30,47)."""
import numpy as np
import pytest

import oracle_lib as O

TOL = 1e-4  # src/main_accuracy.cpp:359-360


def synth_db(rng, n, dim, matches):
    """Distribution of tools/gen_dataset.sh: random rows in [-99,99], matching rows in {1,2,3}, query all ones."""
    db = rng.integers(-99, 100, size=(n, dim)).astype(np.float64)
    for i in matches:
        db[i] = rng.integers(1, 4, size=dim)
    return db


def cosine(db, q):
    nrm = np.linalg.norm(db, axis=1, keepdims=True)
    nrm[nrm == 0] = 1.0
    return (db / nrm) @ (q / np.linalg.norm(q))


def test_enroll_layout_is_the_generalised_diagonal_packing(small_params):
    """enroller_diag.cpp:99-156: slot j*dim + r of ciphertext g*dim + i holds M_{g*per+j}[r][(r+i) mod dim]."""
    P = small_params
    rng = np.random.default_rng(0)
    n, dim, per = 1100, P.dim, P.slots // P.dim
    db = rng.normal(size=(n, dim))
    T = P.L.hyo_enroll_num_cts(P.h, n)
    assert T == -(-(-(-n // dim)) // per) * dim == 2 * dim
    slots = np.zeros(P.slots)
    for t in (0, 1, dim - 1, dim, 2 * dim - 1):
        P.L.hyo_enroll_layout_row(P.h, db.ctypes.data, n, t, slots.ctypes.data)
        g, i = divmod(t, dim)
        for j in (0, 1, per - 1):
            for r in (0, 5, dim - 1):
                v = (g * per + j) * dim + r
                want = db[v, (r + i) % dim] if v < n else 0.0
                assert slots[j * dim + r] == want


def test_normalize_matches_reference_semantics(small_params):
    P = small_params
    x = np.arange(1.0, P.dim + 1)
    y = x.copy()
    P.L.hyo_normalize(y.ctypes.data, P.dim)
    assert np.allclose(y, x / np.linalg.norm(x), rtol=0, atol=1e-15)
    z = np.zeros(P.dim)
    P.L.hyo_normalize(z.ctypes.data, P.dim)  # vector_utils.cpp:45: zero vector passes through
    assert not z.any()


@pytest.mark.parametrize("matvec", ["hoisted", "bsgs", 16])
@pytest.mark.parametrize("n,matches", [(1500, [0, 700, 1499]), (64, [63]), (1, [0]), (1024, []), (1025, [1024])])
def test_hydia_path_small_ring(small_params, small_keys, n, matches, matvec):
    """both forms of the mat-vec: the reference's hoisted rotations and the baby-step / giant-step restatement (pre-rotated diagonals)"""
    P, Or = small_params, O.Oracle(small_params, small_keys)
    rng = np.random.default_rng(n)
    db = synth_db(rng, n, P.dim, matches)
    query = np.ones(P.dim)
    cos = cosine(db, query)
    dbc = Or.enroll(db.copy(), 99, matvec=matvec)
    assert len(dbc) == P.L.hyo_enroll_num_cts(P.h, n)
    q = Or.encrypt_query(query, 5, 1)
    sim = Or.compute_similarity(q, dbc, n)
    G = -(-n // P.slots)
    assert len(sim) == G and sim[0].nl == P.nQ - 1 and sim[0].npoly == 2
    scores = np.concatenate([Or.decrypt(sim[i]) for i in range(G)])
    assert np.abs(scores[:n] - cos).max() < TOL
    assert n == len(scores) or np.abs(scores[n:]).max() < TOL  # padded rows score 0
    idx = Or.index_scenario(q, dbc, n)
    assert idx[0].nl == 1
    assert Or.decrypt_index(idx) == sorted(matches)
    mem = Or.membership_scenario(q, dbc, n)
    assert Or.decrypt_membership(mem) == (len(matches) > 0)


def test_bsgs_layout_is_the_hoisted_layout_rotated_in_the_clear(small_params):
    """slot vector of ciphertext t in the baby-step / giant-step form = the reference layout's (enroller_diag.cpp:99-156) rotated by
    -B (i div B) slots, B = hyo_bsgs_babies (8 at vector_dim 64, 32 at 512): what Rot_{Bg} undoes after the inner sums"""
    P = small_params
    B = P.L.hyo_bsgs_babies(P.h)
    assert B == 8 and O.Params(log_n=12, depth=3, dim=512).L.hyo_bsgs_babies(O.Params(log_n=12, depth=3, dim=512).h) == 32
    rng = np.random.default_rng(2)
    n = 1500
    db = rng.standard_normal((n, P.dim))
    plain, rot = np.zeros(P.slots), np.zeros(P.slots)
    for t in (0, 7, 8, 9, 63, 64 + 17, 64 + 63):
        P.L.hyo_enroll_layout_row(P.h, db.ctypes.data, n, t, plain.ctypes.data)
        for Bx in (B, 2 * B, P.dim):  # the square-root split, a coarser one, and B = dim = the reference layout itself
            P.L.hyo_enroll_layout_row_bsgs(P.h, db.ctypes.data, n, t, rot.ctypes.data, Bx)
            sh = Bx * ((t % P.dim) // Bx)
            assert np.array_equal(rot, np.roll(plain, sh)), (t, Bx)  # slot s takes slot s - sh


def test_zero_vector_row_and_enroll_normalises_in_place(small_params, small_keys):
    P, Or = small_params, O.Oracle(small_params, small_keys)
    rng = np.random.default_rng(1)
    db = synth_db(rng, 100, P.dim, [3])
    db[10] = 0.0
    cos = cosine(db, np.ones(P.dim))
    arr = db.copy()
    dbc = Or.enroll(arr, 1)
    assert np.allclose(np.linalg.norm(np.delete(arr, 10, axis=0), axis=1), 1.0)  # enroller_diag.cpp:32-35 mutates
    assert not arr[10].any()
    sim = Or.compute_similarity(Or.encrypt_query(np.ones(P.dim), 5, 1), dbc, 100)
    s = Or.decrypt(sim[0])
    assert np.abs(s[:100] - cos).max() < TOL and abs(s[10]) < TOL


def test_compare_guard_and_plain_curve(small_params, small_keys):
    P, Or = small_params, O.Oracle(small_params, small_keys)
    x = np.linspace(-1, 1, P.slots)
    ct = Or.encrypt(x, 3, 1)
    P.L.hyo_drop_to(P.h, ct.h, P.nQ - 1)
    out = Or.decrypt(Or.chebyshev_compare(ct, 0.44, 10))
    ref = np.array([P.L.hyo_compare_plain(float(v), 0.44, 59) for v in x])
    assert np.abs(out - ref).max() < TOL
    # decisions: everything below the transition band -> < 1, above -> >= 1
    assert (out[x < 0.43] < 1.0).all() and (out[x > 0.47] >= 1.0).all()
    # openFHE_wrapper.cpp:146-149: depth outside 7..15 -> message, input returned unchanged
    same = Or.chebyshev_compare(ct, 0.44, 6)
    assert same.nl == ct.nl and np.array_equal(same.data(), ct.data())
    # lower depths of the reference's DEPTH_TO_DEGREE table
    for depth, degree in ((7, 5), (8, 13), (9, 27)):
        o = Or.decrypt(Or.chebyshev_compare(ct, 0.44, depth))
        r = np.array([P.L.hyo_compare_plain(float(v), 0.44, degree) for v in x])
        assert np.abs(o - r).max() < TOL


def numpy_compare_plain(x, delta, degree):
    """chebyshevCompare's plain composite (src/openFHE_wrapper.cpp:143-185) with numpy only: Chebyshev interpolation of the step at the
    degree + 1 Chebyshev nodes (what EvalChebyshevFunction derives), then f4 (:158-169), then + 1"""
    k = np.arange(degree + 1)
    nodes = np.cos(np.pi * (k + 0.5) / (degree + 1))
    f = np.where(nodes >= delta, 1.0, -1.0)
    c = np.array([2.0 / (degree + 1) * np.sum(f * np.cos(np.pi * j * (k + 0.5) / (degree + 1))) for j in range(degree + 1)])
    c[0] *= 0.5
    y = np.polynomial.chebyshev.chebval(x, c)
    f4 = [0, 315 / 128, 0, -420 / 128, 0, 378 / 128, 0, -180 / 128, 0, 35 / 128]
    return np.polynomial.polynomial.polyval(y, f4) + 1.0


def test_compare_depths_11_to_15(small_params):
    """The upper half of the reference's DEPTH_TO_DEGREE table (src/openFHE_wrapper.cpp:153-155: depths 11..15 = degrees 119, 247, 495,
    1007, 2031; approach 5 itself uses depth 10).  (i) the oracle's plain composite equals an independent numpy evaluation of the same
    construction; (ii) the oracle's ENCRYPTED evaluation on a 16-level chain (reduced ring) agrees with it within the 1e-4 tolerance."""
    x = np.concatenate([np.linspace(-1, 1, 33), np.linspace(0.40, 0.48, 16)])  # each oracle call re-derives the O(degree^2) coefficients
    L = O.lib()
    degrees = {11: 119, 12: 247, 13: 495, 14: 1007, 15: 2031}
    for depth, degree in degrees.items():
        got = np.array([L.hyo_compare_plain(float(v), 0.44, degree) for v in x])
        assert np.abs(got - numpy_compare_plain(x, 0.44, degree)).max() < 1e-6, depth
    P = O.Params(log_n=11, depth=16, dim=64)
    K = O.Keys(P, 5, rotations=[])
    Or = O.Oracle(P, K)
    xs = np.linspace(-1, 1, P.slots)
    ct = Or.encrypt(xs, 3, 1)
    P.L.hyo_drop_to(P.h, ct.h, P.nQ - 1)
    for depth, degree in degrees.items():
        out_ct = Or.chebyshev_compare(ct, 0.44, depth)
        assert out_ct.nl == P.nQ - 1 - depth
        out = Or.decrypt(out_ct)
        ref = numpy_compare_plain(xs, 0.44, degree)  # pinned to the oracle's own composite in (i)
        assert np.abs(out - ref).max() < TOL, depth
        band = 4.0 / degree + 0.01  # the transition narrows with the degree
        assert (out[xs < 0.44 - band] < 1.0).all() and (out[xs > 0.44 + band] >= 1.0).all(), depth


@pytest.mark.parametrize("n,matches", [(1300, [0, 1299]), (5, [2])])
def test_hers_path_small_ring(small_params, small_keys, n, matches):
    """Approach 4 (HERS) on the oracle: column packing, one query ciphertext per dimension, relin + rescale per
    product — same scores and decisions as plaintext cosine."""
    P, Or = small_params, O.Oracle(small_params, small_keys)
    rng = np.random.default_rng(n)
    db = synth_db(rng, n, P.dim, matches)
    query = np.ones(P.dim)
    cos = cosine(db, query)
    dbc = Or.hers_enroll(db.copy(), 4)
    G = -(-n // P.slots)
    assert len(dbc) == G * P.dim
    q = Or.hers_encrypt_query(query, 6)
    assert len(q) == P.dim
    sim = Or.hers_compute_similarity(q, dbc, n)
    scores = np.concatenate([Or.decrypt(sim[i]) for i in range(G)])
    assert np.abs(scores[:n] - cos).max() < TOL and (n == len(scores) or np.abs(scores[n:]).max() < TOL)
    assert Or.decrypt_index(Or.hers_index_scenario(q, dbc, n)) == sorted(matches)
    assert Or.decrypt_membership(Or.hers_membership_scenario(q, dbc, n)) is True
    # layout: slot k of ciphertext (m, j) is coordinate j of vector m*slots + k (enroller_hers.cpp:108-113)
    slots = np.zeros(P.slots)
    dbn = db / np.linalg.norm(db, axis=1, keepdims=True)
    P.L.hyo_hers_layout_row(P.h, dbn.ctypes.data, n, (G - 1) * P.dim + 3, slots.ctypes.data)
    k = min(n - (G - 1) * P.slots, P.slots) - 1
    assert slots[k] == dbn[(G - 1) * P.slots + k, 3] and (k + 1 == P.slots or slots[k + 1] == 0.0)


def test_custom_prime_chain_context():
    """hyo_params_create_custom (the OpenFHE-adapter path): the whole path on a caller-supplied prime chain and
    caller-supplied 2N-th roots; bad chains are refused."""
    moduli, roots = O.alt_prime_chain(11)
    P = O.Params(log_n=11, depth=11, dim=64, moduli=moduli, roots=roots, n_p=4)
    assert np.array_equal(P.moduli, moduli) and np.array_equal(P.roots, roots) and (P.nQ, P.nP) == (12, 4)
    dflt = O.Params(log_n=11, depth=11, dim=64)
    assert not set(int(v) for v in dflt.moduli) & set(int(v) for v in moduli)
    K = O.Keys(P, 3)
    Or = O.Oracle(P, K)
    rng = np.random.default_rng(5)
    n = 700
    db = rng.integers(-99, 100, size=(n, P.dim)).astype(np.float64)
    db[13] = rng.integers(1, 4, size=P.dim)
    query = np.ones(P.dim)
    cos = (db / np.linalg.norm(db, axis=1, keepdims=True)) @ (query / np.linalg.norm(query))
    dbc = Or.enroll(db.copy(), 4)
    q = Or.encrypt_query(query, 6, 1)
    sim = Or.compute_similarity(q, dbc, n)
    scores = Or.decrypt(sim[0])
    assert np.abs(scores[:n] - cos).max() < 1e-4
    idx = Or.index_scenario(q, dbc, n)
    found = Or.decrypt_index(idx)
    assert 13 in found and all(cos[i] > 0.43 for i in found)
    bad = moduli.copy()
    bad[3] += 2
    with pytest.raises(ValueError):
        O.Params(log_n=11, depth=11, dim=64, moduli=bad, n_p=4)


def test_file_handoff_variant_equals_in_memory_path(small_params, small_keys, tmp_path):
    """The reference's sender re-reads serial/db_diagonal/index<t>.bin inside loop B (sender_diag.cpp:85-94); the oracle's
    file-backed indexScenario (bench.py's disk-reread CPU baseline) gives the same ciphertexts as the in-memory one."""
    P, K = small_params, small_keys
    Or = O.Oracle(P, K)
    rng = np.random.default_rng(3)
    n = 2100
    db = synth_db(rng, n, P.dim, [7, 2000])
    dbc = Or.enroll(db, 9, matvec="hoisted")  # the reference's own form: its files feed its hoisted sender
    Or.write_db_files(dbc, tmp_path)
    assert len(list(tmp_path.iterdir())) == len(dbc)
    q = Or.encrypt_query(np.ones(P.dim), 5, 1)
    a, b = Or.index_scenario(q, dbc, n), Or.index_scenario_files(q, tmp_path, n)
    assert len(a) == len(b) == 3
    for g in range(3):
        assert np.array_equal(a[g].data(), b[g].data())
    assert Or.decrypt_index(b) == [7, 2000]

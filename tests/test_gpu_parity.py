"""GPU parity (run with -m gpu on an MI355X): every HIP path is called through the C-ABI (include/hydia.h) and
compared BIT-EXACTLY with the CPU oracle on the same inputs — all of it is integer arithmetic, so the bar is
equality, not a tolerance.  Decrypted scores are additionally checked against plaintext cosine to the reference's
own 1e-4 (src/main_accuracy.cpp:359-360).  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def im():
    import image_matching_amd as im
    return im


def make_ctx(im, P):
    prm = im.default_params(log_n=P.log_n, mult_depth=P.nQ - 1, vector_dim=P.dim, dnum=P.dnum)
    cc = im.Context(prm, 0)
    assert np.array_equal(cc.moduli, P.moduli) and np.array_equal(cc.roots, P.roots)
    return cc


def load_keys(cc, K, rotations):
    cc.import_eval_key(0, K.relin())
    for r in rotations:
        cc.import_eval_key(r, K.rot_key(r))
    cc.import_public_key(K.pk())
    cc.import_secret_key(K.s_ntt())


def load_db(cc, dbc, n):
    cc.db_alloc(n)
    assert cc.db_num_cts(n) == len(dbc)
    for t in range(len(dbc)):
        cc.db_import_ct(t, dbc[t].data())
    if getattr(dbc, "bsgs", False):  # the oracle enrolled pre-rotated diagonals (baby-step / giant-step split)
        cc.db_set_babies(dbc.babies)


@pytest.fixture(scope="module")
def small(im):
    P = O.Params(log_n=11, depth=11, dim=64)
    K = O.Keys(P, 7)
    cc = make_ctx(im, P)
    load_keys(cc, K, K.rotations)
    yield P, K, O.Oracle(P, K), cc
    cc.close()


@pytest.mark.parametrize("log_n,env", [(11, None), (13, None), (15, None), (15, "HYDIA_P2_WG_SYNC")])
def test_ntt_bit_exact(im, log_n, env, monkeypatch):
    """(15, HYDIA_P2_WG_SYNC): the plain N = 2^15 transforms through round 4's workgroup-synchronous pass 2 (chunk-wide phase C behind
    s_barrier) — the parity variant of round 5's wave-synchronous default"""
    P = O.Params(log_n=log_n, depth=11, dim=64)
    if env:
        monkeypatch.setenv(env, "1")
    cc = make_ctx(im, P)
    if env:
        monkeypatch.delenv(env)
    rng = np.random.default_rng(log_n)
    for m in (0, 1, 6, 11, 12, 15):
        q = int(P.moduli[m])
        a = rng.integers(0, q, size=(3, P.N), dtype=np.uint64)
        a[1] = q - 1
        a[2, 1:] = 0
        want = np.stack([P.ntt_fwd(row, m) for row in a])
        got = cc.ntt(a, m)
        assert np.array_equal(got, want), (log_n, m)
        assert np.array_equal(cc.ntt(got, m, inverse=True), a)
        want_inv = np.stack([P.ntt_inv(row, m) for row in a])
        assert np.array_equal(cc.ntt(a, m, inverse=True), want_inv)
    cc.close()


def test_evaluator_primitives_bit_exact(small):
    P, K, Or, cc = small
    rng = np.random.default_rng(1)
    z, w = rng.uniform(-1, 1, P.slots), rng.uniform(-1, 1, P.slots)
    a, b = Or.encrypt(z, 1, 1), Or.encrypt(w, 1, 2)
    ga, gb = cc.import_ct(a.data(), a.scale), cc.import_ct(b.data(), b.scale)
    # EvalMultNoRelin / Relinearize / Rescale
    d = Or.mult_norelin(a, b)
    gd = cc.eval_mult_no_relin(ga, gb)
    assert np.array_equal(gd.export()[0], d.data())
    Or.relin(d); cc.relinearize(gd)
    assert np.array_equal(gd.export()[0], d.data())
    Or.rescale(d); cc.rescale(gd)
    assert np.array_equal(gd.export()[0], d.data()) and gd.shape()[2] == P.nQ - 1
    assert abs(gd.shape()[3] / d.scale - 1) < 1e-15
    # rotations (full key switch) incl. partial digits at lower levels
    for r in (1, 5, 63, 64, 512):
        assert np.array_equal(cc.eval_rotate(ga, r).export()[0], Or.rotate(a, r).data()), r
    cur, gcur = a, ga
    while cur.nl > 1:
        cur, gcur = Or.mult(cur, cur), cc.eval_mult(gcur, gcur)
        assert np.array_equal(gcur.export()[0], cur.data()), cur.nl
    assert np.array_equal(cc.eval_rotate(gcur, 8).export()[0], Or.rotate(cur, 8).data())
    # a batch of two ciphertexts behaves like two single ones
    both = cc.import_ct(np.stack([a.data(), b.data()]), a.scale)
    sq = cc.eval_mult(both, both).export()
    assert np.array_equal(sq[0], Or.mult(a, a).data()) and np.array_equal(sq[1], Or.mult(b, b).data())


# the last two cases are the auto rule's MIDDLE tiers at dim 64 (5 blocks -> 32 babies, 13 -> 64 = all hoisted, group-sequential),
# the analogues of the 128- / 256-baby splits BASELINE configs 4 and 5 get per GPU (full ring: tests/test_gpu_full_ring.py)
@pytest.mark.parametrize("n,matches", [(1500, [0, 700, 1499]), (64, [63]), (1, [0]), (1024, []), (1025, [1024]),
                                       (5 * 1024 - 3, [0, 2600, 5 * 1024 - 4]), (13 * 1024 - 7, [9, 13 * 1024 - 8])])
def test_hydia_sender_bit_exact_small_ring(im, small, n, matches):
    P, K, Or, cc = small
    if n > 4096:
        assert O.auto_babies(P.dim, -(-n // P.slots)) == cc.auto_babies(-(-n // P.slots)) == (32 if n < 8192 else 64)
    rng = np.random.default_rng(n)
    db = rng.integers(-99, 100, size=(n, P.dim)).astype(np.float64)
    for i in matches:
        db[i] = rng.integers(1, 4, size=P.dim)
    query = np.ones(P.dim)
    nrm = np.linalg.norm(db, axis=1, keepdims=True)
    cos = (db / nrm) @ (query / np.linalg.norm(query))
    dbc = Or.enroll(db.copy(), 99)
    load_db(cc, dbc, n)
    q = Or.encrypt_query(query, 5, 1)
    gq = cc.import_ct(q.data(), q.scale)
    sender = im.DiagonalSender(cc, n)
    # loop A
    rot, grot = Or.rotate_query(q), sender.rotateQuery(gq).export()
    for i in range(P.dim):
        assert np.array_equal(grot[i], rot[i].data()), i
    # computeSimilarity
    sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq)
    G = -(-n // P.slots)
    assert gsim.shape()[:3] == (G, 2, P.nQ - 1)
    gs = gsim.export()
    for g in range(G):
        assert np.array_equal(gs[g], sim[g].data())
    scores = np.concatenate([Or.decrypt(sim[g]) for g in range(G)])
    assert np.abs(scores[:n] - cos).max() < TOL
    # indexScenario / membershipScenario
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
    gi = gidx.export()
    for g in range(G):
        assert np.array_equal(gi[g], idx[g].data())
    if n <= 4096:  # (64-dimensional random rows cross the 0.44 threshold by chance in larger databases: the planted ones must be among the hits)
        assert Or.decrypt_index(idx) == sorted(matches)
    assert set(matches) <= set(Or.decrypt_index(idx))
    mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
    assert np.array_equal(gmem.export()[0], mem.data())
    assert Or.decrypt_membership(mem) == (len(matches) > 0)


def test_comparator_depths_and_guard(small):
    P, K, Or, cc = small
    x = np.linspace(-1, 1, P.slots)
    ct = Or.encrypt(x, 3, 1)
    P.L.hyo_drop_to(P.h, ct.h, P.nQ - 1)
    g = cc.import_ct(ct.data(), ct.scale)
    for depth in (7, 8, 9, 10):
        assert np.array_equal(cc.chebyshev_compare(g, 0.44, depth).export()[0], Or.chebyshev_compare(ct, 0.44, depth).data())
    same = cc.chebyshev_compare(g, 0.44, 6)  # openFHE_wrapper.cpp:146-149
    assert np.array_equal(same.export()[0], ct.data())


def test_auto_rule_is_mirrored_by_the_oracle_helper(im):
    """tests/oracle_lib.auto_babies (what Or.enroll(matvec=None) uses) == the product's rule (hydia_auto_babies) for both rings, so a
    test that lets both sides choose compares like with like"""
    for prm, dim in ((im.default_params(), 512), (im.default_params(log_n=11, vector_dim=64), 64)):
        cc = im.Context(prm, 0)
        for blocks in list(range(1, 70)) + [100, 1000]:
            assert cc.auto_babies(blocks) == O.auto_babies(dim, blocks), (dim, blocks)
        cc.set_matvec("hoisted")
        assert cc.auto_babies(1) == dim and cc.get_matvec() == "hoisted"
        cc.set_matvec(dim // 4)
        assert cc.auto_babies(50) == dim // 4
        with pytest.raises(im.HydiaError):
            cc.set_matvec(48)  # not a power of two dividing vector_dim
        cc.close()


def test_comparator_depths_11_to_15(im):
    """The upper half of the reference's DEPTH_TO_DEGREE table (src/openFHE_wrapper.cpp:153-155: degrees 119 .. 2031) on a 16-level
    chain at the reduced ring: GPU == oracle bit for bit at every depth, and the decrypted output within 1e-4 of the plain composite
    (approach 5 itself runs depth 10; these are the rest of chebyshevCompare's contract)."""
    from test_oracle_path import numpy_compare_plain
    P = O.Params(log_n=11, depth=16, dim=64)
    K = O.Keys(P, 5, rotations=[])
    Or = O.Oracle(P, K)
    cc = make_ctx(im, P)
    load_keys(cc, K, [])
    x = np.linspace(-1, 1, P.slots)
    ct = Or.encrypt(x, 3, 1)
    P.L.hyo_drop_to(P.h, ct.h, P.nQ - 1)
    g = cc.import_ct(ct.data(), ct.scale)
    for depth, degree in ((11, 119), (12, 247), (13, 495), (14, 1007), (15, 2031)):
        want = Or.chebyshev_compare(ct, 0.44, depth)
        got = cc.chebyshev_compare(g, 0.44, depth)
        assert got.shape()[:3] == (1, 2, P.nQ - 1 - depth)
        assert np.array_equal(got.export()[0], want.data()), depth
        assert np.abs(Or.decrypt(want) - numpy_compare_plain(x, 0.44, degree)).max() < TOL, depth
    assert np.array_equal(cc.chebyshev_compare(g, 0.44, 16).export()[0], ct.data())  # :146-149: outside 7..15 -> unchanged
    cc.close()


def test_error_behaviour(im, small):
    P, K, Or, cc = small
    fresh = make_ctx(im, P)
    q = Or.encrypt_query(np.ones(P.dim), 5, 1)
    gq = fresh.import_ct(q.data(), q.scale)
    with pytest.raises(im.HydiaError) as e:  # no rotation keys / no database
        im.DiagonalSender(fresh, 10).computeSimilarity(gq)
    assert e.value.code == -2
    with pytest.raises(im.HydiaError):
        fresh.import_ct(np.zeros((1, 5, 2, P.N), dtype=np.uint64), 1.0)
    # empty database (the reference prints an error and returns, enroller_diag.cpp:17-28): refused, nothing becomes resident
    with pytest.raises(im.HydiaError) as e:
        im.DiagonalEnroller(cc, 0).serializeDB(np.zeros((0, P.dim)), seed=1)
    assert e.value.code == -1
    # a database beyond the HBM of the device: a clean device error, and the context stays usable
    with pytest.raises(im.HydiaError) as e:
        fresh.db_alloc(1 << 29)
    assert e.value.code == -3
    fresh.keygen(7)
    db = np.ones((3, P.dim))
    im.DiagonalEnroller(fresh, 3).serializeDB(db, seed=2)
    fq = im.DiagonalReceiver(fresh, 3).encryptQuery(np.ones(P.dim), seed=3)
    assert im.DiagonalReceiver(fresh, 3).decryptIndex(im.DiagonalSender(fresh, 3).indexScenario(fq)) == [0, 1, 2]
    # a query that is not at level 0 is refused (sender_diag.cpp multiplies fresh ciphertexts only)
    low = fresh.import_ct(fq.export()[:, :, :5], fq.shape()[3])
    with pytest.raises(im.HydiaError):
        im.DiagonalSender(fresh, 3).computeSimilarity(low)
    fresh.close()


def test_reference_dataset_2_10_full_ring(im):
    """BASELINE config 1/2: ./ImageMatching ../test/2_10.dat 5 on N = 2^15 — GPU sender vs oracle, bit exact, and
    the reference's expected answers: membership true, index [0], scores within 1e-4 of plaintext cosine."""
    P = O.Params()
    K = O.Keys(P, 20250725)
    Or = O.Oracle(P, K)
    cc = make_ctx(im, P)
    load_keys(cc, K, K.rotations)
    g = np.load(os.path.join(GOLDEN, "dataset_2_10.npz"))
    n, query, db = int(g["n"]), g["query"].astype(np.float64), g["db"].astype(np.float64)
    dbc = Or.enroll(db, 99)
    load_db(cc, dbc, n)
    q = Or.encrypt_query(query, 5, 1)
    gq = cc.import_ct(q.data(), q.scale)
    sender = im.DiagonalSender(cc, n)
    sim = Or.compute_similarity(q, dbc, n)
    gsim = sender.computeSimilarity(gq)
    assert np.array_equal(gsim.export()[0], sim[0].data())
    scores = Or.decrypt(sim[0])
    assert np.abs(scores[:n] - g["cosine"]).max() < TOL
    cmp_ct = Or.chebyshev_compare(sim[0], 0.44, 10)
    gidx = sender.indexScenario(gq)
    assert np.array_equal(gidx.export()[0], cmp_ct.data())
    vals = Or.decrypt(cmp_ct)
    assert [int(i) for i in np.nonzero(vals >= 1.0)[0]] == [0]
    # decrypt the GPU membership ciphertext with the oracle's secret key
    gmem = sender.membershipScenario(gq)
    gm = O.Ct(P, P.L.hyo_ct_alloc(P.h, 2, 1, gmem.shape()[3]))
    gm.data()[:] = gmem.export()[0]
    assert Or.decrypt_membership(gm) is True
    cc.close()


@pytest.mark.parametrize("one_pass", [False, True])
def test_ntt15_pair_path_and_mixed_limbs(im, monkeypatch, one_pass):
    """Even polynomial counts take the two-polynomials-per-workgroup pass; both arithmetic back ends (60-bit integer,
    45-bit FP64) must agree with the oracle bit for bit, forward and inverse — through the two-pass kernels and through the
    one-pass kernel (k_ntt15_1p: 1024-thread workgroup, 32 coefficients per lane, LDS transposes only)."""
    if one_pass:
        monkeypatch.setenv("HYDIA_NTT_1PASS", "1")
        monkeypatch.setenv("HYDIA_NTT_1PASS_MIN", "1")
    P = O.Params()
    cc = make_ctx(im, P)
    rng = np.random.default_rng(15)
    for m in (0, 5, 11, 13):
        q = int(P.moduli[m])
        a = rng.integers(0, q, size=(4, P.N), dtype=np.uint64)
        a[3] = q - 1
        f = cc.ntt(a, m)
        assert np.array_equal(f, np.stack([P.ntt_fwd(r, m) for r in a])), m
        assert np.array_equal(cc.ntt(f, m, inverse=True), a), m
    cc.close()


def test_custom_prime_chain_context_bit_exact(im):
    """hydia_ctx_create_custom: a context on a caller-supplied prime chain and 2N-th roots (what an OpenFHE
    adapter passes in) runs the whole path bit-exact against the oracle configured with the same chain; malformed chains are
    refused with HYDIA_ERR_ARG-class errors instead of computing garbage."""
    moduli, roots = O.alt_prime_chain(11)
    P = O.Params(log_n=11, depth=11, dim=64, moduli=moduli, roots=roots, n_p=4)
    K = O.Keys(P, 3)
    Or = O.Oracle(P, K)
    cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0, moduli=moduli, roots=roots, n_p=4)
    assert np.array_equal(cc.moduli, moduli) and np.array_equal(cc.roots, roots)
    cc.keygen(3)
    assert np.array_equal(cc.export_eval_key(0), K.relin())
    rng = np.random.default_rng(5)
    n = 1500
    db = rng.integers(-99, 100, size=(n, P.dim)).astype(np.float64)
    db[13] = rng.integers(1, 4, size=P.dim)
    query = np.ones(P.dim)
    a, b = db.copy(), db.copy()
    dbc = Or.enroll(a, 4)
    im.DiagonalEnroller(cc, n).serializeDB(b, seed=4)
    q = Or.encrypt_query(query, 6, 1)
    gq = im.DiagonalReceiver(cc, n).encryptQuery(query, seed=6, nonce=1)
    assert np.array_equal(gq.export()[0], q.data())
    sender = im.DiagonalSender(cc, n)
    sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq).export()
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq).export()
    for g in range(len(sim)):
        assert np.array_equal(gsim[g], sim[g].data()) and np.array_equal(gidx[g], idx[g].data())
    mem = Or.membership_scenario(q, dbc, n)
    assert np.array_equal(sender.membershipScenario(gq).export()[0], mem.data())
    cc.close()
    bad = moduli.copy()
    bad[3] += 2
    with pytest.raises(im.HydiaError):
        im.Context(im.default_params(log_n=11, vector_dim=64), 0, moduli=bad, n_p=4)
    with pytest.raises(im.HydiaError):
        im.Context(im.default_params(log_n=11, vector_dim=64), 0, moduli=moduli, roots=moduli, n_p=4)


def test_group_sequential_custom_chain_with_46bit_prime_small_ring(im):
    """The adapter path's corner: a caller-supplied chain in which ONE scaling prime is at least 2^46 cannot take the
    46-bit residue units, so a group-sequential database (more than 8 blocks, hoisted) falls back to 48-bit units on that context
    (context.cpp db_relayout, capi.cpp hydia_db_residue_bits).  Enrolled ciphertexts, computeSimilarity and indexScenario of a
    10-block database equal the oracle's on the same chain, bit for bit."""
    from sympy import isprime
    moduli, roots = O.alt_prime_chain(11)
    M = 2 << 11
    c = (1 << 46) + (1 << 40)
    c += 1 - c % M
    while not isprime(c):
        c += M
    assert (1 << 46) <= c < (1 << 47)
    x = 2
    while pow(pow(x, (c - 1) // M, c), M // 2, c) != c - 1:
        x += 1
    moduli, roots = moduli.copy(), roots.copy()
    moduli[5], roots[5] = c, pow(x, (c - 1) // M, c)
    P = O.Params(log_n=11, depth=11, dim=64, moduli=moduli, roots=roots, n_p=4)
    K = O.Keys(P, 3)
    Or = O.Oracle(P, K)
    cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0, moduli=moduli, roots=roots, n_p=4)
    assert np.array_equal(cc.moduli, moduli)
    cc.keygen(3)
    blocks = 10
    n = blocks * P.slots - 3
    rng = np.random.default_rng(46)
    db = rng.integers(-99, 100, size=(n, P.dim)).astype(np.float64)
    db[n // 2] = rng.integers(1, 4, size=P.dim)
    query = np.ones(P.dim)
    dbc = Or.enroll(db.copy(), 4, matvec="hoisted")
    cc.set_matvec("hoisted")
    im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=4)
    assert cc.db_kind() == 5 and cc.db_group() == 2 and cc.db_residue_bits() == 48  # group-sequential, 48-bit fallback
    for t in (0, P.dim + 5, len(dbc) - 1):
        assert np.array_equal(cc.db_export_ct(t), dbc[t].data()), t
    q = Or.encrypt_query(query, 6, 1)
    gq = im.DiagonalReceiver(cc, n).encryptQuery(query, seed=6, nonce=1)
    assert np.array_equal(gq.export()[0], q.data())
    sender = im.DiagonalSender(cc, n)
    sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq).export()
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq).export()
    assert len(sim) == blocks
    for g in range(blocks):
        assert np.array_equal(gsim[g], sim[g].data()) and np.array_equal(gidx[g], idx[g].data()), g
    cc.close()


def test_fast_paths_equal_plain_pipeline_full_ring(im, monkeypatch):
    """Every fast-path decision at N = 2^15 is bit-neutral: the default engine (FP64 NTT butterflies on the 45-bit limbs, 48-bit
    packed database and rotation keys, merged ModDown+Rescale, NTT pass 2 fused with the inner product, two comparator lanes) and the plain one
    (integer butterflies everywhere, 8-byte database and keys, separate relinearise / rescale, unfused inner product, one lane) give
    identical residues for a 3-block database (batched X = 3 evaluator ops, uneven lane split) — index, membership and scores."""
    variants = [
        {},
        {"HYDIA_NTT_INT": "1", "HYDIA_DB_UNPACKED": "1", "HYDIA_KEYS_UNPACKED": "1", "HYDIA_NO_MERGE_RESCALE": "1", "HYDIA_NO_FUSE_IP": "1",
         "HYDIA_LANES": "1", "HYDIA_NO_FORK": "1", "HYDIA_NO_FUSE_LOOPA": "1", "HYDIA_NO_COLFUSE": "1"},
        {"HYDIA_NO_FUSE_IP": "1", "HYDIA_LANES": "3"},
        {"HYDIA_NTT_1PASS": "1", "HYDIA_NTT_1PASS_MIN": "1"},  # the one-pass kernel (one HBM round trip) for every FP64 limb transform
        {"HYDIA_NTT_1PASS": "1"},  # ... only for launches of at least 1024 limb-polynomials
        # round-2 fusions switched off one group at a time: per-digit ModUp launches, loop A's special-prime inner product as its own
        # kernel, the relinearisation's special-prime rows through the accumulator, unsliced conversion targets, paired small transforms
        {"HYDIA_MODUP_PER_DIGIT": "1", "HYDIA_LOOPA_SEPARATE_IP": "1", "HYDIA_RELIN_SEPARATE_INTT": "1", "HYDIA_LOOPA_INT_IP": "1"},
        {"HYDIA_NTT_NO_PM": "1", "HYDIA_RELIN_TWO_IP_LAUNCHES": "1"},
        # round 3: pass 1' / base conversion / pass 1 as three kernels instead of the column-fused one (default arithmetics)
        {"HYDIA_NO_COLFUSE": "1"},
        # the remaining launch-shape switches: loop B one block per wave / two waves per workgroup, interleaving group 1 in the merged
        # inner product; the ring-size-generic transform kernels wherever a plain transform runs (unfused pipeline)
        {"HYDIA_TENSOR_BPP": "1", "HYDIA_TENSOR_NW": "2", "HYDIA_IP_GROUP": "1"},
        # round 4: the dropped limb's inverse pass 2 as its own launch (default: inside the merged inner-product kernel's tail)
        {"HYDIA_NO_DROP_IN_IP": "1", "HYDIA_INT_EPILOGUE": "1"},
        # ... every degree-2 ciphertext materialised by k_tensor (default: d0, d1, d2 formed inside the consumers of the merged pipeline);
        # then that with the integer epilogue, so the product operands go through both epilogue arithmetics
        {"HYDIA_NO_PROD_FUSE": "1"},
        {"HYDIA_INT_EPILOGUE": "1"},
        {"HYDIA_NO_CSUB_FUSE": "1"},  # ... only the Chebyshev steps with a subtrahend (T3, T5, T7) keep k_tensor<true>
        # ... relinearise-only / rotation key switches (EvalSum, the giant steps of the split mat-vec) through modup_digits + ks_apply
        {"HYDIA_NO_KS_FUSE": "1", "HYDIA_NO_RESCALE_CF": "1"},  # ... and Rescale's spread + first pass as k_ntt15_p1<false, 2>  # ... and the merged epilogue in integers on every limb (default: FP64 below 2^47)
        {"HYDIA_NTT_GENERIC": "1", "HYDIA_NO_COLFUSE": "1", "HYDIA_NO_FUSE_IP": "1", "HYDIA_NO_FUSE_LOOPA": "1", "HYDIA_NO_MERGE_RESCALE": "1"},
        # the unfused pipeline on the default arithmetics (FP64 + lazy pseudo-Mersenne butterflies through the plain epilogues)
        {"HYDIA_NO_MERGE_RESCALE": "1", "HYDIA_NO_FUSE_IP": "1", "HYDIA_NO_FUSE_LOOPA": "1", "HYDIA_KEYS_UNPACKED": "1"},  # Harvey [0, 4q) butterflies for the 60-bit primes; two inner-product launches
        # round 5's changes back in round 4's forms: the wide column-fused conversions, per-lane twiddle loads in pass 2, loop A's last pass
        # limbs-fastest, the plain transforms through the workgroup-synchronous pass 2
        {"HYDIA_COLFUSE_WIDE": "1", "HYDIA_NO_TW_LDS": "1", "HYDIA_LOOPA_LIMB_FASTEST": "1", "HYDIA_P2_WG_SYNC": "1"},
        {"HYDIA_NO_TW_LDS": "1"},
    ]
    n = 40000
    rng = np.random.default_rng(77)
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    for i in (5, 20000, n - 1):
        db[i] = rng.integers(1, 4, size=512)
    results = []
    for env in variants:
        for k in ("HYDIA_NTT_INT", "HYDIA_DB_UNPACKED", "HYDIA_KEYS_UNPACKED", "HYDIA_NO_MERGE_RESCALE", "HYDIA_NO_FUSE_IP", "HYDIA_LANES",
                  "HYDIA_NTT_1PASS", "HYDIA_NTT_1PASS_MIN", "HYDIA_NO_FORK", "HYDIA_NO_FUSE_LOOPA", "HYDIA_MODUP_PER_DIGIT",
                  "HYDIA_LOOPA_SEPARATE_IP", "HYDIA_RELIN_SEPARATE_INTT", "HYDIA_LOOPA_INT_IP", "HYDIA_NTT_NO_PM", "HYDIA_RELIN_TWO_IP_LAUNCHES",
                  "HYDIA_NO_COLFUSE", "HYDIA_TENSOR_BPP", "HYDIA_TENSOR_NW", "HYDIA_IP_GROUP", "HYDIA_NTT_GENERIC", "HYDIA_NO_DROP_IN_IP", "HYDIA_INT_EPILOGUE", "HYDIA_NO_PROD_FUSE", "HYDIA_NO_KS_FUSE", "HYDIA_NO_RESCALE_CF", "HYDIA_NO_CSUB_FUSE",
                  "HYDIA_COLFUSE_WIDE", "HYDIA_NO_TW_LDS", "HYDIA_LOOPA_LIMB_FASTEST", "HYDIA_P2_WG_SYNC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cc = im.Context()
        cc.keygen(31)
        im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=8)
        sender = im.DiagonalSender(cc, n)
        q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(512), seed=2, nonce=9)
        results.append((sender.computeSimilarity(q).export(), sender.indexScenario(q).export(), sender.membershipScenario(q).export()))
        if not env:
            assert im.DiagonalReceiver(cc, n).decryptIndex(sender.indexScenario(q)) == [5, 20000, n - 1]
        cc.close()
    for r in results[1:]:
        for a, b in zip(results[0], r):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("blocks,group,env", [(10, 2, {}), (16, 8, {}), (12, 4, {"HYDIA_TENSOR_BPP": "1"}), (16, 8, {"HYDIA_DB_48BIT": "1"})])
def test_group_sequential_database_bit_exact_small_ring(im, small, blocks, group, env, tmp_path, monkeypatch):
    """A hoisted database of more than 8 blocks lies group-sequentially in HBM (loop B's 24-bit-halves kernel reads
    it).  Same ciphertexts in, same ciphertexts out as the oracle — through the GPU enroller and through ciphertext-by-ciphertext
    import; export, save / load (the file is ciphertext-major whatever the resident layout) and a ciphertext-major context agree."""
    P, K, Or, cc = small
    own = None
    if env:  # another loop-B tiling = another group shape: the layout is fixed by the context that allocates the database
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        cc = own = make_ctx(im, P)
        for k in env:
            monkeypatch.delenv(k)
        load_keys(cc, K, K.rotations)
    n = blocks * P.slots - 3
    rng = np.random.default_rng(blocks)
    db = rng.integers(-99, 100, size=(n, P.dim)).astype(np.float64)
    for i in (0, n // 2, n - 1):
        db[i] = rng.integers(1, 4, size=P.dim)
    query = np.ones(P.dim)
    dbc = Or.enroll(db.copy(), 41, matvec="hoisted")
    q = Or.encrypt_query(query, 5, 1)
    sim = Or.compute_similarity(q, dbc, n)
    gq = cc.import_ct(q.data(), q.scale)
    cc.set_matvec("hoisted")
    try:
        im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=41)
        assert cc.db_group() == group and cc.db_kind() == 5
        assert cc.db_residue_bits() == (48 if "HYDIA_DB_48BIT" in env else 46)  # round 4: 46-bit residues in 736-byte units
        for t in (0, 1, P.dim - 1, P.dim, len(dbc) // 2 + 3, len(dbc) - 1):
            assert np.array_equal(cc.db_export_ct(t), dbc[t].data()), t
        sender = im.DiagonalSender(cc, n)
        gs = sender.computeSimilarity(gq).export()
        for g in range(blocks):
            assert np.array_equal(gs[g], sim[g].data()), g
        want_idx = sender.indexScenario(gq).export()
        # (64-dimensional random rows cross the 0.44 threshold by chance now and then: the planted matches must be among the hits)
        assert {0, n // 2, n - 1} <= set(im.DiagonalReceiver(cc, n).decryptIndex(sender.indexScenario(gq)))
        # declaring another form re-orders the resident database (here: to blocks of 8 and back); the ciphertexts stay what they were
        cc.db_set_babies(8)
        assert cc.db_kind() == 6 and cc.db_group() > 0
        assert np.array_equal(cc.db_export_ct(P.dim + 1), dbc[P.dim + 1].data())
        cc.db_set_babies(P.dim)
        assert cc.db_kind() == 5 and cc.db_group() == group
        assert np.array_equal(sender.computeSimilarity(gq).export(), gs)
        path = str(tmp_path / "db.bin")
        cc.db_save(path)
        # ciphertext by ciphertext (the adapter's path), then the file, into the same context
        load_db(cc, dbc, n)
        assert cc.db_group() == group
        assert np.array_equal(sender.computeSimilarity(gq).export(), gs)
        cc.db_load(path)
        assert cc.db_group() == group and cc.db_stats()[0] == n
        assert np.array_equal(sender.indexScenario(gq).export(), want_idx)
        # a context that keeps every database ciphertext-major reads the same file and computes the same ciphertexts
        monkeypatch.setenv("HYDIA_DB_CT_MAJOR", "1")
        c2 = make_ctx(im, P)
        monkeypatch.delenv("HYDIA_DB_CT_MAJOR")
        load_keys(c2, K, K.rotations)
        c2.set_matvec("hoisted")
        c2.db_load(path)
        assert c2.db_group() == 0
        g2 = c2.import_ct(q.data(), q.scale)
        assert np.array_equal(im.DiagonalSender(c2, n).indexScenario(g2).export(), want_idx)
        c2.db_set_babies(P.dim)  # a no-op declaration stays legal
        c2.close()
    finally:
        cc.set_matvec("auto")
        if own is not None:
            own.close()


def test_db_relayout_without_room_fails_cleanly_and_bad_declarations_are_refused(im):
    """hydia_db_set_babies on a database that is laid out for another form re-orders it through a second buffer.  A 2^20-vector
    database (142 GiB, beside 12 GiB of keys) cannot have one on a 288 GiB GPU: the call must fail with HYDIA_ERR_DEVICE before anything is touched — same
    ciphertexts, same declared form, same layout, and the context still answers queries.  Declarations that are not a form
    (0, 1, 3, negative, > vector_dim) are argument errors (round-3 advice: 0 used to mark a hoisted database as pre-rotated)."""
    cc = im.Context()
    try:
        cc.set_matvec("hoisted")
        cc.fill_eval_keys_random(1)  # 12 GiB of keys beside the database: two copies of it cannot both fit any more
        cc.db_fill_random(1 << 20, 5)
        assert cc.db_kind() == 5 and cc.db_babies() == 512 and cc.db_group() == 8 and cc.db_stats()[2] > 140 << 30
        before = {t: cc.db_export_ct(t) for t in (0, 777, 32767)}
        with pytest.raises(im.HydiaError) as e:
            cc.db_set_babies(256)
        assert e.value.code == -3 and "second buffer" in str(e.value)  # HYDIA_ERR_DEVICE
        assert cc.db_kind() == 5 and cc.db_babies() == 512 and cc.db_group() == 8
        for t, want in before.items():
            assert np.array_equal(cc.db_export_ct(t), want), t
        for bad in (0, 1, 3, -2, 1024, 48):
            with pytest.raises(im.HydiaError) as e:
                cc.db_set_babies(bad)
            assert e.value.code == -1, bad  # HYDIA_ERR_ARG
        cc.db_set_babies(512)  # the form it has: a no-op
        # a smaller database on the same context: the re-ordering goes through and comes back
        cc.db_fill_random(9 * 16384, 6)
        want = cc.db_export_ct(600)
        cc.db_set_babies(256)
        assert cc.db_kind() == 6 and cc.db_babies() == 256 and np.array_equal(cc.db_export_ct(600), want)
        cc.db_set_babies(512)
        assert cc.db_kind() == 5 and np.array_equal(cc.db_export_ct(600), want)
        # rotation ranges are validated on the unsigned values before anything is allocated (first + count must not wrap)
        rng = np.random.default_rng(0)
        q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
        gq = cc.import_ct(q, cc.delta)
        snd = im.DiagonalSender(cc, 9 * 16384)
        for first, count in ((0x7fffffff, 0x7fffffff), (512, 1), (0, 513), (0xffffffff, 2)):
            with pytest.raises(im.HydiaError) as e:
                snd.rotateQueryRange(gq, first, count)
            assert e.value.code == -1, (first, count)
        assert snd.rotateQueryRange(gq, 510, 2).shape()[0] == 2
        del gq
    finally:
        cc.set_matvec("auto")
        cc.close()

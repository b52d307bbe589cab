"""Full-ring (N = 2^15, the production kernels: register-radix NTT, FP64 butterflies, packed DB and keys, fused passes, lanes)
BIT-EXACT parity of the whole path against the CPU oracle, inside the -m gpu run (round-1 review: the client side and the
multi-block sender were only covered at N = 2^11 or through 1e-4 on decrypted scores):
  (i)   key generation, query encryption, enrolled database ciphertexts, decryption (decoded doubles too)
  (ii)  a 3-block database (n = 40000): computeSimilarity, indexScenario, membershipScenario
  (iii) the GPU comparator on the 16384 inputs of the reference's published transfer curve (tools/figures/signApprox.csv,
        column `combined`; tolerance 1e-4 = src/main_accuracy.cpp:359-360) — compared DIRECTLY with the published values
  (iv)  test/2_11.dat like 2_10: sender bit-exact, answers `true`, `[0]`
Nothing here reads /root/reference: fixtures are tests/golden/*.npz."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-4
KEY_SEED = 20250725


@pytest.fixture(scope="module")
def im():
    import image_matching_amd as im
    return im


@pytest.fixture(scope="module")
def full(im):
    P = O.Params()
    K = O.Keys(P, KEY_SEED)
    cc = im.Context()
    assert np.array_equal(cc.moduli, P.moduli) and np.array_equal(cc.roots, P.roots)
    cc.keygen(KEY_SEED)
    yield P, K, O.Oracle(P, K), cc
    cc.close()


def test_client_side_bit_exact_full_ring(im, full):
    P, K, Or, cc = full
    # (i) keys: secret, public, relinearisation, first / last hoisted rotation, an EvalSum rotation
    assert np.array_equal(cc.export_secret_key(), K.s_ntt())
    assert np.array_equal(cc.export_public_key(), K.pk())
    assert np.array_equal(cc.export_eval_key(0), K.relin())
    for r in (1, 2, 255, 511, 512, 8192):
        assert np.array_equal(cc.export_eval_key(r), K.rot_key(r)), r
    # query encryption (receiver_diag.cpp:13-26) and general encryption incl. the 16384-slot encode FFT
    qv = np.arange(1.0, 513.0)
    gq = im.DiagonalReceiver(cc, 10).encryptQuery(qv, seed=5, nonce=1)
    assert np.array_equal(gq.export()[0], Or.encrypt_query(qv, 5, 1).data())
    rng = np.random.default_rng(3)
    z = rng.uniform(-1, 1, (2, P.slots))
    g = cc.encrypt(z, 11, 40)
    data = g.export()
    cts = [Or.encrypt(z[i], 11, 40 + i) for i in range(2)]
    for i in range(2):
        assert np.array_equal(data[i], cts[i].data()), i
    # decryption: decoded doubles are IDENTICAL (same IEEE operation order), at level 0, on 3 components and on one limb
    dec = cc.decrypt(g)
    for i in range(2):
        assert np.array_equal(dec[i], Or.decrypt(cts[i]))
        assert np.abs(dec[i] - z[i]).max() < 1e-7
    d = Or.mult_norelin(cts[0], cts[1])
    assert np.array_equal(cc.decrypt(cc.import_ct(d.data(), d.scale))[0], Or.decrypt(d))
    cur = cts[0]
    while cur.nl > 1:
        cur = Or.mult(cur, cur)
    assert np.array_equal(cc.decrypt(cc.import_ct(cur.data(), cur.scale))[0], Or.decrypt(cur))
    # enrolment (enroller_diag.cpp:12-53 incl. k_diag_pack at 32 sub-blocks per ciphertext): ragged database, zero vector
    n = 20000
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    db[7] = 0.0
    a, b = db.copy(), db.copy()
    dbc = Or.enroll(a, 99)
    im.DiagonalEnroller(cc, n).serializeDB(b, seed=99)
    assert np.array_equal(a, b) and cc.db_stats()[:2] == (n, len(dbc)) and len(dbc) == 1024
    for t in (0, 1, 7, 511, 512, 777, 1023):
        assert np.array_equal(cc.db_export_ct(t), dbc[t].data()), t


@pytest.mark.parametrize("matvec", ["hoisted", "bsgs"])
def test_three_block_sender_bit_exact_full_ring(im, full, matvec):
    """(ii) n = 40000 -> G = 3 blocks on the default fast path (batched X = 3 tails, uneven lane split), in BOTH forms of the mat-vec:
    the reference's 511 hoisted rotations (sender_diag.cpp:22-26) and the baby-step / giant-step form (31 babies, 16 giant steps per
    block on pre-rotated diagonals) — each equal to the oracle's restatement of that form, bit for bit."""
    P, K, Or, cc = full
    cc.set_matvec(matvec)
    n = 40000
    rng = np.random.default_rng(77)
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    planted = [5, 20000, n - 1]
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    cos = (db / np.linalg.norm(db, axis=1, keepdims=True)) @ (query / np.linalg.norm(query))
    a = db.copy()
    dbc = Or.enroll(a, 8, matvec=matvec)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=8)
    cc.set_matvec("auto")
    assert cc.db_kind() == (6 if matvec == "bsgs" else 5)
    q = Or.encrypt_query(query, 2, 9)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    gq = receiver.encryptQuery(query, seed=2, nonce=9)
    assert np.array_equal(gq.export()[0], q.data())
    sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq)
    assert len(sim) == 3 and gsim.shape()[:3] == (3, 2, P.nQ - 1)
    gs = gsim.export()
    for g in range(3):
        assert np.array_equal(gs[g], sim[g].data()), g
    scores = cc.decrypt(gsim).reshape(-1)
    assert np.abs(scores[:n] - cos).max() < TOL and np.abs(scores[n:]).max() < TOL
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
    gi = gidx.export()
    for g in range(3):
        assert np.array_equal(gi[g], idx[g].data()), g
    assert receiver.decryptIndex(gidx) == planted == Or.decrypt_index(idx)
    mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
    assert np.array_equal(gmem.export()[0], mem.data())
    assert receiver.decryptMembership(gmem) is True


@pytest.mark.parametrize("blocks,matvec,babies,group,checks", [(4, None, 128, 8, "sim idx mem"), (13, None, 256, 2, "idx"),
                                                               (10, "hoisted", 512, 2, "sim")])
def test_auto_tiers_and_headline_kernel_bit_exact_full_ring(im, full, blocks, matvec, babies, group, checks):
    """Round-3 review: the auto rule's 128-baby (4-12 blocks: what BASELINE config 4's database and every 8-block shard of config 5
    get) and 256-baby (13-40 blocks) splits were compared with the oracle nowhere, and the headline loop-B kernel (k_hydia_tensor24 on
    the group-sequential layout, hoisted databases of more than 8 blocks) only at N = 2^11.  Here at N = 2^15 / dim 512 on ragged
    databases: similarity, index and membership ciphertexts equal the oracle's restatement of the same split, bit for bit."""
    P, K, Or, cc = full
    n = blocks * P.slots - 5
    rng = np.random.default_rng(1000 + blocks)
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.int8).astype(np.float64)
    planted = [3, n // 2, n - 1]
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    cc.set_matvec("auto" if matvec is None else matvec)
    try:
        dbc = Or.enroll(db.copy(), 8, matvec=matvec)
        assert dbc.babies == babies == (O.auto_babies(512, blocks) if matvec is None else 512)
        im.DiagonalEnroller(cc, n).serializeDB(db, seed=8)
    finally:
        cc.set_matvec("auto")
    del db
    assert cc.db_babies() == babies and cc.db_kind() == (5 if babies == 512 else 6) and cc.db_group() == group
    for t in (0, 511, 512 * (blocks - 1) + 130, 512 * blocks - 1):
        assert np.array_equal(cc.db_export_ct(t), dbc[t].data()), t
    q = Or.encrypt_query(query, 2, 9)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    gq = receiver.encryptQuery(query, seed=2, nonce=9)
    # (the oracle runs on the box's 16 cores: each case takes the checks that are new for it — index = similarity + comparator)
    if "sim" in checks:
        sim = Or.compute_similarity(q, dbc, n)
        gs = sender.computeSimilarity(gq).export()
        assert len(sim) == blocks
        for g in range(blocks):
            assert np.array_equal(gs[g], sim[g].data()), g
        del sim, gs
    if "idx" in checks:
        idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
        gi = gidx.export()
        for g in range(blocks):
            assert np.array_equal(gi[g], idx[g].data()), g
        assert receiver.decryptIndex(gidx) == planted == Or.decrypt_index(idx)
        del idx, gidx
    else:
        assert receiver.decryptIndex(sender.indexScenario(gq)) == planted
    if "mem" in checks:
        mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
        assert np.array_equal(gmem.export()[0], mem.data()) and receiver.decryptMembership(gmem) is True
    del dbc, gq
    cc.db_alloc(1)  # give the HBM back to the tests that follow


@pytest.mark.parametrize("blocks,pick", [(16, 1), (24, 1), (8, 1), (16, 0), (16, 2)])
def test_loop_b_against_host_recomputation_full_ring(blocks, pick):
    """The headline kernel directly: loop B as a query runs it (Context::similarity_accumulate_rot) at N = 2^15, dim = 512 — the
    24-bit-halves kernel on the group-sequential layout (16 and 24 blocks with 46-bit residues, 16 with 48-bit ones), the 128-bit
    kernel on ciphertext-major databases (8 blocks; 16 with the layout forced) — against unsigned __int128 on the host: 0 mismatches over every (block, limb, coefficient).
    tests/csrc/loop_b_check.cpp, built by __graft_entry__.build(); this is the check that found ROCm 7.2's miscompile of that kernel."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "loop_b_check")
    assert os.path.exists(exe), "tests/csrc/loop_b_check is built by __graft_entry__.build()"
    r = subprocess.run([exe, str(blocks), "512", "15", str(pick)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("group-sequential" if (pick and blocks > 8) else "ciphertext-major") in r.stdout
    assert ("46-bit" if (pick == 1 and blocks > 8) else "48-bit") in r.stdout
    assert " 0 mismatches of %d " % (blocks * 12 * 32768) in r.stdout


def test_config3_one_full_block_bit_exact_full_ring(im, full):
    """BASELINE config 3 in its stated form: n = 16384 exactly (one FULL block, all 32 sub-blocks of every ciphertext populated):
    similarity, index and membership ciphertexts equal the oracle's; answers = the planted matches, incl. the last slot."""
    P, K, Or, cc = full
    n = 16384
    rng = np.random.default_rng(14)
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    planted = [0, 8191, n - 1]
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    cos = (db / np.linalg.norm(db, axis=1, keepdims=True)) @ (query / np.linalg.norm(query))
    dbc = Or.enroll(db.copy(), 8)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=8)
    assert cc.db_stats()[:2] == (n, 512) and len(dbc) == 512
    q = Or.encrypt_query(query, 2, 9)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    gq = receiver.encryptQuery(query, seed=2, nonce=9)
    sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq)
    assert len(sim) == 1 and np.array_equal(gsim.export()[0], sim[0].data())
    assert np.abs(cc.decrypt(gsim)[0] - cos).max() < TOL
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
    assert np.array_equal(gidx.export()[0], idx[0].data())
    assert receiver.decryptIndex(gidx) == planted == Or.decrypt_index(idx)
    mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
    assert np.array_equal(gmem.export()[0], mem.data()) and receiver.decryptMembership(gmem) is True


def test_five_special_primes_below_2p48_full_ring(im):
    """Round 5: a caller-supplied chain with FIVE special primes (here 47-bit ones: every limb but q_0 then runs on the FP64 pipe — the secondary
    configuration of tools/exp_fp64_special_primes.py) goes through the FUSED pipeline: the ModDown conversions take the five-source
    instantiation of the narrow column-fused kernel (k_ntt15_colfuse8<*, 5>).  Keys, query, similarity and index ciphertexts of a one-block
    database equal the oracle's on the same chain, bit for bit; the answers are the planted matches."""
    from sympy import isprime
    base = O.Params()
    M = 2 << 15
    c, p5 = (1 << 47) - ((1 << 47) % M) + 1, []
    while len(p5) < 5:
        c -= M
        if isprime(c):
            p5.append(c)
    moduli = np.array([int(x) for x in base.moduli[:base.nQ]] + p5, dtype=np.uint64)
    base.close()
    P = O.Params(moduli=moduli, n_p=5)
    assert P.nP == 5 and P.dnum == 3
    K = O.Keys(P, 31)
    Or = O.Oracle(P, K)
    cc = im.Context(im.default_params(), 0, moduli=moduli, roots=P.roots, n_p=5)
    assert np.array_equal(cc.moduli, P.moduli) and cc.nP == 5
    cc.keygen(31)
    assert np.array_equal(cc.export_eval_key(0), K.relin()) and np.array_equal(cc.export_eval_key(1), K.rot_key(1))
    n = 3000
    rng = np.random.default_rng(15)
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    planted = [7, n - 1]
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    query = np.ones(512)
    for matvec in ("hoisted",):  # the reference's form: 511 hoisted rotations (loop A's five-source ModDown), one relinearisation (five-source merged ModDown + Rescale)
        dbc = Or.enroll(db.copy(), 8, **({"matvec": matvec} if matvec else {}))
        cc.set_matvec(matvec or "auto")
        im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=8)
        q = Or.encrypt_query(query, 2, 9)
        receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
        gq = receiver.encryptQuery(query, seed=2, nonce=9)
        assert np.array_equal(gq.export()[0], q.data())
        sim, gsim = Or.compute_similarity(q, dbc, n), sender.computeSimilarity(gq)
        assert np.array_equal(gsim.export()[0], sim[0].data()), matvec
        idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
        assert np.array_equal(gidx.export()[0], idx[0].data()), matvec
        assert receiver.decryptIndex(gidx) == planted
    cc.set_matvec("auto")
    cc.close()


def test_gpu_comparator_reproduces_published_transfer_curve(im, full):
    """(iii) chebyshevCompare(0.44, 10) on the GPU, decrypted, against the reference's own published output."""
    P, K, Or, cc = full
    g = np.load(os.path.join(GOLDEN, "sign_approx.npz"))
    x, want = g["input"], g["combined"]
    assert len(x) == cc.slots
    ct = cc.encrypt(x, 2, 1)
    cc.level_reduce(ct, cc.nQ - 1)  # the comparator runs on a level-1 score ciphertext
    out_ct = cc.chebyshev_compare(ct, 0.44, 10)
    assert out_ct.shape()[:3] == (1, 2, 1)
    out = cc.decrypt(out_ct)[0]
    assert np.abs(out - want).max() < TOL
    assert ((out >= 1.0) == (want >= 1.0)).mean() > 0.999  # decisions agree except inside the printed-digit band
    assert abs(out[np.argmin(np.abs(x - 0.9268))] - 2.0) < 1e-3 and abs(out[np.argmin(np.abs(x - 0.1355))]) < 1e-3
    # and it is the oracle's ciphertext, bit for bit
    oc = Or.encrypt(x, 2, 1)
    P.L.hyo_drop_to(P.h, oc.h, P.nQ - 1)
    assert np.array_equal(out_ct.export()[0], Or.chebyshev_compare(oc, 0.44, 10).data())


def test_reference_dataset_2_11_bit_exact_full_ring(im, full):
    """(iv) ./ImageMatching ../test/2_11.dat 5: GPU enroller + sender vs oracle bit for bit; `true`, `[0]`; scores within 1e-4."""
    P, K, Or, cc = full
    g = np.load(os.path.join(GOLDEN, "dataset_2_11.npz"))
    n, query, db = int(g["n"]), g["query"].astype(np.float64), g["db"].astype(np.float64)
    a = db.copy()
    dbc = Or.enroll(a, 99)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=99)
    for t in (0, 3, 511):
        assert np.array_equal(cc.db_export_ct(t), dbc[t].data()), t
    q = Or.encrypt_query(query, 5, 1)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    gq = receiver.encryptQuery(query, seed=5, nonce=1)
    sim = Or.compute_similarity(q, dbc, n)
    gsim = sender.computeSimilarity(gq)
    assert np.array_equal(gsim.export()[0], sim[0].data())
    scores = cc.decrypt(gsim)[0]
    assert np.array_equal(scores, Or.decrypt(sim[0]))
    assert np.abs(scores[:n] - g["cosine"]).max() < TOL and np.abs(scores[n:]).max() < TOL
    idx, gidx = Or.index_scenario(q, dbc, n), sender.indexScenario(gq)
    assert np.array_equal(gidx.export()[0], idx[0].data())
    assert receiver.decryptIndex(gidx) == [0] == list(g["expected_index"])
    mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
    assert np.array_equal(gmem.export()[0], mem.data())
    assert receiver.decryptMembership(gmem) is True and bool(g["expected_membership"])


def test_hoisted_rotations_bit_exact_full_ring(im, full):
    """Loop A on its own at N = 2^15 (the fused path: inner product of the special-prime limbs, ModDown transform whose epilogue forms
    the Q-limb inner product on the fly from the packed keys, automorphism scatter): every one of the 512 rotated query ciphertexts
    equals the oracle's EvalFastRotation (sender_diag.cpp:20-26)."""
    P, K, Or, cc = full
    qv = np.linspace(-1.0, 1.0, 512)
    q = Or.encrypt_query(qv, 9, 3)
    gq = im.DiagonalReceiver(cc, 10).encryptQuery(qv, seed=9, nonce=3)
    assert np.array_equal(gq.export()[0], q.data())
    rot = Or.rotate_query(q)
    grot = im.DiagonalSender(cc, 10).rotateQuery(gq).export()
    assert grot.shape[0] == 512
    for i in range(512):
        assert np.array_equal(grot[i], rot[i].data()), i
    # decrypted: slot s of rotation i holds the (tiled) query coordinate (s + i) mod 512
    z = cc.decrypt(cc.import_ct(grot[5:6], gq.shape()[3]))[0]
    qn = qv / np.linalg.norm(qv)
    assert np.abs(z[:512] - np.roll(qn, -5)).max() < 1e-7


def test_no_match_and_hers_bit_exact_full_ring(im, full):
    """A database without a match (membership `false`, empty index) and approach 4 (HERS) at N = 2^15, ciphertexts equal to the oracle's."""
    P, K, Or, cc = full
    rng = np.random.default_rng(5)
    n = 3000
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    query = np.ones(512)
    a = db.copy()
    dbc = Or.enroll(a, 4)
    im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=4)
    q = Or.encrypt_query(query, 6, 1)
    receiver, sender = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    gq = receiver.encryptQuery(query, seed=6, nonce=1)
    mem, gmem = Or.membership_scenario(q, dbc, n), sender.membershipScenario(gq)
    assert np.array_equal(gmem.export()[0], mem.data())
    assert receiver.decryptMembership(gmem) is False and receiver.decryptIndex(sender.indexScenario(gq)) == []
    del dbc
    # HERS: column packing, 512 query ciphertexts, relinearise + rescale per product (sender_hers.cpp:60-87)
    n = 1200
    db = rng.integers(-99, 100, size=(n, 512)).astype(np.float64)
    db[77] = rng.integers(1, 4, size=512)
    a = db.copy()
    hdb = Or.hers_enroll(a, 4)
    im.HersEnroller(cc, n).serializeDB(db.copy(), seed=4)
    assert cc.db_stats()[1] == len(hdb) == 512
    for t in (0, 300, 511):
        assert np.array_equal(cc.db_export_ct(t), hdb[t].data()), t
    hq = Or.hers_encrypt_query(query, 6, 1000)
    ghq = im.HersReceiver(cc, n).encryptQuery(query, seed=6, nonce=1000)
    hs = im.HersSender(cc, n)
    sim, gsim = Or.hers_compute_similarity(hq, hdb, n), hs.computeSimilarity(ghq).export()
    assert np.array_equal(gsim[0], sim[0].data())
    idx, gidx = Or.hers_index_scenario(hq, hdb, n), hs.indexScenario(ghq)
    assert np.array_equal(gidx.export()[0], idx[0].data())
    assert receiver.decryptIndex(gidx) == [77]
    hmem, ghmem = Or.hers_membership_scenario(hq, hdb, n), hs.membershipScenario(ghq)
    assert np.array_equal(ghmem.export()[0], hmem.data()) and receiver.decryptMembership(ghmem) is True

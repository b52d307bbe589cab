"""Diagnostic: distribution of the comparator's decrypted output over non-matching slots and the membership sum, vs DB size."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im

cc = im.Context()
cc.keygen(21)
for log2n in (14, 17, 20):
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    db = rng.integers(-99, 100, size=(n, 512), dtype=np.int8).astype(np.float64)
    planted = sorted(set([0, n // 2, n - 1]))
    for i in planted:
        db[i] = rng.integers(1, 4, size=512)
    im.DiagonalEnroller(cc, n).serializeDB(db, seed=4)
    r, s = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    q = r.encryptQuery(np.ones(512), seed=5, nonce=1)
    sim = cc.decrypt(s.computeSimilarity(q)).reshape(-1)
    idx = cc.decrypt(s.indexScenario(q)).reshape(-1)
    mem = cc.decrypt(s.membershipScenario(q))[0]
    mask = np.ones(n, bool); mask[planted] = False
    print("2^%d: non-match out mean %.3e std %.3e min %.3e max %.3e sum %.4f | match out %s | membership slot0 %.4f slot1 %.4f (expected %d)" % (
        log2n, idx[mask].mean(), idx[mask].std(), idx[mask].min(), idx[mask].max(), idx[mask].sum(), idx[planted], mem[0], mem[1], 2 * len(planted)), flush=True)
cc.close()

"""Soak: many queries in a row on one context; pool and device memory must reach a steady state and results stay identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cc = im.Context()
cc.keygen(3)
n = 1 << log2n
rng = np.random.default_rng(1)
db = rng.integers(-99, 100, size=(n, 512), dtype=np.int8).astype(np.float64)
db[7] = rng.integers(1, 4, size=512)
im.DiagonalEnroller(cc, n).serializeDB(db, seed=4)
r, s = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
q = r.encryptQuery(np.ones(512), seed=5)
ref = s.indexScenario(q).export()
mem0 = None
every = min(50, max(1, iters // 2))  # at least two checkpoints
t0 = time.time()
for i in range(iters):
    which = i % 3
    out = s.indexScenario(q) if which == 0 else (s.membershipScenario(q) if which == 1 else s.computeSimilarity(q))
    if i % every == every - 1:
        cc.sync()
        live, cached, peak = cc.memory_stats()
        same = np.array_equal(s.indexScenario(q).export(), ref)
        print("iter %d: pool live %.2f GiB cached %.2f GiB peak %.2f GiB, %.1f ms/query, result identical %s" % (
            i + 1, live / 2**30, cached / 2**30, peak / 2**30, (time.time() - t0) / (i + 1) * 1e3, same), flush=True)
        if mem0 is None:
            mem0 = peak
        assert same
cc.sync()
live, cached, peak = cc.memory_stats()
assert peak == mem0, "pool kept growing: %s vs %s" % (peak, mem0)
print("soak ok: %d queries, pool steady at %.2f GiB" % (iters, peak / 2**30))
cc.close()

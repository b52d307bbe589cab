import sys, time, numpy as np
sys.path.insert(0, '.')
import image_matching_amd as im
cc = im.Context(); cc.keygen(5)
n = 1 << 14
cc.db_fill_random(n, 3)
s = im.DiagonalSender(cc, n)
q = im.DiagonalReceiver(cc, n).encryptQuery(np.ones(512), seed=1, nonce=1)
for _ in range(3): r = s.indexScenario(q)
cc.sync()
for k in range(5):
    t0 = time.perf_counter(); r = s.indexScenario(q); t1 = time.perf_counter(); cc.sync(); t2 = time.perf_counter()
    print("enqueue %.2f ms, until done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))

#!/usr/bin/env python3
"""Profiling driver: random keys / DB / query residues (kernel cost is data independent), times computeSimilarity
and indexScenario and prints the HIP-event time of the loop-B tensor kernel.  Usage: prof_similarity.py [log2_n] [iters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "both"
n = 1 << log2n
t0 = time.time()
cc = im.Context()
print("context %.2fs" % (time.time() - t0), flush=True)
t0 = time.time()
cc.fill_eval_keys_random(1)
cc.db_fill_random(n, 2)
print("fill keys+db %.2fs  db=%s" % (time.time() - t0, cc.db_stats()), flush=True)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
snd = im.DiagonalSender(cc, n)
for name, fn in (("computeSimilarity", snd.computeSimilarity), ("indexScenario", snd.indexScenario)):
    if what not in ("both", name):
        continue
    fn(gq); cc.sync()
    cc.kernel_time_reset()
    t0 = time.time()
    for _ in range(iters):
        r = fn(gq)
    cc.sync()
    dt = (time.time() - t0) / iters
    ms, k = cc.kernel_time("hydia_tensor")
    ip, k2 = cc.kernel_time("ks_inner_product")
    G = cc.db_stats()[1] // cc.dim
    gb = (G * cc.dim * 2 * cc.nQ * cc.N * 8 + cc.dim * 2 * cc.nQ * cc.N * 8 + G * 3 * cc.nQ * cc.N * 8) / 1e9
    print("%s n=2^%d: %.3f ms/query -> %.0f vectors/s | tensor kernel %.3f ms/launch (%d) = %.0f GB/s algorithmic | "
          "inner_product %.3f ms total/query" % (name, log2n, dt * 1e3, n / dt, ms / max(k, 1), k, gb / (ms / max(k, 1) / 1e3),
                                                 ip / iters), flush=True)
print("pool (live, cached, peak) GiB:", [round(v / 2 ** 30, 2) for v in cc.memory_stats()])

#!/usr/bin/env python3
"""Same-box A/B of engine switches: for every configuration (a comma-separated list of VAR=value, "-" = defaults) a fresh context
with random keys / database / query residues (kernel cost is data independent), Q indexScenario queries at 2^L vectors, ms per
query.  HBM rates differ by ~2 % from box to box, so configurations are only compared within ONE call.
Usage: ab_env.py L Q cfg [cfg ...]     e.g.  ab_env.py 20 10 - HYDIA_LANES=3 HYDIA_LANES=1"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

L, Q = int(sys.argv[1]), int(sys.argv[2])
n = 1 << L
for rep in range(int(os.environ.get("AB_REPEATS", "1"))):
    for cfg in sys.argv[3:]:
        kv = [] if cfg == "-" else [a.split("=", 1) for a in cfg.split(",")]
        for k, v in kv:
            os.environ[k] = v
        cc = im.Context()
        cc.fill_eval_keys_random(1)
        cc.db_fill_random(n, 2)
        rng = np.random.default_rng(0)
        q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
        gq = cc.import_ct(q, cc.delta)
        snd = im.DiagonalSender(cc, n)
        for _ in range(2):
            r = snd.indexScenario(gq)
        cc.sync()
        t0 = time.time()
        for _ in range(Q):
            r = snd.indexScenario(gq)
        cc.sync()
        ms = (time.time() - t0) / Q * 1e3
        print("2^%d  %-60s %8.3f ms per query" % (L, cfg, ms), flush=True)
        del r, gq, snd
        cc.close()
        for k, _ in kv:
            del os.environ[k]

#!/usr/bin/env python3
"""Loop A alone (DiagonalSender::rotateQuery = the 511 hoisted rotations) with the byte ledger: ms per call and the ledger, for
tools/kernel_rooflines.py.  Usage: prof_rotate.py [calls]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cc = im.Context()
cc.fill_eval_keys_random(1)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
snd = im.DiagonalSender(cc, 16384)
r = snd.rotateQuery(gq)
cc.sync()
im.byte_ledger(1)
t0 = time.time()
for _ in range(Q):
    r = snd.rotateQuery(gq)
cc.sync()
ms = (time.time() - t0) / Q * 1e3
led = im.byte_ledger(0)
out = {"log2n": 14, "scenario": "rotateQuery (loop A)", "queries": Q, "ms_per_query": ms, "db_bytes": 0,
       "ledger": {k: {"launches": v[0], "bytes": v[1]} for k, v in led.items()}}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ledger_rot.json"), "w"), indent=1)
print("rotateQuery: %.3f ms per call" % ms)

#!/usr/bin/env python3
"""Run Q identical indexScenario queries (random keys / DB / query residues: kernel cost is data independent) with the byte
ledger recording, and write {queries, ms_per_query, ledger} to gpurun_out/ledger_q<log2n>.json.  Meant to run under
`rocprofv3 --kernel-trace --stats`, whose per-kernel totals tools/kernel_rooflines.py divides the ledger's bytes by.
Usage: prof_query_ledger.py [log2_n] [queries] [indexScenario|membershipScenario|computeSimilarity]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "indexScenario"
n = 1 << log2n
if os.environ.get("HYDIA_EXP_CHAIN") == "5x47":  # the secondary configuration of tools/exp_fp64_special_primes.py: five 47-bit special primes, dnum 3
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from exp_fp64_special_primes import fp64_chain
    cc = im.Context(im.default_params(), 0, moduli=fp64_chain(5)[0], n_p=5)
else:
    cc = im.Context()
cc.fill_eval_keys_random(1)
cc.db_fill_random(n, 2)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
fn = getattr(im.DiagonalSender(cc, n), what)
fn(gq)            # untimed: builds the packed key shadow, fills the pool
cc.sync()
im.byte_ledger(1)
t0 = time.time()
for _ in range(Q):
    r = fn(gq)
cc.sync()
ms = (time.time() - t0) / Q * 1e3
led = im.byte_ledger(0)
out = {"log2n": log2n, "scenario": what, "queries": Q, "ms_per_query": ms, "db_bytes": cc.db_stats()[2],
       "ledger": {k: {"launches": v[0], "bytes": v[1]} for k, v in led.items()}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ledger_q%d.json" % log2n), "w"), indent=1)
print("%s n=2^%d: %.3f ms/query, %d kernels in the ledger, %d launches per query" %
      (what, log2n, ms, len(led), sum(v[0] for v in led.values()) // Q), flush=True)
del r, gq
cc.close()

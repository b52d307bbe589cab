#!/usr/bin/env python3
"""indexScenario ms per query with loop B and the comparator tail pipelined over chunks of blocks (Context::index_pipelined), for
several (lanes, chunks) settings on the same random database; the answers of all settings must be the same bytes.
Usage: prof_pipe.py <log2 blocks> "<lanes>:<chunks> ..."     (chunks 1 = the one-launch loop B + lane-split tail)"""
import gc
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

G = 1 << int(sys.argv[1])
settings = [tuple(int(v) for v in s.split(":")) for s in sys.argv[2].split()]
rng = np.random.default_rng(0)
digests = {}
for lanes, chunks in settings:
    os.environ["HYDIA_LANES"] = str(lanes)
    os.environ["HYDIA_PIPE"] = str(chunks)
    cc = im.Context()
    cc.set_matvec("hoisted")
    cc.fill_eval_keys_random(1)
    q = np.stack([np.random.default_rng(5).integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
    gq = cc.import_ct(q, cc.delta)
    cc.db_fill_random(G * 16384, 2)
    snd = im.DiagonalSender(cc, G * 16384)
    r = snd.indexScenario(gq)
    cc.sync()
    digests[(lanes, chunks)] = hashlib.sha256(r.export().tobytes()).hexdigest()[:16]
    reps = 6
    t0 = time.time()
    for _ in range(reps):
        r = snd.indexScenario(gq)
    cc.sync()
    ms = (time.time() - t0) / reps * 1e3
    print("blocks %3d  lanes %d  chunks %d : %7.2f ms / query   answer %s   pool %s" % (G, lanes, chunks, ms, digests[(lanes, chunks)], cc.memory_stats()), flush=True)
    del r, snd, gq
    cc.close()
    del cc
    gc.collect()
assert len(set(digests.values())) == 1, digests
print("all settings give the same ciphertext bytes")

#!/usr/bin/env python3
"""One-GPU components of the multi-GPU step model of DESIGN.md section 7: loop A on 1/R of the rotations, and the rest of an
indexScenario query (loop B + per-block tails on rotations that are already there) for the G/R blocks a rank holds.
Usage: prof_scaling_components.py [log2 of the TOTAL database, default 20]  ->  gpurun_out/scaling_components_<log2n>.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

LOG2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G_total = max(1, (1 << LOG2N) // 16384)
out = {"log2n_total": LOG2N, "blocks_total": G_total, "ranks": {}}


def timed(fn, reps=5):
    fn()
    cc.sync()
    t0 = time.time()
    for _ in range(reps):
        fn()
    cc.sync()
    return (time.time() - t0) / reps * 1e3


for R in (1, 2, 4, 8):
    G = max(1, G_total // R)
    cc = im.Context()
    cc.set_matvec("hoisted")
    cc.fill_eval_keys_random(1)
    cc.db_fill_random(G * 16384, 2)
    rng = np.random.default_rng(0)
    q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
    gq = cc.import_ct(q, cc.delta)
    snd = im.DiagonalSender(cc, G * 16384)
    rot = snd.rotateQuery(gq)
    share = 512 // R
    keep = []
    t_a = timed(lambda: keep.append(snd.rotateQueryRange(gq, 512 - share if R > 1 else 0, share)) or keep.clear())
    t_rest = timed(lambda: snd.indexScenarioRotated(rot))
    t_full = timed(lambda: snd.indexScenario(gq))
    row = {"blocks_per_gpu": G, "loop_a_share_ms": round(t_a, 3), "rest_on_given_rotations_ms": round(t_rest, 3),
           "whole_query_replicated_loop_a_ms": round(t_full, 3)}
    cc.set_matvec("auto")
    if cc.auto_babies(G) < cc.dim:  # the split the auto rule picks for this many blocks, when it is not the hoisted form timed above
        cc.db_fill_random(G * 16384, 2)
        row["auto_babies"] = cc.db_babies()
        row["whole_query_auto_split_ms"] = round(timed(lambda: snd.indexScenario(gq)), 3)
    out["ranks"][R] = row
    print(R, row, flush=True)
    del rot, gq, snd
    cc.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scaling_components_%d.json" % LOG2N), "w"), indent=1)

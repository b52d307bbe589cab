#!/usr/bin/env python3
"""indexScenario ms per query for every split of the mat-vec (babies B) at every database size: the data behind the auto rule
(Context::auto_babies).  Random keys / residues (cost is data independent).  Usage: prof_matvec_sweep.py [max log2 blocks, default 6]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

MAXLG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
BLOCKS = [int(v) for v in sys.argv[2].split()] if len(sys.argv) > 2 else [1 << lg for lg in range(MAXLG + 1)]
BABIES = [int(v) for v in sys.argv[3].split()] if len(sys.argv) > 3 else [32, 64, 128, 256, 512]
cc = im.Context()
cc.fill_eval_keys_random(1)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
lines = ["blocks  " + "  ".join("B=%-6d" % b for b in BABIES) + "  auto"]
for G in BLOCKS:
    row = []
    for B in BABIES:
        cc.set_matvec("hoisted" if B == 512 else B)
        cc.db_fill_random(G * 16384, 2)
        assert cc.db_babies() == B
        snd = im.DiagonalSender(cc, G * 16384)
        snd.indexScenario(gq)
        cc.sync()
        reps = 5 if G <= 16 else 3
        t0 = time.time()
        for _ in range(reps):
            r = snd.indexScenario(gq)
        cc.sync()
        row.append((time.time() - t0) / reps * 1e3)
        del r
    cc.set_matvec("auto")
    lines.append("%-6d  " % G + "  ".join("%8.2f" % t for t in row) + "  B=%d" % cc.auto_babies(G))
    print(lines[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "matvec_sweep.txt"), "w").write("\n".join(lines) + "\n")

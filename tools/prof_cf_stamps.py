#!/usr/bin/env python3
"""Where does a workgroup of the narrow column-fused conversion kernel spend its time?  A DIAGNOSTIC library (tools/ab/cfstamp.so: colfuse.hip with
cycle-counter stamps in wave 0 of every workgroup, accumulated per section with atomics; built by hand from a scratch copy, see
profiles/r05/experiments.txt entry 14) runs one 2^L indexScenario; the sums are divided by the workgroup / target counts.
Usage: HYDIA_LIBPATH=tools/ab/cfstamp.so HYDIA_LANES=1 python tools/prof_cf_stamps.py [L]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402
from image_matching_amd import hydia as _h  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << L
lib = _h.load_library()
cc = im.Context()
cc.fill_eval_keys_random(1)
cc.db_fill_random(n, 2)
rng = np.random.default_rng(0)
q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
gq = cc.import_ct(q, cc.delta)
snd = im.DiagonalSender(cc, n)
r = snd.indexScenario(gq)
cc.sync()
buf = (C.c_ulonglong * 32)()
lib.hydia_debug_cf_stamps(None, 1)
r = snd.indexScenario(gq)
cc.sync()
lib.hydia_debug_cf_stamps(buf, 0)
allv = list(buf)
PRO = bool(os.environ.get("CF_STAMPS_PROLOGUE"))  # the second diagnostic build: the prologue cut finer (sections 0-5), the whole target loop as one
names = {0: "kernel entry -> table rows arrived, staging stores issued", 1: "first barrier", 2: "wait until every source load has arrived",
         3: "the inverse transforms (paired)", 4: "dropped-limb correction (+ barrier)", 5: "the target loop", 6: "-", 7: "-"} if PRO else {
         0: "before the target loop (source loads, inverse transforms, dropped-limb correction)",
         1: "conversion of an FP64 target (+ wait for its twiddle row, + the previous target's store tail)", 7: "conversion of a 60-bit target (same)",
         2: "phase A butterflies + exchange write", 3: "barrier 1", 4: "phase B (exchange read, butterflies, exchange write)", 5: "barrier 2",
         6: "phase C + the 8 global stores"}
order = (0, 1, 2, 3, 4, 5) if PRO else (0, 1, 7, 2, 3, 4, 5, 6)
for kind, o in (("k_ntt15_colfuse8<false> (ModUp digits, loop A's ModDown)", 0), ("k_ntt15_colfuse8<true> (merged ModDown + Rescale, Rescale)", 16)):
    v = allv[o:o + 16]
    wgs, tgts = v[15], v[14]
    if not wgs:
        continue
    tot = sum(v[i] for i in range(8))
    print("%s: %d workgroups, %.1f targets each, %.0f cycles per workgroup, %.0f per target iteration" % (kind, wgs, tgts / wgs, tot / wgs, (tot - v[0]) / max(tgts, 1)))
    for i in order:
        print("  %5.1f %%  %8.0f cycles per workgroup  %s" % (100.0 * v[i] / tot, v[i] / wgs, names[i]))
del r, gq, snd
cc.close()

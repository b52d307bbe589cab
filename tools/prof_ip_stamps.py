#!/usr/bin/env python3
"""In-kernel section timing of the fused inner-product kernels (k_ntt15_p2_ip_all / k_ntt15_p2_ip): a DIAGNOSTIC library built by hand
(ntt15.hip with shader-clock stamps in p2_body for the inner-product modes: profiles/r05/ip_stamps.patch) as HYDIA_LIBPATH=tools/ab/ipstamp.so.
Usage: HYDIA_LIBPATH=tools/ab/ipstamp.so HYDIA_LANES=1 python tools/prof_ip_stamps.py [L]"""
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
lib.hydia_debug_ip_stamps(None, 1)
r = snd.indexScenario(gq)
cc.sync()
lib.hydia_debug_ip_stamps(buf, 0)
v = list(buf)
names = ["entry -> end of phase A (digit loads arrive, phase A, exchange write)", "phase B (exchange, butterflies, exchange)",
         "phase C, first group (exchange read, stages 13-14, key loads, products; + the tail's phase C')", "phase C, second group", "the tail's phases B', A' + stores"]
for kind, o in (("FP64 rows without the tail (the Q limbs' rows)", 0), ("FP64 rows with the tail (the dropped limb)", 8), ("60-bit rows without the tail", 16),
                ("60-bit rows with the tail (the special-prime rows)", 24)):
    w = v[o:o + 8]
    if not w[7]:
        continue
    tot = sum(w[:5])
    print("%s: %d workgroups, %.0f cycles each" % (kind, w[7], tot / w[7]))
    for i in range(5):
        if w[i]:
            print("  %5.1f %%  %8.0f cycles  %s" % (100.0 * w[i] / tot, w[i] / w[7], names[i]))
del r, gq, snd
cc.close()

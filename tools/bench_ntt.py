#!/usr/bin/env python3
"""NTT microbenchmark (N = 2^15): ms and TB/s per limb-transform for the FP64 limbs (one-pass vs HYDIA_NTT_2PASS=1) and the
60-bit limbs, forward and inverse, at several batch sizes.  Algorithmic bytes = 512 KiB per limb-transform."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im  # noqa: E402

if not os.environ.get("HYDIA_NTT_2PASS"):
    os.environ.setdefault("HYDIA_NTT_1PASS", "1")
os.environ.setdefault("HYDIA_NTT_1PASS_MIN", "1")  # the microbenchmark compares the kernels at every batch size
cc = im.Context()
tag = "2-pass" if os.environ.get("HYDIA_NTT_2PASS") else "1-pass"
for polys in (2, 64, 1024):
    for name, first, cnt in (("fp64 limbs 1-11", 1, 11), ("60-bit q0+P", 12, 4)):
        for inv in (False, True):
            ms = cc.bench_ntt(polys, first, cnt, inv, 20 if polys < 1024 else 5)
            lp = polys * cnt
            print("%-7s %-16s %s polys=%5d: %8.3f ms  %7.3f us/limb-poly  %6.2f TB/s algorithmic (512 KiB each)"
                  % (tag if first == 1 else "2-pass", name, "inv" if inv else "fwd", polys, ms, ms * 1e3 / lp, lp * 524288 / ms / 1e9), flush=True)
cc.close()

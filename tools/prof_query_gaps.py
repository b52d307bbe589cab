#!/usr/bin/env python3
"""How much of a small query is NOT kernel execution?  (what a single-launch relinearise + rescale could remove at most.)
Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prof_query_gaps.py run L`: Q indexScenario queries at 2^L vectors,
each followed by a sync and a 30 ms pause, so that every query is one burst of kernels in the trace.  Then
`python3 tools/prof_query_gaps.py report DIR` : per burst — wall (first start to last end), the union of the kernel intervals (some kernel
running on ANY stream), the idle remainder, launches, and the sum of durations (> union where the two lanes overlap)."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(L, Q=6):
    import numpy as np
    sys.path.insert(0, ROOT)
    import image_matching_amd as im
    n = 1 << L
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
    for _ in range(Q):
        time.sleep(0.03)
        t0 = time.time()
        r = snd.indexScenario(gq)
        cc.sync()
        print("query wall (host): %.3f ms" % ((time.time() - t0) * 1e3), flush=True)
    del r, gq, snd
    cc.close()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    bursts, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-8:]) > 10_000_000:  # > 10 ms of nothing: the pause between two queries
            bursts.append(cur)
            cur = []
        cur.append(r)
    bursts.append(cur)
    bursts = [b for b in bursts if 50 < len(b) < 2000][-5:]  # the timed queries (set-up bursts are far longer)
    print("%8s %10s %10s %10s %12s" % ("launches", "wall ms", "busy ms", "idle ms", "sum of kernels"))
    for b in bursts:
        wall = (max(x[1] for x in b) - b[0][0]) / 1e6
        busy, end = 0, b[0][0]
        for s, e, _ in b:
            if e > end:
                busy += e - max(s, end)
                end = e
        print("%8d %10.3f %10.3f %10.3f %12.3f" % (len(b), wall, busy / 1e6, wall - busy / 1e6, sum(e - s for s, e, _ in b) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2])

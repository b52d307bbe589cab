#!/usr/bin/env python3
"""HyDia (approach 5) vs HERS (approach 4) on the same GPU stack — the comparison of tools/figures/approach{4,5}.csv in the
reference.  One JSON line per database size: ms per query for index and membership, both approaches.
Synthetic data of tools/gen_dataset.sh's distribution; results are checked (index == planted matches)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im  # noqa: E402


def timed(cc, fn, reps):
    fn()
    cc.sync()
    t0 = time.time()
    for _ in range(reps):
        r = fn()
    cc.sync()
    return (time.time() - t0) / reps * 1e3, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, nargs="+", default=[10, 14, 17])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    cc = im.Context()
    cc.keygen(20250725)
    for l2 in args.log2n:
        n = 1 << l2
        rng = np.random.default_rng(l2)
        db = rng.integers(-99, 100, size=(n, 512), dtype=np.int8).astype(np.float64)
        planted = sorted(set([0, n // 2, n - 1]))
        for i in planted:
            db[i] = rng.integers(1, 4, size=512)
        query = np.ones(512)
        row = {"log2n": l2, "n": n}
        for name, Enr, Rec, Snd in (("hydia", im.DiagonalEnroller, im.DiagonalReceiver, im.DiagonalSender),
                                    ("hers", im.HersEnroller, im.HersReceiver, im.HersSender)):
            t0 = time.time()
            Enr(cc, n).serializeDB(db.copy(), seed=3)
            cc.sync()
            t_enroll = time.time() - t0
            r, s = Rec(cc, n), Snd(cc, n)
            t0 = time.time()
            q = r.encryptQuery(query, seed=5)
            cc.sync()
            t_query = (time.time() - t0) * 1e3
            reps = args.reps if name == "hydia" or l2 <= 17 else 1
            ms_idx, idx = timed(cc, lambda: s.indexScenario(q), reps)
            ms_mem, mem = timed(cc, lambda: s.membershipScenario(q), reps)
            ok = r.decryptIndex(idx) == planted and r.decryptMembership(mem) is True
            row[name] = {"enroll_s": round(t_enroll, 2), "query_encrypt_ms": round(t_query, 2), "query_ciphertexts": len(q),
                         "index_ms": round(ms_idx, 2), "membership_ms": round(ms_mem, 2), "correct": bool(ok)}
        row["hers_over_hydia_index"] = round(row["hers"]["index_ms"] / row["hydia"]["index_ms"], 2)
        print(json.dumps(row), flush=True)
    cc.close()


if __name__ == "__main__":
    main()

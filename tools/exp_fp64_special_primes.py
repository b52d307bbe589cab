#!/usr/bin/env python3
"""VERDICT r4 item 7 (a SECONDARY figure, never the headline): what does an MI355X-first choice of special primes buy?
The reference's parameter set (OpenFHE defaults) has dnum = 3 and four ~60-bit special primes P: every P-limb butterfly runs on the
lazy 60-bit integer path (17 instructions, v_mad_u64_u32 chains) where a 45-bit limb takes 7 FP64 instructions.  Through
hydia_ctx_create_custom the same ciphertext chain Q (60 + 11 x 45 bits) gets P = four 47-bit primes (188 bits) and dnum = 4 — digits of
three limbs, the widest 60 + 45 + 45 = 150 bits <= 188, so the key-switching noise bound holds; log2(QP) = 743 < 881 — and EVERY limb
but q_0 runs on the FP64 pipe.  Prints ms per indexScenario query for both chains at 2^L vectors (random residues: kernel cost is data
independent) and, at 2^14 with real ciphertexts, checks the decrypted index and the scores against plaintext cosine (< 1e-4).
Usage: exp_fp64_special_primes.py [L=20] [queries=10]"""
import json
import os
import sys
import time

import numpy as np
from sympy import isprime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402


def fp64_chain(count=4, bits=47):
    d, mod, _ = im.describe_params(im.default_params())
    nq = d["n_q"]
    M = 2 << 15
    c = (1 << bits) - ((1 << bits) % M) + 1
    p = []
    while len(p) < count:
        c -= M
        if isprime(c):
            p.append(c)
    return np.array(list(mod[:nq]) + p, dtype=np.uint64), nq


def timed(cc, n, Q):
    cc.fill_eval_keys_random(1)
    cc.db_fill_random(n, 2)
    rng = np.random.default_rng(0)
    q = np.stack([rng.integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
    gq = cc.import_ct(q, cc.delta)
    snd = im.DiagonalSender(cc, n)
    for _ in range(2):
        r = snd.indexScenario(gq)
    cc.sync()
    t0 = time.time()
    for _ in range(Q):
        r = snd.indexScenario(gq)
    cc.sync()
    ms = (time.time() - t0) / Q * 1e3
    del r, gq, snd
    return ms


def check(cc, n=16384, dim=512):
    """real ciphertexts: planted matches found, scores within 1e-4 of plaintext cosine"""
    rng = np.random.default_rng(3)
    db = rng.integers(-99, 100, size=(n, dim)).astype(np.float64)
    planted = [0, n // 2, n - 1]
    for i in planted:
        db[i] = rng.integers(1, 4, size=dim)
    query = np.ones(dim)
    cos = (db @ query) / (np.linalg.norm(db, axis=1) * np.linalg.norm(query))
    cc.keygen(11)
    im.DiagonalEnroller(cc, n).serializeDB(db.copy(), seed=12)
    rcv, snd = im.DiagonalReceiver(cc, n), im.DiagonalSender(cc, n)
    qc = rcv.encryptQuery(query, seed=13)
    scores = cc.decrypt(snd.computeSimilarity(qc))[0][:n]
    idx = rcv.decryptIndex(snd.indexScenario(qc))
    return {"max_score_error": float(np.abs(scores - cos).max()), "index": idx, "index_ok": idx == planted,
            "membership": bool(rcv.decryptMembership(snd.membershipScenario(qc)))}


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    if len(sys.argv) > 3 and sys.argv[3] == "5x47":  # dnum 3 with FIVE 47-bit special primes (235 bits >= the 195-bit first digit)
        moduli, nq = fp64_chain(5)
        cc = im.Context(im.default_params(), 0, moduli=moduli, n_p=5)
        print("five 47-bit special primes, dnum 3: check at 2^14:", check(cc), flush=True)
        cc.close()
        cc = im.Context(im.default_params(), 0, moduli=moduli, n_p=5)
        print("five 47-bit special primes, dnum 3: 2^%d: %.3f ms per query" % (L, timed(cc, 1 << L, Q)), flush=True)
        cc.close()
        return
    if len(sys.argv) > 3 and sys.argv[3] == "6x41":  # SIX special primes (246 bits): outside the fused pipeline (cf_ok() false) — the unfused kernels must serve it
        moduli, nq = fp64_chain(6, 41)
        cc = im.Context(im.default_params(), 0, moduli=moduli, n_p=6)
        print("six 41-bit special primes, dnum 3 (unfused pipeline): check at 2^14:", check(cc), flush=True)
        cc.close()
        return
    moduli, nq = fp64_chain()
    out = {"log2n": L, "queries": Q, "special_primes_fp64": [int(x) for x in moduli[nq:]], "dnum_fp64": 4}
    for name in ("reference chain (dnum 3, P = 4 x 60 bit)", "FP64 special primes (dnum 4, P = 4 x 47 bit)"):
        fp = name.startswith("FP64")
        cc = im.Context(im.default_params(dnum=4), 0, moduli=moduli, n_p=4) if fp else im.Context()
        ms = timed(cc, 1 << L, Q)
        cc.close()
        cc = im.Context(im.default_params(dnum=4), 0, moduli=moduli, n_p=4) if fp else im.Context()
        chk = check(cc)
        cc.close()
        out[name] = {"ms_per_query": round(ms, 3), "vectors_per_s": round((1 << L) / ms * 1e3), "check_2p14": chk}
        print("%-50s 2^%d: %8.3f ms per query   check at 2^14: %s" % (name, L, ms, chk), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp64_special_primes_q%d.json" % L), "w"), indent=1)


if __name__ == "__main__":
    main()

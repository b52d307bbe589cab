#!/usr/bin/env python3
"""Would the comparator tails gain from running BESIDE loop B?  Two contexts on one GPU, two host threads: A streams the database
(computeSimilarity on G blocks: loop A + loop B + one relinearisation per block), B runs the comparator on a batch of score ciphertexts.
Prints A alone, B alone, and both at once: (A || B) close to max(A, B) means the two overlap, close to A + B means they only take turns.
Usage: probe_overlap.py [log2 blocks, default 6] [comparator batch, default = blocks]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_matching_amd as im  # noqa: E402

G = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 6)
X = int(sys.argv[2]) if len(sys.argv) > 2 else G


def make(seed):
    cc = im.Context()
    cc.set_matvec("hoisted")
    cc.fill_eval_keys_random(seed)
    q = np.stack([np.random.default_rng(seed).integers(0, int(m), size=(2, cc.N), dtype=np.uint64) for m in cc.moduli[:cc.nQ]], axis=1)
    return cc, q


a, qa = make(1)
a.db_fill_random(G * 16384, 2)
ga = a.import_ct(qa, a.delta)
snd = im.DiagonalSender(a, G * 16384)
b, qb = make(2)
scores = np.stack([qb[:, :b.nQ - 1]] * X)  # X two-component ciphertexts one level down, like a similarity batch
gb = b.import_ct(scores, b.delta)


def run_a(n):
    for _ in range(n):
        r = snd.computeSimilarity(ga)
    a.sync()


def run_b(n):
    for _ in range(n):
        r = b.chebyshev_compare(gb)
    b.sync()


def timed(fns, n=4):
    for f in fns:
        f(1)
    t0 = time.time()
    th = [threading.Thread(target=f, args=(n,)) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return (time.time() - t0) / n * 1e3


ta, tb = timed([run_a]), timed([run_b])
tab = timed([run_a, run_b])
print("blocks %d, comparator batch %d: A alone %.2f ms, B alone %.2f ms, A || B %.2f ms per pair (sum %.2f, max %.2f)" % (G, X, ta, tb, tab, ta + tb, max(ta, tb)))

#!/usr/bin/env python3
"""Build tests/golden/*.npz from the DATA files the reference ships (run once, here; the GPU box has no
/root/reference).  Nothing of the reference's code is executed or copied: the inputs are its two synthetic datasets
(test/2_10.dat, test/2_11.dat; format: n, 512-value query, n rows of 512 — src/main.cpp:56-57,216-230) and the
published comparator transfer curve tools/figures/signApprox.csv (columns input, combined).  The expected outputs
stored beside them are plain numpy cosine scores (the quantity src/main_accuracy.cpp:359-360 compares decrypted
scores with, tolerance 1e-4) and the decision results `true` / `[0]` that follow from the >= 1.0 rule of
src/receiver/receiver_hers.cpp:30,47 with MATCH_THRESHOLD 0.44 (include/config.h:9).
"""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def read_dat(path):
    tok = open(path).read().split()
    n = int(tok[0])
    vals = np.array(tok[1:], dtype=np.float64)
    dim = len(vals) // (n + 1)
    return n, vals[:dim], vals[dim:].reshape(n, dim)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in ("2_10", "2_11"):
        n, q, db = read_dat(os.path.join(REF, "test", name + ".dat"))
        assert np.abs(db).max() < 128 and np.all(db == np.round(db))
        cos = (db / np.linalg.norm(db, axis=1, keepdims=True)) @ (q / np.linalg.norm(q))
        idx = np.nonzero(cos >= 0.44)[0]
        np.savez_compressed(os.path.join(OUT, "dataset_%s.npz" % name), n=n, query=q.astype(np.int8),
                            db=db.astype(np.int8), cosine=cos, expected_index=idx.astype(np.int64),
                            expected_membership=np.array(len(idx) > 0))
        print(name, n, "matches", idx, "cos[0]", cos[0], "max other", np.sort(cos)[-2])
    rows = np.genfromtxt(os.path.join(REF, "tools", "figures", "signApprox.csv"), delimiter=",", skip_header=1)
    np.savez_compressed(os.path.join(OUT, "sign_approx.npz"), input=rows[:, 0], combined=rows[:, 1])
    print("signApprox", rows.shape)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Write a tests/golden/dataset_*.npz fixture back out in the reference's .dat text format (n, query, n rows) so that
image_matching_amd/ImageMatching can be run on it: tools/npz_to_dat.py tests/golden/dataset_2_10.npz /tmp/2_10.dat"""
import sys

import numpy as np

g = np.load(sys.argv[1])
with open(sys.argv[2], "w") as f:
    f.write("%d\n" % int(g["n"]))
    f.write(" ".join(str(int(v)) for v in g["query"]) + " \n")
    for row in g["db"]:
        f.write(" ".join(str(int(v)) for v in row) + " \n")

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, oracle_lib as O, image_matching_amd as im
P = O.Params(log_n=11, depth=11, dim=64); K = O.Keys(P, 7); Or = O.Oracle(P, K)
cc = im.Context(im.default_params(log_n=11, vector_dim=64), 0); cc.keygen(7)
rng = np.random.default_rng(2)
z = rng.uniform(-1, 1, (3, P.slots)); z[2] = 0.0
g = cc.encrypt(z, 11, 40)
dec = cc.decrypt(g)
for i in range(3):
    o = Or.encrypt(z[i], 11, 40 + i)
    b = Or.decrypt(o)
    single = cc.decrypt(cc.import_ct(o.data(), o.scale))[0]
    for name, a in (("batch", dec[i]), ("single", single)):
        d = np.nonzero(a != b)[0]
        print(i, name, "mismatches", len(d), "max abs diff", np.abs(a - b).max())
        for j in d[:6]:
            print("   idx", j, a[j].hex(), b[j].hex())

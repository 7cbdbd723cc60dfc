import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D = 16384, 8
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
def mk(opts={}):
    h = _lib.Handle(0)
    for k, v in opts.items(): h.set_option(k, v)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
    h.fit(); h.fit()
    return h, [p["ms"] for p in h.phases() if p["name"].startswith("chol")][0]
a, t = mk(); print("A:", t)
b, t = mk(); print("B (A alive):", t)
a.close(); c, t = mk(); print("C (A closed, B alive):", t)
b.close(); c.close()
d, t = mk(); print("D (all closed):", t)
d.close()
e, t = mk({"reserve_cus": 0}); print("E reserve 0 (unmasked bulk stream):", t)
e.close()
f, t = mk({"reserve_cus": 16}); print("F reserve 16:", t)
f.close()
N2, M = 16384, 10000
Xs = rng.uniform(0, 1, (M, D))
g, t = mk(); g.set_candidates(Xs); g.fit_predict(True); g.fit_predict(True)
print("G fused:", [p for p in g.phases() if p["name"].startswith("chol")][0]["ms"]); g.fit(); print("G fit after fused:", [p["ms"] for p in g.phases() if p["name"].startswith("chol")][0])

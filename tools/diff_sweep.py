"""Randomised differential check of the round-4 routes against the default route (test tooling): the pair step of the in-panel
factorisation (inner_tiles = 2, any panel width, both schedulers) and the small-M path (small_m = 8 vs 0), random shapes.  Prints the
largest relative differences seen; exits 1 beyond 1e-9."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
h = _lib.Handle(0)
worst = {"lml": 0.0, "mean": 0.0, "var": 0.0, "grad": 0.0, "dvdx": 0.0}
for case in range(n_cases):
    N = int(rng.choice([1, 2, 100, 127, 128, 129, 255, 256, 257, 383, 384, 500, 640, 768, 769, 1000, 1500, 2047, 2048, 2600, 3100]))
    D = int(rng.choice([1, 2, 5, 8, 13])); M = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 40, 200]))
    kern = int(rng.integers(2)); ard = int(rng.integers(2)); noise = float(rng.choice([1e-1, 1e-2, 1e-3]))
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(-0.05, 1.05, (M, D))
    ls = rng.uniform(0.3, 1.2, D if ard else 1) * np.sqrt(D) * 0.5
    opts = dict(panel_tiles=int(rng.integers(1, 9)), lookahead_min_tiles=int(rng.choice([0, 4, 40])), emulate_fp64=int(rng.integers(2)))
    res = {}
    for route in ("default", "new"):
        for k, v in opts.items(): h.set_option(k, v)
        h.set_option("inner_tiles", 2 if route == "new" else 1)
        h.set_option("small_m", 8 if route == "new" else 0)
        h.set_data(X, Y); h.set_params(kern, ard, 1.3, ls, noise); h.set_candidates(Xs)
        if rng.integers(2) and route == "new":
            (lml, _, _), mu, var = h.fit_predict(True)
        else:
            lml = h.fit()[0]; mu, var = h.predict(True)
        g = h.lml_grad(ls.size)
        dm, dv = h.predict_grad()
        res[route] = (lml, mu, var, np.r_[g[0], g[1], g[2]], dv)
    a, b = res["default"], res["new"]
    rel = lambda x, y: float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / max(np.max(np.abs(y)), 1e-300))
    d = {"lml": abs(a[0] - b[0]) / max(abs(a[0]), 1.0), "mean": rel(b[1], a[1]), "var": rel(b[2], a[2]), "grad": rel(b[3], a[3]), "dvdx": rel(b[4], a[4])}
    for k in worst: worst[k] = max(worst[k], d[k])
    if max(d.values()) > 1e-9:
        print("case", case, dict(N=N, D=D, M=M, kern=kern, ard=ard, noise=noise, **opts), d, flush=True)
print("cases", n_cases, "worst relative differences", {k: "%.1e" % v for k, v in worst.items()}, flush=True)
h.close()
sys.exit(1 if max(worst.values()) > 1e-9 else 0)

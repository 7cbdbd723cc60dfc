"""Wall time of gp_fit at C3 for option settings given as k=v[,v2...] (test tooling)."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D = int(os.environ.get("N", 16384)), 8
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
keys = [a.split("=")[0] for a in sys.argv[1:]]
vals = [[int(v) for v in a.split("=")[1].split(",")] for a in sys.argv[1:]]
ref = None
for combo in itertools.product(*vals):
    for k, v in zip(keys, combo):
        h.set_option(k, v)
    h.fit(); h.fit(); h.synchronize()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        out = h.fit()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = sorted(ts)[len(ts) // 2]
    chol = [p for p in h.phases() if p["name"] == "cholesky"][0]
    if ref is None: ref = out[0]
    print(dict(zip(keys, combo)), "fit median %.2f ms  chol %.2f ms %.1f TF  lml rel diff %.1e" % (ms, chol["ms"], chol["flops"] / chol["ms"] / 1e9, abs(out[0] - ref) / abs(ref)), flush=True)
h.close()

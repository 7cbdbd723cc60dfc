import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D, M = 16384, 8, 10000
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
def chol():
    return [p["ms"] for p in h.phases() if p["name"].startswith("chol")][0]
h.fit(); h.fit(); print("fit only:", chol())
h.set_candidates(Xs); h.fit(); print("after set_candidates:", chol())
h.predict(True); h.fit(); print("after predict:", chol())
h.fit(); print("again:", chol())
h.fit_predict(True); print("fused phase:", chol())
h.fit(); print("fit after fused:", chol())
h.fit(); print("again:", chol())
h.predict(True); h.fit(); print("after predict:", chol())
h.close()
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
def wall(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3
h.fit()
print("wall fit (fresh handle):", [round(wall(h.fit), 2) for _ in range(3)], "dev", sum(p["ms"] for p in h.phases()))
h.set_candidates(Xs); h.predict(True)
print("wall fit after predict:", [round(wall(h.fit), 2) for _ in range(3)], "dev", sum(p["ms"] for p in h.phases()))
print("wall predict:", [round(wall(lambda: h.predict(True)), 2) for _ in range(3)], "dev", sum(p["ms"] for p in h.phases()))
print("wall fused:", [round(wall(lambda: h.fit_predict(True)), 2) for _ in range(3)], "dev", sum(p["ms"] for p in h.phases()))
print("wall fit after fused:", [round(wall(h.fit), 2) for _ in range(3)], "dev", sum(p["ms"] for p in h.phases()))

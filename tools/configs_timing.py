"""Wall time of the BASELINE.json configurations other than the bench line (C2, C4 one shard, C5), for DESIGN.md (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

def data(N, D, M, seed=1234):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    f = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D)
    Y = f + 0.05 * np.random.default_rng(seed + 1).standard_normal((N, 1)); Y = (Y - Y.mean()) / Y.std()
    return X, Y, np.random.default_rng(seed + 2).uniform(0, 1, (M, D))

def tm(fn, n=3):
    fn(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r

h = _lib.Handle(0)
# C2: N=4096, D=4 RBF: K-build + Cholesky
X, Y, Xs = data(4096, 4, 8)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.5], 1e-2)
ms, _ = tm(h.fit, 10); ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
print("C2 fit %.2f ms" % ms, ph, "cholesky %.1f TFLOP/s" % (4096**3 / 3 / ph["cholesky"] / 1e9), flush=True)
# C4, one rank: N=16384, D=8 Matern-5/2, fit once + EI over 125 000 candidates + argbest
X, Y, Xs = data(16384, 8, 125000)
h.set_data(X, Y); h.set_params(1, 0, 1.0, [0.25 * np.sqrt(8)], 1e-2)
msf, _ = tm(h.fit, 3)
h.set_candidates(Xs)
def ei():
    h.predict(True)                     # posterior at the resident shard (8 chunks of mc_max candidates)
    f = h.fmin(); return h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
mse, r = tm(ei, 2)
print("C4 shard: fit %.1f ms, EI + argbest over 125000 candidates %.1f ms (%.1f TFLOP/s in the candidate solve)" % (msf, mse, 16384.0**2 * 125000 / mse / 1e9), r, flush=True)
# C5: N=32768, D=16 ARD-RBF: LML + (D+2) gradients per evaluation
X, Y, Xs = data(32768, 16, 8)
h.set_data(X, Y); h.set_params(0, 1, 1.0, 0.2 + 0.04 * np.arange(16), 1e-2)
def ev():
    l = h.fit(); return l, h.lml_grad(16)
ms, r = tm(ev, 3)
h.fit(); phf = {p["name"]: round(p["ms"], 2) for p in h.phases()}
h.lml_grad(16); phg = {p["name"]: round(p["ms"], 2) for p in h.phases()}
print("C5 evaluation (LML + 18 gradients) %.1f ms" % ms, phf, phg, "cholesky %.1f TFLOP/s, potri %.1f TFLOP/s" % (32768.0**3 / 3 / phf["cholesky"] / 1e9, 2 * 32768.0**3 / 3 / sum(v for k, v in phg.items() if k.startswith("potri")) / 1e9), flush=True)
h.close()

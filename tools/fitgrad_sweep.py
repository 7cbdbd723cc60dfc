"""gp_fit_grad vs gp_fit + gp_lml_grad: wall time and equality, for pipeline settings (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D = int(os.environ.get("N", 16384)), 8
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 1, 1.0, 0.5 + 0.05 * np.arange(D), 1e-2)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    h.set_option(k, int(v))
def sep():
    l = h.fit(); return l, h.lml_grad(D)
def fus():
    return h.fit_grad(D)
def tm(fn, n=4):
    fn(); h.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    return (time.perf_counter() - t0) / n * 1e3, r
ts, rs = tm(sep); tf, rf = tm(fus)
same = rs[0] == rf[0] and rs[1][0] == rf[1][0] and np.array_equal(rs[1][1], rf[1][1]) and rs[1][2] == rf[1][2]
print(sys.argv[1:], "separate %.2f ms  fused %.2f ms  bitwise-equal %s" % (ts, tf, same), {p["name"]: round(p["ms"], 2) for p in h.phases()}, flush=True)
h.close()

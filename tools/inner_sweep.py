"""gp_fit / gp_fit_predict / emulated fit against the in-panel step (inner_tiles, inner_min_rows) at several N (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

def data(N, D, M, seed=1234):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D) + 0.05 * rng.standard_normal((N, 1))
    return X, (Y - Y.mean()) / Y.std(), np.random.default_rng(seed + 2).uniform(0, 1, (M, D))

def med(fn, n):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r

settings = [("inner_tiles", 1, 0)] + [("inner_tiles", 2, m) for m in (0, 8, 16, 24, 32, 48, 64, 96)]
if len(sys.argv) > 1:
    settings = [("inner_tiles", 2, int(a)) if a != "off" else ("inner_tiles", 1, 0) for a in sys.argv[1:]]
h = _lib.Handle(0)
for N, D, M in ((4096, 4, 2000), (8192, 8, 5000), (16384, 8, 10000)):
    X, Y, Xs = data(N, D, M)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    ref = None
    for _, it, mr in settings:
        h.set_option("inner_tiles", it); h.set_option("inner_min_rows", mr)
        out = []
        for emu in (0, 1):
            h.set_option("emulate_fp64", emu)
            tf, r = med(h.fit, 5 if N >= 16384 else 15)
            ts, _ = med(lambda: h.fit_predict(True), 3 if N >= 16384 else 8)
            out.append((tf, ts, r[0]))
        if ref is None: ref = out[0][2]
        print("N=%5d inner_tiles=%d min_rows=%3d | fp64: fit %7.3f ms  fit_predict %7.3f ms | emulated: fit %7.3f ms  fit+predict %7.3f ms | lml rel %.1e"
              % (N, it, mr, out[0][0], out[0][1], out[1][0], out[1][1], abs(out[0][2] - ref) / abs(ref)), flush=True)
h.close()

"""A/B of schedule options on fit and predict at the headline size (test tooling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib
N, D, M = 16384, 8, 10000
rng = np.random.default_rng(1234)
X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
h = _lib.Handle(0)
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
opts = [dict(a.split("=") for a in grp.split(",")) for grp in sys.argv[1:]] or [{}]
for rnd in range(2):
    for o in opts:
        for k, v in o.items(): h.set_option(k, int(v))
        h.fit(); h.predict(True)
        tf, tp = [], []
        for r in range(3):
            t0 = time.perf_counter(); h.fit(); t1 = time.perf_counter(); h.predict(True); t2 = time.perf_counter()
            tf.append(t1-t0); tp.append(t2-t1)
        tfp = []
        for r in range(3):
            t0 = time.perf_counter(); h.fit_predict(True); tfp.append(time.perf_counter() - t0)
        print(o, "fit %.2f ms predict %.2f ms | fused fit_predict %.2f ms" % (min(tf)*1e3, min(tp)*1e3, min(tfp)*1e3))
h.close()

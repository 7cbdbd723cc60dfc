"""gp_fit and the whole bench step at C3: true fp64 vs emulate_fp64 (trailing update + candidate solve on int8 MFMA)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib
import bench
N, D, M = 16384, 8, 10000
X, Y, Xs = bench.synthetic(N, D, M)
h = _lib.Handle(0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); h.set_option(k, int(v))
h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25 * np.sqrt(D)], 1e-2); h.set_candidates(Xs)
ref = None
for emu, efit in ((0, 0), (1, 1), (1, 8), (0, 0), (1, 8)):
    h.set_option("emulate_fp64", emu); h.set_option("emulate_fit", 1 if efit else 0)
    if efit: h.set_option("rns_group_fit", efit)
    h.fit(); h.predict(True)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        lml = h.fit()[0]
    h.synchronize(); tf = (time.perf_counter() - t0) / 4 * 1e3
    phf = {p["name"]: round(p["ms"], 3) for p in h.phases()}
    t0 = time.perf_counter()
    for _ in range(4):
        h.fit(); mu, var = h.predict(True); f = h.fmin(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
    h.synchronize(); ts = (time.perf_counter() - t0) / 4 * 1e3
    h.fit_predict(True)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        (lmlf, _, _), muf, varf = h.fit_predict(True); f = h.fmin(); h.acq_argbest(_lib.GP_ACQ_EI, 0.01, f, -1)
    h.synchronize(); tfu = (time.perf_counter() - t0) / 4 * 1e3
    phu = {p["name"]: round(p["ms"], 3) for p in h.phases()}
    print("   fused gp_fit_predict + EI: %.2f ms = %.2f it/s  same as two calls: %s  phases %s" % (tfu, 1e3 / tfu, bool(lmlf == lml and np.array_equal(muf, mu) and np.array_equal(varf, var)), phu))
    if ref is None: ref = (lml, mu.copy(), var.copy())
    print("emulate_fp64=%d rns_group_fit=%d  fit %.2f ms (cholesky %.2f = %.1f TFLOP/s eq)  step(fit+predict+EI) %.2f ms = %.2f it/s   lml rel diff %.1e  var rel diff %.1e"
          % (emu, efit, tf, phf["cholesky"], N ** 3 / 3.0 / phf["cholesky"] / 1e9, ts, 1e3 / ts, abs(lml - ref[0]) / abs(ref[0]),
             np.max(np.abs(var - ref[2]) / ref[2])))
h.close()

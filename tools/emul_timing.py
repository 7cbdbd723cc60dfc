"""Candidate solve at C3 (N=16384, D=8, M=10^4): true fp64 vs the int8 residue prototype ("emulate_fp64")."""
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
h.fit()
for emu, variant, dbg in ((0, 8, 1), (1, 1, 1), (1, 2, 1), (1, 4, 1), (1, 8, 0), (1, 8, 1), (0, 8, 1), (1, 8, 1)):
    h.set_option("emulate_fp64", emu)
    h.set_option("rns_group", variant); h.set_option("rns_interleave", dbg)
    h.predict(True)
    h.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        mu, var = h.predict(True)
    h.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
    print("emulate_fp64=%d group %d interleave %d  predict %.2f ms  (%.1f TFLOP/s fp64-equivalent)  phases %s" % (emu, variant, dbg, dt, float(N) * N * M / dt / 1e9, ph))
    if emu == 0: ref = (mu.copy(), var.copy())
    else: print("   max |mean diff| %.2e   max rel var diff %.2e" % (np.max(np.abs(mu - ref[0])), np.max(np.abs(var - ref[1]) / ref[1])))
h.close()

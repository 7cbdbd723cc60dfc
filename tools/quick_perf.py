"""Quick timing of the headline configuration (test tooling; uses no oracle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

def main(N=16384, D=8, M=10000, panel=4, reps=3, kernel=0):
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(2*np.pi*X).sum(1, keepdims=True)/np.sqrt(D) + 0.05*rng.standard_normal((N, 1))
    Y = (Y - Y.mean())/Y.std()
    Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0)
    h.set_option("panel_tiles", panel)
    h.set_data(X, Y)
    h.set_params(kernel, 0, 1.0, [0.25*np.sqrt(D)], 1e-2)
    h.set_candidates(Xs)
    for r in range(reps):
        h.profile(True)
        t0 = time.perf_counter()
        lml, logdet, jit = h.fit()
        t1 = time.perf_counter()
        ph_fit = h.phases()
        mu, var = h.predict(True)
        t2 = time.perf_counter()
        ph_pr = h.phases()
        gs = h.gemm_stats()
        print("rep %d N=%d panel=%d: fit %.2f ms predict %.2f ms total %.2f ms -> %.2f it/s  lml=%.6f jit=%g" % (r, N, panel, (t1-t0)*1e3, (t2-t1)*1e3, (t2-t0)*1e3, 1.0/(t2-t0), lml, jit))
        for p in ph_fit + ph_pr:
            extra = ""
            if p["flops"] > 0 and p["ms"] > 0: extra = " %.2f TFLOP/s" % (p["flops"]/p["ms"]/1e9)
            if p["bytes"] > 0 and p["ms"] > 0: extra += " %.1f GB/s" % (p["bytes"]/p["ms"]/1e6)
            print("    %-12s %9.3f ms%s" % (p["name"], p["ms"], extra))
        print("    gemm: %d launches %.2f ms %.2f TFLOP/s" % (gs["launches"], gs["ms"], gs["flops"]/max(gs["ms"],1e-9)/1e9))
    h.close()

if __name__ == "__main__":
    args = [int(a) for a in sys.argv[1:]]
    main(*args)

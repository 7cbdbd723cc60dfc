"""Per-launch GEMM efficiency at the headline size (test tooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_optimization_amd import _lib

def run(N=16384, D=8, M=10000, panel=4):
    rng = np.random.default_rng(1234)
    X = rng.uniform(0, 1, (N, D)); Y = rng.standard_normal((N, 1)); Xs = rng.uniform(0, 1, (M, D))
    h = _lib.Handle(0); h.set_option("panel_tiles", panel)
    h.set_data(X, Y); h.set_params(0, 0, 1.0, [0.25*np.sqrt(D)], 1e-2); h.set_candidates(Xs)
    h.fit(); h.predict(True)
    for what in ("fit", "predict"):
        h.profile(True)
        if what == "fit": h.fit()
        else: h.predict(True)
        tiles, K, ms = h.gemm_trace()
        fl = 2.0*128*128*np.abs(K)*tiles
        print("== %s panel=%d: %d launches, %.2f ms total, %.2f TF overall" % (what, panel, len(ms), ms.sum(), fl.sum()/ms.sum()/1e9))
        for kk in sorted(set(np.abs(K))):
            sel = np.abs(K) == kk
            t = tiles[sel]; m = ms[sel]; f = fl[sel]
            print("   K=%5d: %4d launches, %8.2f ms, %6.2f TF ; tiles min/med/max %d/%d/%d" % (kk, sel.sum(), m.sum(), f.sum()/m.sum()/1e9, t.min(), np.median(t), t.max()))
            big = t >= 2048
            if big.any(): print("       launches with >=2048 tiles: %d, %.2f ms, %.2f TF" % (big.sum(), m[big].sum(), f[big].sum()/m[big].sum()/1e9))
            small = t < 512
            if small.any(): print("       launches with <512 tiles: %d, %.2f ms, %.2f TF, avg %.1f us" % (small.sum(), m[small].sum(), f[small].sum()/m[small].sum()/1e9, 1e3*m[small].mean()))
        if what == "predict":
            for t_, k_, m_ in zip(tiles, K, ms):
                print("      tiles %6d K %6d  %8.3f ms  %6.2f TF" % (t_, k_, m_, 2.0*128*128*abs(k_)*t_*(0.5 if k_ < 0 else 1.0)/m_/1e9))
        h.profile(False)
    h.close()

if __name__ == "__main__":
    run(panel=8)

"""Random sweep: emulate_fp64 against the true-fp64 device path on problems far from the bench line (test tooling).

Kernel, ARD, D, N, panel width, variance over six decades, lengthscales over 2.5 decades, noise from 1e-8 to 1 -- for each
draw: gp_fit + gp_predict both ways, differences of LML / mean / variance, whether the jitter ladder took the same rung,
and how often the residue path had to fall back (non-finite data are not drawn here, so that count must stay 0)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_optimization_amd import _lib

n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(20260)
h = _lib.Handle(0)
rows, fails = [], []
for it in range(n_draws):
    N = int(rng.integers(130, 3000)); D = int(rng.integers(1, 11)); M = int(rng.integers(1, 400))
    kernel = int(rng.integers(0, 2)); ard = int(rng.integers(0, 2)); pt = int(rng.integers(1, 9))
    variance = float(10 ** rng.uniform(-3, 3)); noise = float(variance * 10 ** rng.uniform(-8, 0))
    ls = (10 ** rng.uniform(-1.5, 1.0, D if ard else 1)).tolist()
    X = rng.uniform(0, 1, (N, D)); Y = np.sin(3 * X.sum(1, keepdims=True)) * np.sqrt(variance) + np.sqrt(noise) * rng.standard_normal((N, 1))
    Xs = rng.uniform(-0.1, 1.1, (M, D))
    h.set_option("panel_tiles", pt)
    out = []
    try:
        for emu in (0, 1):
            h.set_option("emulate_fp64", emu)
            h.set_data(X, Y); h.set_params(kernel, ard, variance, ls, noise); h.set_candidates(Xs)
            f = h.fit(); mu, var = h.predict(True)
            out.append((f, mu.copy(), var.copy()))
    except Exception as e:  # noqa: BLE001
        fails.append(dict(draw=it, N=N, D=D, kernel=kernel, ard=ard, pt=pt, variance=variance, noise=noise, emu=emu, error=str(e)[:200]))
        continue
    (f0, m0, v0), (f1, m1, v1) = out
    rows.append(dict(draw=it, N=N, D=D, M=M, kernel=kernel, ard=ard, pt=pt, variance=variance, noise_over_variance=noise / variance,
                     jitter_fp64=f0[2], jitter_emulated=f1[2],
                     lml_rel=abs(f1[0] - f0[0]) / max(abs(f0[0]), 1e-300),
                     mean_rel=float(np.max(np.abs(m1 - m0)) / max(np.max(np.abs(m0)), 1e-300)),
                     var_rel_to_prior=float(np.max(np.abs(v1 - v0)) / (variance + noise))))
h.set_option("emulate_fp64", 0)
h.close()
def pct(k):
    a = np.array([r[k] for r in rows]); return dict(median=float(np.median(a)), p90=float(np.percentile(a, 90)), max=float(a.max()))
summary = dict(draws=n_draws, completed=len(rows), errors=fails, same_jitter_rung=int(sum(r["jitter_fp64"] == r["jitter_emulated"] for r in rows)),
               lml_rel=pct("lml_rel"), mean_rel=pct("mean_rel"), var_rel_to_prior=pct("var_rel_to_prior"),
               worst=sorted(rows, key=lambda r: -max(r["lml_rel"], r["mean_rel"], r["var_rel_to_prior"]))[:5])
print(json.dumps(summary, indent=1))
json.dump(dict(summary=summary, rows=rows), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "emul_sweep.json"), "w"))

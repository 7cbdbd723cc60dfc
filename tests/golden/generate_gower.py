"""Generate tests/golden/gp_gower.npz (BUILD CONTAINER ONLY: needs /root/reference).

The configuration the fork exists for (run.py:166-185,1206-1258): a mixed design space of four discrete and two
continuous variables, the Gower product kernel for K (stationary.py:116-135) with the fork's Euclidean gradient formulas
around it (stationary.py:336-364 inside gp.py:407-454), EI / LCB / MPI, ``estimate_L`` and the candidate-table batch loop
with local penalisation.

As in generate_golden.py every LAPACK-level step, ``get_quantiles`` and ``normalize`` go through the reference's verbatim
modules; here the design space is the reference's verbatim ``GPyOpt.core.task.space.Design_space`` too (its import chain
-- variables.py, errors.py, util/general.py -- is all reference files).  The kernel, inference glue and acquisition
formulas come from oracle/cpu_ref.py, whose acquisition layer oracle/pin_against_reference.py pins bit-identical to the
verbatim GPyOpt classes.  Run:  python tests/golden/generate_gower.py
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import cpu_ref as O  # noqa: E402
from oracle import ref_leaf  # noqa: E402
from generate_golden import use_reference_modules  # noqa: E402

# run.py:177-185 with made-up catalogue sizes and ranges
DOMAIN = [{'name': 'motor', 'type': 'discrete', 'domain': tuple(range(6))},
          {'name': 'propeller', 'type': 'discrete', 'domain': tuple(range(9))},
          {'name': 'quad_battery', 'type': 'discrete', 'domain': tuple(range(4))},
          {'name': 'plane_battery', 'type': 'discrete', 'domain': tuple(range(3))},
          {'name': 'distanceFromCenterline', 'type': 'continuous', 'domain': (12.0, 48.0)},
          {'name': 'beam_length', 'type': 'continuous', 'domain': (25.4, 100.0)}]
# (N, M, seed, kernel variance).  Under Gower the diagonal of K is variance^6 while Kdiag stays variance
# (stationary.py:131-133 against :195-198), so variance > 1 drives the predictive variance negative and GPModel's clip to
# 1e-10 (gpmodel.py:99) takes over nearly everywhere: the third case keeps that regime (ties, EI = 0, 0 * inf gradients).
CASES = [(64, 48, 0, 1.0), (300, 120, 1, 0.8), (150, 60, 2, 1.25), (100, 40, 3, 0.9)]
ARD_SEEDS = (3,)          # the last case carries one Euclidean lengthscale per variable (ARD=True): under Gower they enter the gradient
                          # formulas only (stationary.py:188-191,336-364), K ignores them
KERNELS = ["Mat52", "rbf"]                    # GPyOpt's default kernel first (gpmodel.py:58)
NOISES = [1e-6, 1e-2]                         # exact_feval=True (gpmodel.py:72-73) first
BATCH, TABLE = 5, 2000


def objective(X):
    """A smooth-in-the-continuous, level-dependent response (inputs only; nothing of the reference's)."""
    lvl = 0.35 * np.cos(1.3 * X[:, 0]) + 0.2 * np.sin(0.9 * X[:, 1]) - 0.15 * (X[:, 2] == 2) + 0.1 * X[:, 3]
    return (lvl + np.sin(X[:, 4] / 7.0) + 0.6 * np.cos(X[:, 5] / 17.0) + 1e-4 * (X[:, 4] - 30.0) ** 2)[:, None]


def make_case(space, mixed, N, M, seed, variance, kname, noise):
    rng = np.random.default_rng(100 + seed)
    X, Xs = mixed.draw(rng, N), mixed.draw(rng, M)
    Xs[:4] = X[:4]                                    # candidates ON training rows: every factor at r = 0
    Xs[4:8, :4] = X[4:8, :4]                          # same levels, other continuous values
    Y = O.normalize(objective(X) + 0.02 * rng.standard_normal((N, 1)))   # bo.py:236-254, normalize_Y=True
    ard = seed in ARD_SEEDS
    ls = (0.6 + 0.35 * np.arange(6)) if ard else np.array([1.0 + 0.5 * seed])   # the kernel's own (Euclidean) lengthscale parameter(s)
    kern = O.make_kernel(kname, 6, variance, ls, ARD=ard, Gower=True, space=space)
    gp = O.OracleGP(X, Y, kern, noise)
    gm = O.OracleGPModel(gp)
    post = gp.posterior
    rows = np.linspace(0, N - 1, 8).astype(int)
    mu, var = gp.predict(Xs)
    dmdx, dvdx = gp.predictive_gradients(Xs)
    fmin = gm.get_fmin()
    ei, dei = O.acq_EI_withGradients(gm, Xs, 0.01, fmin)
    lcb, dlcb = O.acq_LCB_withGradients(gm, Xs, 2.0)
    mpi, dmpi = O.acq_MPI_withGradients(gm, Xs, 0.01, fmin)
    out = dict(X=X, Y=Y, Xs=Xs, variance=variance, lengthscale=ls, ard=int(ard), noise=noise, kernel=0 if kname == "rbf" else 1,
               rows=rows, K_rows=post["K"][rows], Kx_rows=kern.K(Xs, X)[:8], lml=post["lml"], logdet=post["logdet"],
               alpha=post["alpha"], mu=mu, var=var, dmdx=dmdx, dvdx=dvdx, fmin=fmin,
               neg_EI=-ei, neg_dEI=-dei, neg_LCB=-lcb, neg_dLCB=-dlcb, neg_MPI=-mpi, neg_dMPI=-dmpi)
    # estimate_L (batch_local_penalization.py:52-70) under a fixed numpy seed, and the table loop of run.py:1234-1258
    table = mixed.draw(np.random.default_rng(200 + seed), TABLE)
    out["table"] = table
    for base in ("EI", "LCB", "MPI"):
        lp = O.OracleLP(gm, space, base)
        np.random.seed(1000 + seed)
        rows_b, L, Min = O.lp_table_batch(lp, table, BATCH)
        out["lp_rows_" + base] = np.asarray(rows_b)
        out["lp_r_" + base], out["lp_s_" + base] = lp.r_x0, lp.s_x0          # balls of the first BATCH - 1 rows
        out["lp_final_" + base] = lp.acquisition_function(table)               # the last penalised score vector
        f, df = lp.acquisition_function_withGradients(Xs[8:24])
        out["lp_val_" + base], out["lp_grad_" + base] = f, df
    out["L"], out["Min"], out["np_seed"] = L, Min, 1000 + seed
    # what estimate_L starts its polish from (batch_local_penalization.py:60-64): the steepest of 500 draws + the inputs.
    # The polish itself differentiates by forward differences of step 1e-8 (scipy's default for L-BFGS-B without a
    # jacobian), so its end point moves by 1e-4..1e-1 relative under 1e-10..1e-9 relative changes of the gradients
    # (profiles/r05_estimate_L_sensitivity.txt); the start is the part of L that two float64 paths can agree on.
    np.random.seed(1000 + seed)
    pool = np.vstack([O.samples_multidimensional_uniform(mixed.get_bounds(), 500), X])
    slope = np.sqrt((gp.predictive_gradients(pool)[0][:, :, 0] ** 2).sum(1))
    out["L_start"], out["L_start_row"] = slope.max(), int(np.argmax(slope))
    return out


def main():
    if not ref_leaf.available():
        raise SystemExit("reference tree absent")
    use_reference_modules()
    ref_leaf.load_acquisitions()
    space_mod = importlib.import_module("GPyOpt.core.task.space")
    space = space_mod.Design_space(DOMAIN)            # the reference's own class: lengthscales(), get_*_dims()
    mixed = O.MixedSpace(DOMAIN)
    assert space.lengthscales() == mixed.lengthscales() and space.get_bounds() == mixed.get_bounds()
    assert space.get_continuous_dims() == mixed.get_continuous_dims()
    assert space.get_discrete_dims() == mixed.get_discrete_dims()
    allcases = {}
    n = 0
    for N, M, seed, variance in CASES:
        for kname in KERNELS:
            for noise in NOISES:
                tag = "G_N%d_M%d_s%d_%s_n%g" % (N, M, seed, kname, noise)
                for k, v in make_case(space, mixed, N, M, seed, variance, kname, noise).items():
                    allcases[tag + "/" + k] = np.asarray(v)
                n += 1
    allcases["domain_json"] = np.asarray(json.dumps(DOMAIN))
    path = os.path.join(HERE, "gp_gower.npz")
    np.savez_compressed(path, **allcases)
    print("wrote %s: %d cases, %.2f MB" % (path, n, os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()

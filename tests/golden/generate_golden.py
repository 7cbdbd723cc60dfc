"""Generate tests/golden/*.npz (BUILD CONTAINER ONLY: needs /root/reference).

Every linear-algebra step below goes through the reference's own verbatim
modules -- GPy.util.linalg (tdot, pdinv/jitchol, dpotrs, dtrtrs, dpotri),
GPy.util.diag.add, GPyOpt.util.general (get_quantiles, normalize) -- by
substituting them for the oracle's restatements while the vectors are produced
(oracle/pin_against_reference.py shows the two are bit-identical anyway).  The
kernel functions, inference glue and acquisition formulas, whose reference
modules need the absent ``paramz``, come from oracle/cpu_ref.py (line-by-line
restatements with file:line citations).

Cases: seeds x (N, D, M) x {RBF, Matern52} x {iso, ARD} x noise
{1e-2, 1e-6}.  To keep the fixtures small (< 3 MB) K and L are stored as 8
sampled rows plus the diagonal of L; everything else is stored in full.
Run:  python tests/golden/generate_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cpu_ref as O  # noqa: E402
from oracle import ref_leaf  # noqa: E402

SIZES = [((64, 2, 32), (0, 1, 2)), ((512, 2, 256), (0, 1, 2)), ((512, 8, 256), (0, 1, 2)), ((300, 5, 77), (3,))]
KERNELS = [("rbf", False), ("rbf", True), ("Mat52", False), ("Mat52", True)]
NOISES = [1e-2, 1e-6]


_ORIG_PDINV = O.pdinv


def use_reference_modules():
    linalg, diag, normalizer, general = ref_leaf.load()
    O.tdot = linalg.tdot
    O.symmetrify = linalg.symmetrify
    O.dpotrs = linalg.dpotrs
    O.dtrtrs = linalg.dtrtrs
    O.dpotri = linalg.dpotri
    O.dtrtri = linalg.dtrtri
    O.diag_add = diag.add
    O.get_quantiles = general.get_quantiles
    O.normalize = general.normalize

    def pdinv(A, maxtries=5, with_Li=True):
        Ai, L, Li, logdet = linalg.pdinv(A, maxtries)
        return Ai, L, Li, logdet, 0.0
    O.pdinv = pdinv


def make_case(N, D, M, seed, kname, ard, noise):
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=seed)
    ls = O.default_lengthscale(D, ard)
    variance = 1.0 + 0.1 * seed
    kern = O.make_kernel(kname, D, variance, ls, ARD=ard)
    gp = O.OracleGP(X, Y, kern, noise)
    post = gp.posterior
    gm = O.OracleGPModel(gp)
    rows = np.linspace(0, N - 1, 8).astype(int)
    dvar, dlen, dnoise = gp.gradients()
    mu, var = gp.predict(Xs)
    mu0, var0 = gp.predict_noiseless(Xs)
    dmdx, dvdx = gp.predictive_gradients(Xs)
    fmin = gm.get_fmin()
    ei, dei = O.acq_EI_withGradients(gm, Xs, 0.01, fmin)
    lcb, dlcb = O.acq_LCB_withGradients(gm, Xs, 2.0)
    mpi, dmpi = O.acq_MPI_withGradients(gm, Xs, 0.01, fmin)
    out = dict(
        X=X, Y=Y, Xs=Xs, variance=variance, lengthscale=ls, noise=noise, ard=int(ard),
        kernel=0 if kname == "rbf" else 1, rows=rows,
        K_rows=post["K"][rows], L_rows=np.tril(post["L"])[rows], L_diag=np.diag(post["L"]).copy(),
        logdet=post["logdet"], alpha=post["alpha"], lml=post["lml"],
        dvariance=dvar, dlengthscale=dlen, dnoise=dnoise,
        Wi_rows=post["Wi"][rows],
        mu=mu, var=var, var_noiseless=var0, dmdx=dmdx, dvdx=dvdx, fmin=fmin,
        neg_EI=-ei, neg_dEI=-dei, neg_LCB=-lcb, neg_dLCB=-dlcb, neg_MPI=-mpi, neg_dMPI=-dmpi,
        argmin_EI=int(np.argmin(-ei)), argmin_LCB=int(np.argmin(-lcb)), argmin_MPI=int(np.argmin(-mpi)),
    )
    if M <= 40:
        out["cov_full"] = gp.predict(Xs, full_cov=True)[1]
    return out


def main():
    if not ref_leaf.available():
        raise SystemExit("reference tree absent")
    use_reference_modules()
    allcases = {}
    n = 0
    for (N, D, M), seeds in SIZES:
        for seed in seeds:
            for kname, ard in KERNELS:
                for noise in NOISES:
                    tag = "N%d_D%d_M%d_s%d_%s_%s_n%g" % (N, D, M, seed, kname, "ard" if ard else "iso", noise)
                    c = make_case(N, D, M, seed, kname, ard, noise)
                    for k, v in c.items():
                        allcases[tag + "/" + k] = np.asarray(v)
                    n += 1
    # multi-output + normaliser case (P = 3)
    rng = np.random.default_rng(11)
    X = rng.uniform(0, 1, (200, 3)); Xs = rng.uniform(0, 1, (40, 3))
    Y = np.c_[np.sin(X.sum(1)), np.cos(2 * X[:, 0]), X[:, 1] ** 2] * 3 + 5 + 0.05 * rng.standard_normal((200, 3))
    kern = O.RBF(3, 0.9, [0.3, 0.5, 0.7], ARD=True)
    gp = O.OracleGP(X, Y, kern, 0.02, normalizer=True)
    mu, var = gp.predict(Xs)
    dv, dl, dn = gp.gradients()
    allcases.update({"multi/X": X, "multi/Y": Y, "multi/Xs": Xs, "multi/mu": mu, "multi/var": var,
                     "multi/lml": np.asarray(gp.log_likelihood()), "multi/alpha": gp.posterior["alpha"],
                     "multi/dvariance": np.asarray(dv), "multi/dlengthscale": dl, "multi/dnoise": np.asarray(dn)})
    # jitter ladder cases (the oracle's jitchol -- pinned bit-identical to the reference's -- reports the jitter)
    O.pdinv = _ORIG_PDINV
    # duplicated inputs and a negative "noise" make Ky indefinite by a known margin
    Xd = np.repeat(np.random.default_rng(5).uniform(0, 1, (8, 2)), 8, axis=0)
    Yd = np.random.default_rng(6).standard_normal((64, 1))
    kd = O.RBF(2, 1.0, 0.5)
    for name, noise in (("jit1", -1e-8 - 1e-7), ("jit3", -1e-8 - 3e-5)):
        p = O.exact_gaussian_inference(kd, Xd, Yd, noise)
        allcases.update({name + "/X": Xd, name + "/Y": Yd, name + "/noise": np.asarray(noise),
                         name + "/jitter": np.asarray(p["jitter"]), name + "/lml": np.asarray(p["lml"]),
                         name + "/logdet": np.asarray(p["logdet"])})
    try:
        O.exact_gaussian_inference(kd, Xd, Yd, -1e-8 - 0.5)
        raise SystemExit("expected LinAlgError")
    except np.linalg.LinAlgError:
        allcases.update({"jitfail/X": Xd, "jitfail/Y": Yd, "jitfail/noise": np.asarray(-1e-8 - 0.5)})
    path = os.path.join(HERE, "gp_golden.npz")
    np.savez_compressed(path, **allcases)
    print("wrote %s: %d cases, %.2f MB" % (path, n, os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()

"""Extended-precision truth for the stress goldens (noise 1e-6) -> tests/golden/gp_truth.npz.

Why: at noise 1e-6 cond(Ky) is 1e8..1e9 and every float64 implementation -- LAPACK behind the reference as much as the
HIP path -- carries cond * eps ~ 1e-7 of error in solve-dependent quantities.  Comparing the two float64 results with
each other at 1e-6 would test luck; comparing both with the mathematically exact value of the same expressions shows
which tolerance is evidence.  This script evaluates the reference's formulas (same file:line as oracle/cpu_ref.py) on
the SAME float64 inputs as the golden cases, in numpy.longdouble (x87 80-bit, eps 1.1e-19: 11 more bits than float64,
so its own cond * eps error is ~1e-10) with direct pairwise differences for the distances, an unblocked Cholesky and
substitutions written out below (no LAPACK), and mpmath (50 digits) for the normal cdf.  The stored values are the
truth rounded to float64.

Independent of /root/reference and of oracle/: only the inputs (X, Y, Xs, hyper-parameters) are read from
gp_golden.npz.  Run:  python tests/golden/generate_truth.py      (a few minutes)
"""
import os

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LD = np.longdouble
mp.mp.dps = 50


def ld(a):
    return np.asarray(a, dtype=LD)


def chol_lower(A):
    """Unblocked left-looking Cholesky in the array's precision."""
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        v = A[j:, j] - L[j:, :j] @ L[j, :j]
        L[j, j] = np.sqrt(v[0])
        L[j + 1:, j] = v[1:] / L[j, j]
    return L


def solve_lower(L, B):
    X = np.array(B, dtype=L.dtype, copy=True)
    for i in range(L.shape[0]):
        X[i] = (X[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def solve_upper_from_lower_T(L, B):
    """Solve L^T X = B."""
    X = np.array(B, dtype=L.dtype, copy=True)
    for i in range(L.shape[0] - 1, -1, -1):
        X[i] = (X[i] - L[i + 1:, i] @ X[i + 1:]) / L[i, i]
    return X


def scaled_dist(X, X2, ls):
    """r_ij = |(x_i - x'_j) / l| by direct differences (the quantity stationary.py:155-193 approximates)."""
    d = (X[:, None, :] - X2[None, :, :]) / ls[None, None, :]
    return np.sqrt(np.sum(d * d, axis=-1))


def k_of_r(kernel, variance, r):
    if kernel == 0:   # rbf.py:50-51
        return variance * np.exp(-LD(0.5) * r * r)
    s5 = np.sqrt(LD(5))   # stationary.py:575-576
    return variance * (1 + s5 * r + LD(5) / 3 * r * r) * np.exp(-s5 * r)


def dk_dr(kernel, variance, r):
    if kernel == 0:   # rbf.py:53-54
        return -r * k_of_r(kernel, variance, r)
    s5 = np.sqrt(LD(5))   # stationary.py:578-579
    return variance * (LD(10) / 3 * r - 5 * r - 5 * s5 / 3 * r * r) * np.exp(-s5 * r)


def _mpf(x):
    hi = float(x)
    return mp.mpf(hi) + mp.mpf(float(x - LD(hi)))


def _from_mpf(v):
    hi = float(v)
    return LD(hi) + LD(float(v - mp.mpf(hi)))


def norm_cdf(u):
    out = np.empty(u.shape, dtype=LD)
    for i, x in np.ndenumerate(u):
        out[i] = _from_mpf(mp.erfc(-_mpf(x) / mp.sqrt(2)) / 2)
    return out


def truth_case(X, Y, Xs, kernel, ard, variance, lengthscale, noise, rows):
    X, Y, Xs = ld(X), ld(Y), ld(Xs)
    N, D = X.shape
    variance = LD(float(variance))
    ls = ld(lengthscale) if ard else np.full(D, LD(float(lengthscale[0])), dtype=LD)
    diag = LD(float(noise)) + LD(1e-8)           # exact_gaussian_inference.py:55-56
    r = scaled_dist(X, X, ls)
    K = k_of_r(kernel, variance, r)
    Ky = K + diag * np.eye(N, dtype=LD)
    L = chol_lower(Ky)
    logdet = 2 * np.sum(np.log(np.diag(L)))      # linalg.py:208
    alpha = solve_upper_from_lower_T(L, solve_lower(L, Y))
    lml = LD(0.5) * (-N * np.log(2 * LD(np.pi)) - logdet - np.sum(alpha * Y))   # :62
    Wi = solve_upper_from_lower_T(L, solve_lower(L, np.eye(N, dtype=LD)))
    Wi = (Wi + Wi.T) / 2
    dL_dK = LD(0.5) * (alpha @ alpha.T - Wi)     # :70
    dnoise = np.trace(dL_dK)                     # gaussian.py:78-79
    dvariance = np.sum(K * dL_dK) / variance     # stationary.py:224
    dL_dr = dk_dr(kernel, variance, r) * dL_dK
    if ard:                                      # stationary.py:228-235,260-261
        inv = np.where(r != 0, 1 / np.where(r != 0, r, 1), 0)
        tmp = dL_dr * inv
        dlen = np.array([-np.sum(tmp * (X[:, q:q + 1] - X[:, q:q + 1].T) ** 2) / ls[q] ** 3 for q in range(D)])
    else:                                        # stationary.py:237-238
        dlen = np.array([-np.sum(dL_dr * r) / ls[0]])
    # posterior (posterior.py:273-302), likelihood noise (gaussian.py:109)
    rs = scaled_dist(X, Xs, ls)
    Kx = k_of_r(kernel, variance, rs)
    mu = Kx.T @ alpha
    tmp = solve_lower(L, Kx)
    var0 = (variance - np.sum(tmp * tmp, axis=0))[:, None]
    var = var0 + LD(float(noise))
    # predictive gradients (gp.py:407-454, stationary.py:336-352)
    M = Xs.shape[0]
    inv_s = np.where(rs != 0, 1 / np.where(rs != 0, rs, 1), 0)       # [N, M]
    g = dk_dr(kernel, variance, rs) * inv_s
    dX = (Xs[None, :, :] - X[:, None, :]) / (ls ** 2)[None, None, :]   # [N, M, D]
    dmdx = np.einsum("nm,nmd->md", g * alpha, dX)[:, :, None]
    A = -2 * (Wi @ Kx)                                                 # [N, M]: -2 K(Xs,X) Wi transposed
    dvdx = np.einsum("nm,nmd->md", g * A, dX)
    # GPModel (gpmodel.py:95-142) and acquisitions (general.py:113-129, EI.py, LCB.py, MPI.py), negated (base.py:33-50)
    mu_train = K @ alpha
    fmin = mu_train.min()
    v = np.clip(var, LD(1e-10), None)
    s = np.sqrt(v)
    dsdx = dvdx / (2 * s)
    dm = dmdx[:, :, 0]
    jitter, w = LD(0.01), LD(2.0)
    s_q = np.where(s < LD(1e-10), LD(1e-10), s)
    u = (fmin - mu - jitter) / s_q
    phi = np.exp(-LD(0.5) * u * u) / np.sqrt(2 * LD(np.pi))
    Phi = norm_cdf(u)
    ei, dei = s_q * (u * Phi + phi), dsdx * phi - Phi * dm
    lcb, dlcb = -mu + w * s, -dm + w * dsdx
    mpi, dmpi = Phi, -(phi / s_q) * (dm + dsdx * u)
    f64 = lambda a: np.asarray(a, dtype=np.float64)   # noqa: E731
    return dict(lml=f64(lml), logdet=f64(logdet), alpha=f64(alpha), L_diag=f64(np.diag(L)), mu=f64(mu), var=f64(var),
                var_noiseless=f64(var0), dvariance=f64(dvariance), dlengthscale=f64(dlen), dnoise=f64(dnoise),
                Wi_rows=f64(Wi[rows]), Wi_absmax=f64(np.max(np.abs(Wi))), dmdx=f64(dmdx), dvdx=f64(dvdx), fmin=f64(fmin), neg_EI=f64(-ei), neg_dEI=f64(-dei),
                neg_LCB=f64(-lcb), neg_dLCB=f64(-dlcb), neg_MPI=f64(-mpi), neg_dMPI=f64(-dmpi))


def mp_check(X, Y, kernel, ard, variance, lengthscale, noise, t):
    """The N = 64 cases once more with 50-digit mpmath arithmetic throughout (Cholesky, solves): how far the longdouble
    truth itself is from exact.  Returns the largest relative deviation over lml and alpha."""
    N, D = X.shape
    ls = [mp.mpf(float(v)) for v in (lengthscale if ard else [lengthscale[0]] * D)]
    var = mp.mpf(float(variance))
    Xm = [[mp.mpf(float(v)) for v in row] for row in X]
    Ky = mp.matrix(N, N)
    s5 = mp.sqrt(5)
    for i in range(N):
        for j in range(i + 1):
            r2 = sum(((Xm[i][d] - Xm[j][d]) / ls[d]) ** 2 for d in range(D))
            if kernel == 0:
                k = var * mp.exp(-r2 / 2)
            else:
                r = mp.sqrt(r2)
                k = var * (1 + s5 * r + mp.mpf(5) / 3 * r2) * mp.exp(-s5 * r)
            Ky[i, j] = Ky[j, i] = k
        Ky[i, i] += mp.mpf(float(noise)) + mp.mpf(1e-8)
    L = mp.cholesky(Ky)
    y = mp.matrix([mp.mpf(float(v)) for v in Y[:, 0]])
    alpha = mp.lu_solve(Ky, y)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(N))
    lml = (-N * mp.log(2 * mp.pi) - logdet - sum(alpha[i] * y[i] for i in range(N))) / 2
    dev = abs((mp.mpf(float(t["lml"])) - lml) / lml)
    amax = max(abs(alpha[i]) for i in range(N))
    dev_a = max(abs(mp.mpf(float(t["alpha"][i, 0])) - alpha[i]) for i in range(N)) / amax
    return float(max(dev, dev_a))


def stress_tags(g):
    tags = sorted({k.split("/")[0] for k in g.files if k.startswith("N")})
    return [t for t in tags if t.endswith("_n1e-06")]   # every stress case of the golden set


def main():
    g = np.load(os.path.join(HERE, "gp_golden.npz"))
    out = {}
    for tag in stress_tags(g):
        c = lambda k: g[tag + "/" + k]   # noqa: E731
        t = truth_case(c("X"), c("Y"), c("Xs"), int(c("kernel")), int(c("ard")), float(c("variance")), c("lengthscale"),
                       float(c("noise")), c("rows"))
        for k, v in t.items():
            out[tag + "/" + k] = v
        note = ""
        if tag.startswith("N64_"):
            mp.mp.dps = 60
            dev = mp_check(c("X"), c("Y"), int(c("kernel")), int(c("ard")), float(c("variance")), c("lengthscale"),
                           float(c("noise")), t)
            mp.mp.dps = 50
            out[tag + "/truth_selfcheck"] = np.asarray(dev)
            assert dev < 1e-12, (tag, dev)
            note = "; longdouble vs 60-digit mpmath: %.1e" % dev
        print(tag, "lml %.12f (golden %.12f)%s" % (float(t["lml"]), float(c("lml")), note))
    path = os.path.join(HERE, "gp_truth.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases, %.2f MB" % (path, len(stress_tags(g)), os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()

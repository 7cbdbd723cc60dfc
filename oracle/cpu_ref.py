"""NumPy/SciPy restatement of the reference's exact-GP hot path (TEST ORACLE).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Same LAPACK/BLAS entry points (scipy.linalg.lapack
dpotrf/dpotrs/dpotri/dtrtrs/dtrtri, blas.dsyrk), same elementwise sequence,
same constants (1e-8 diagonal, jitter mean(diag)*1e-6*10^k, clips 1e-10).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  See oracle/__init__.py for the pinning statement.
"""
import numpy as np
from scipy import linalg as _sla
from scipy.linalg import lapack, blas
from scipy.special import erfc

LOG_2_PI = np.log(2 * np.pi)  # exact_gaussian_inference.py:9


# ----------------------------------------------------------------------------
# GPy/GPy/util/linalg.py, GPy/GPy/util/diag.py
# ----------------------------------------------------------------------------
def symmetrify(A, upper=False):
    """In-place copy of one triangle onto the other.

    GPy/GPy/util/linalg.py:356-379 (``_symmetrify_numpy``: ``A[triu] = A.T[triu]``
    or, with ``upper``, ``A.T[triu] = A[triu]``).  Done blockwise here so that
    N=16384 does not allocate 2x N^2/2 int64 index arrays; values identical.
    """
    n = A.shape[0]
    bs = 2048
    for i0 in range(0, n, bs):
        i1 = min(n, i0 + bs)
        if upper:  # lower <- upper
            if i0 > 0:
                A[i0:i1, :i0] = A[:i0, i0:i1].T
            blk = A[i0:i1, i0:i1]
            iu = np.triu_indices(i1 - i0, 1)
            blk.T[iu] = blk[iu]
        else:  # upper <- lower
            if i0 > 0:
                A[:i0, i0:i1] = A[i0:i1, :i0].T
            blk = A[i0:i1, i0:i1]
            iu = np.triu_indices(i1 - i0, 1)
            blk[iu] = blk.T[iu]
    return A


def tdot(mat):
    """``mat @ mat.T`` via BLAS dsyrk + symmetrify.  linalg.py:299-320."""
    mat = np.asarray(mat, dtype=np.float64)
    if mat.ndim != 2:
        return np.dot(mat, mat.T)
    nn = mat.shape[0]
    out = np.zeros((nn, nn))
    matf = np.asfortranarray(mat)
    out = blas.dsyrk(alpha=1.0, a=matf, beta=0.0, c=out, overwrite_c=1, trans=0, lower=0)
    symmetrify(out, upper=True)
    return np.ascontiguousarray(out)


def diag_view(A):
    """Writable view of the diagonal.  GPy/GPy/util/diag.py:6-40."""
    from numpy.lib.stride_tricks import as_strided
    assert A.ndim == 2 and A.shape[0] == A.shape[1]
    return as_strided(A, shape=(A.shape[0],), strides=((A.shape[0] + 1) * A.itemsize,))


def diag_add(A, b):
    """``diag(A) += b`` in place.  GPy/GPy/util/diag.py:85-98."""
    d = diag_view(A)
    d += b
    return A


def jitchol(A, maxtries=5):
    """Cholesky with the reference's jitter ladder.  linalg.py:56-81.

    First attempt is dpotrf on un-jittered A; on info != 0 require diag > 0,
    then up to ``maxtries`` attempts of cholesky(A + jitter*I) with
    jitter = mean(diag)*1e-6 multiplied by 10 after each failure.
    Returns (L, jitter_used).
    """
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L, 0.0
    diagA = np.diag(A)
    if np.any(diagA <= 0.0):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 1
    while num_tries <= maxtries and np.isfinite(jitter):
        try:
            L = _sla.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
            return L, jitter
        except Exception:
            jitter *= 10
        finally:
            num_tries += 1
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


def dtrtrs(A, B, lower=1, trans=0, unitdiag=0):
    """linalg.py:95-114."""
    A = np.asfortranarray(A)
    return lapack.dtrtrs(A, B, lower=lower, trans=trans, unitdiag=unitdiag)


def dpotrs(A, B, lower=1):
    """linalg.py:116-125."""
    A = np.asfortranarray(A)
    return lapack.dpotrs(A, B, lower=lower)


def dpotri(A, lower=1):
    """linalg.py:127-145 (dpotri then symmetrify)."""
    A = np.asfortranarray(A)
    R, info = lapack.dpotri(A, lower=lower)
    symmetrify(R)
    return R, info


def dtrtri(L):
    """linalg.py:217-227."""
    L = np.asfortranarray(L)
    return lapack.dtrtri(L, lower=1)[0]


def pdinv(A, maxtries=5, with_Li=True):
    """linalg.py:193-214.  Returns (Ai, L, Li, logdet, jitter).

    ``with_Li=False`` skips the dtrtri whose result exact inference never uses
    (the "minimal" CPU-baseline total; the as-GPy total keeps it).
    """
    L, jitter = jitchol(A, maxtries)
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    Li = dtrtri(L) if with_Li else None
    Ai, _ = dpotri(L, lower=1)
    symmetrify(Ai)
    return Ai, L, Li, logdet, jitter


# ----------------------------------------------------------------------------
# GPy/GPy/kern/src/stationary.py, rbf.py
# ----------------------------------------------------------------------------
class Stationary(object):
    """Restatement of ``Stationary`` (stationary.py:23-380) without paramz."""

    name = "stationary"

    def __init__(self, input_dim, variance=1.0, lengthscale=None, ARD=False, Gower=False, space=None):
        # stationary.py:61-82
        self.Gower = Gower
        self.space = space
        self.input_dim = int(input_dim)
        self.ARD = bool(ARD)
        if not ARD:
            lengthscale = np.ones(1) if lengthscale is None else np.asarray(lengthscale, dtype=float).reshape(-1)
            assert lengthscale.size == 1, "Only 1 lengthscale needed for non-ARD kernel"
        else:
            if lengthscale is not None:
                lengthscale = np.asarray(lengthscale, dtype=float).reshape(-1)
                assert lengthscale.size in [1, input_dim], "Bad number of lengthscales"
                if lengthscale.size != input_dim:
                    lengthscale = np.ones(input_dim) * lengthscale
            else:
                lengthscale = np.ones(self.input_dim)
        self.lengthscale = np.array(lengthscale, dtype=float)
        self.variance = float(variance)

    # -- to be provided by subclasses -------------------------------------
    def K_of_r(self, r):
        raise NotImplementedError

    def dK_dr(self, r):
        raise NotImplementedError

    # -- distances -----------------------------------------------------------
    def _unscaled_dist(self, X, X2=None):
        """stationary.py:155-173."""
        if X2 is None:
            Xsq = np.sum(np.square(X), 1)
            r2 = -2.0 * tdot(X) + (Xsq[:, None] + Xsq[None, :])
            diag_view(r2)[:, ] = 0.0
            r2 = np.clip(r2, 0, np.inf)
            return np.sqrt(r2)
        else:
            X1sq = np.sum(np.square(X), 1)
            X2sq = np.sum(np.square(X2), 1)
            r2 = -2.0 * np.dot(X, X2.T) + (X1sq[:, None] + X2sq[None, :])
            r2 = np.clip(r2, 0, np.inf)
            return np.sqrt(r2)

    def _scaled_dist(self, X, X2=None):
        """stationary.py:175-193."""
        if self.ARD:
            if X2 is not None:
                X2 = X2 / self.lengthscale
            return self._unscaled_dist(X / self.lengthscale, X2)
        else:
            return self._unscaled_dist(X, X2) / self.lengthscale

    def K(self, X, X2=None):
        """stationary.py:107-140: the fork's Gower branch (:116-135) and the Euclidean branch (:137-139)."""
        if self.Gower and (self.space is not None):
            const_dims = self.space.get_continuous_dims()
            disc_dims = self.space.get_discrete_dims()
            lengthscale = self.space.lengthscales()
            numDims = X.shape[1]
            if X2 is None:
                X2 = X
            K_1D = [None for dim in range(numDims)]
            for index, const_dim in enumerate(const_dims, start=0):
                r = abs(X[:, np.newaxis, const_dim] - X2[np.newaxis, :, const_dim]) / lengthscale[index]
                K_1D[const_dim] = self.K_of_r(r)
            for disc_dim in disc_dims:
                r = (X[:, np.newaxis, disc_dim] != X2[np.newaxis, :, disc_dim]).astype(int)
                K_1D[disc_dim] = self.K_of_r(r)
            kernel = K_1D[0]
            for dim in range(numDims - 1):
                kernel = kernel * K_1D[dim + 1]
            return kernel
        r = self._scaled_dist(X, X2)
        return self.K_of_r(r)

    def Kdiag(self, X):
        """stationary.py:195-198."""
        ret = np.empty(X.shape[0])
        ret[:] = self.variance
        return ret

    def _inv_dist(self, X, X2=None):
        """stationary.py:251-258."""
        dist = self._scaled_dist(X, X2).copy()
        return 1.0 / np.where(dist != 0.0, dist, np.inf)

    def _lengthscale_grads_pure(self, tmp, X, X2):
        """stationary.py:260-261."""
        return -np.array([np.sum(tmp * np.square(X[:, q:q + 1] - X2[:, q:q + 1].T))
                          for q in range(self.input_dim)]) / self.lengthscale ** 3

    def update_gradients_full(self, dL_dK, X, X2=None):
        """stationary.py:218-238.  Returns (dvariance, dlengthscale)."""
        dvariance = np.sum(self.K(X, X2) * dL_dK) / self.variance
        dL_dr = self.dK_dr(self._scaled_dist(X, X2)) * dL_dK
        if self.ARD:
            tmp = dL_dr * self._inv_dist(X, X2)
            if X2 is None:
                X2 = X
            dlengthscale = self._lengthscale_grads_pure(tmp, X, X2)
        else:
            r = self._scaled_dist(X, X2)
            dlengthscale = np.atleast_1d(-np.sum(dL_dr * r) / self.lengthscale)
        return float(dvariance), np.asarray(dlengthscale, dtype=float).reshape(-1)

    def gradients_X(self, dL_dK, X, X2=None):
        """stationary.py:336-352 (``_gradients_X_pure``)."""
        invdist = self._inv_dist(X, X2)
        dL_dr = self.dK_dr(self._scaled_dist(X, X2)) * dL_dK
        tmp = invdist * dL_dr
        if X2 is None:
            tmp = tmp + tmp.T
            X2 = X
        grad = np.empty(X.shape, dtype=np.float64)
        for q in range(self.input_dim):
            np.sum(tmp * (X[:, q][:, None] - X2[:, q][None, :]), axis=1, out=grad[:, q])
        return grad / self.lengthscale ** 2

    def gradients_X_diag(self, dL_dKdiag, X):
        """stationary.py:366-367."""
        return np.zeros(X.shape)


class RBF(Stationary):
    """rbf.py:12-57."""
    name = "rbf"

    def K_of_r(self, r):
        return self.variance * np.exp(-0.5 * r ** 2)  # rbf.py:50-51

    def dK_dr(self, r):
        return -r * self.K_of_r(r)  # rbf.py:53-54


class Matern52(Stationary):
    """stationary.py:546-579."""
    name = "Mat52"

    def K_of_r(self, r):
        return self.variance * (1 + np.sqrt(5.) * r + 5. / 3 * r ** 2) * np.exp(-np.sqrt(5.) * r)  # :575-576

    def dK_dr(self, r):
        return self.variance * (10. / 3 * r - 5. * r - 5. * np.sqrt(5.) / 3 * r ** 2) * np.exp(-np.sqrt(5.) * r)  # :578-579


KERNELS = {"rbf": RBF, "RBF": RBF, "Mat52": Matern52, "Matern52": Matern52, "matern52": Matern52}


def make_kernel(name, input_dim, variance=1.0, lengthscale=None, ARD=False, Gower=False, space=None):
    return KERNELS[name](input_dim, variance=variance, lengthscale=lengthscale, ARD=ARD, Gower=Gower, space=space)


# ----------------------------------------------------------------------------
# GPy/GPy/util/normalizer.py, GPyOpt/GPyOpt/util/general.py
# ----------------------------------------------------------------------------
class Standardize(object):
    """GPy/GPy/util/normalizer.py:85-108."""

    def __init__(self):
        self.mean = None

    def scale_by(self, Y):
        Y = np.ma.masked_invalid(Y, copy=False)
        self.mean = Y.mean(0).view(np.ndarray)
        self.std = Y.std(0).view(np.ndarray)

    def normalize(self, Y):
        return (Y - self.mean) / self.std

    def inverse_mean(self, X):
        return (X * self.std) + self.mean

    def inverse_variance(self, var):
        return var * (self.std ** 2)


def normalize(Y, normalization_type="stats"):
    """GPyOpt/GPyOpt/util/general.py:203-234."""
    Y = np.asarray(Y, dtype=float)
    if np.max(Y.shape) != Y.size:
        raise NotImplementedError("Only 1-dimensional arrays are supported.")
    if normalization_type == "stats":
        Y_norm = Y - Y.mean()
        std = Y.std()
        if std > 0:
            Y_norm /= std
    elif normalization_type == "maxmin":
        Y_norm = Y - Y.min()
        y_range = np.ptp(Y)
        if y_range > 0:
            Y_norm /= y_range
            Y_norm = 2 * (Y_norm - 0.5)
    else:
        raise ValueError("Unknown normalization type: {}".format(normalization_type))
    return Y_norm


def get_quantiles(acquisition_par, fmin, m, s):
    """GPyOpt/GPyOpt/util/general.py:113-129 (mutates ``s`` like the reference)."""
    if isinstance(s, np.ndarray):
        s[s < 1e-10] = 1e-10
    elif s < 1e-10:
        s = 1e-10
    u = (fmin - m - acquisition_par) / s
    phi = np.exp(-0.5 * u ** 2) / np.sqrt(2 * np.pi)
    Phi = 0.5 * erfc(-u / np.sqrt(2))
    return (phi, Phi, u)


# ----------------------------------------------------------------------------
# GPy inference + prediction
# ----------------------------------------------------------------------------
def exact_gaussian_inference(kern, X, Y, noise_var, maxtries=5, with_Li=True, K=None):
    """ExactGaussianInference.inference, exact_gaussian_inference.py:37-74 (m=0).

    Returns a dict with the Posterior's fields (woodbury_chol=L,
    woodbury_vector=alpha, K), the log marginal likelihood and the gradient
    dict entries (dL_dK, dL_dthetaL), plus Wi/logdet/jitter for inspection.
    """
    YYT_factor = Y
    if K is None:
        K = kern.K(X)
    Ky = K.copy()
    diag_add(Ky, noise_var + 1e-8)
    Wi, LW, LWi, W_logdet, jitter = pdinv(Ky, maxtries, with_Li=with_Li)
    alpha, _ = dpotrs(LW, YYT_factor, lower=1)
    log_marginal = 0.5 * (-Y.size * LOG_2_PI - Y.shape[1] * W_logdet - np.sum(alpha * YYT_factor))
    dL_dK = 0.5 * (tdot(alpha) - Y.shape[1] * Wi)
    dL_dthetaL = np.diag(dL_dK).sum()  # gaussian.py:78-79
    return dict(K=K, L=LW, alpha=alpha, lml=float(log_marginal), dL_dK=dL_dK,
                dL_dthetaL=float(dL_dthetaL), Wi=Wi, logdet=float(W_logdet), jitter=float(jitter))


def raw_predict(kern, X, L, alpha, Xnew, full_cov=False):
    """PosteriorExact._raw_predict, posterior.py:273-302 (2-D woodbury_chol)."""
    Kx = kern.K(X, Xnew)
    mu = np.dot(Kx.T, alpha)
    if mu.ndim == 1:
        mu = mu.reshape(-1, 1)
    if full_cov:
        Kxx = kern.K(Xnew)
        tmp = dtrtrs(L, Kx)[0]
        var = Kxx - tdot(tmp.T)
    else:
        Kxx = kern.Kdiag(Xnew)
        tmp = dtrtrs(L, Kx)[0]
        var = (Kxx - np.square(tmp).sum(0))[:, None]
    return mu, var


def woodbury_inv(L):
    """Posterior.woodbury_inv, posterior.py:176-196 (dpotri + symmetrify)."""
    Wi, _ = dpotri(L, lower=1)
    symmetrify(Wi)
    return Wi


class OracleGP(object):
    """GPRegression restated: gp_regression.py:29-36, core/gp.py:38-110,258-354,407-454.

    Hyper-parameters are plain floats/arrays (no paramz).  ``normalizer=True``
    standardises Y as GP.__init__ does (gp.py:73-84).
    """

    def __init__(self, X, Y, kernel, noise_var=1.0, normalizer=False):
        self.kern = kernel
        self.noise_var = float(noise_var)
        self.set_XY(X, Y, normalizer)

    def set_XY(self, X, Y, normalizer=None):
        X = np.asarray(X, dtype=float)
        Y = np.asarray(Y, dtype=float)
        if normalizer is not None:
            self.normalizer = Standardize() if normalizer else None
        self.X, self.Y = X, Y
        if self.normalizer is not None:
            self.normalizer.scale_by(Y)
            self.Y_normalized = self.normalizer.normalize(Y)
        else:
            self.Y_normalized = Y
        self.output_dim = Y.shape[1]
        self._post = None
        self._Wi = None

    def parameters_changed(self, **kw):
        """gp.py:258-271."""
        self._post = exact_gaussian_inference(self.kern, self.X, self.Y_normalized, self.noise_var, **kw)
        self._Wi = self._post["Wi"]
        return self._post

    @property
    def posterior(self):
        if self._post is None:
            self.parameters_changed()
        return self._post

    def log_likelihood(self):
        """gp.py:273-277."""
        return self.posterior["lml"]

    def gradients(self):
        """Natural-space gradients pushed down in gp.py:268-269.

        Returns (dvariance, dlengthscale[1 or D], dnoise).
        """
        p = self.posterior
        dvar, dlen = self.kern.update_gradients_full(p["dL_dK"], self.X)
        return dvar, dlen, p["dL_dthetaL"]

    def _raw_predict(self, Xnew, full_cov=False):
        p = self.posterior
        return raw_predict(self.kern, self.X, p["L"], p["alpha"], Xnew, full_cov)

    def predict(self, Xnew, full_cov=False, include_likelihood=True):
        """gp.py:297-354 + gaussian.py:102-110 + normalizer.py:98-102."""
        mean, var = self._raw_predict(Xnew, full_cov=full_cov)
        if include_likelihood:
            if full_cov:
                var = var + np.eye(var.shape[0]) * self.noise_var
            else:
                var = var + self.noise_var
        if self.normalizer is not None:
            mean = self.normalizer.inverse_mean(mean)
            var = self.normalizer.inverse_variance(var)
        return mean, var

    def predict_noiseless(self, Xnew, full_cov=False):
        return self.predict(Xnew, full_cov, include_likelihood=False)

    def predict_quantiles(self, X, quantiles=(2.5, 97.5)):
        """gp.py:384-405 with Gaussian.predictive_quantiles (likelihoods/gaussian.py:118-119):
        ``[norm.ppf(q/100) * sqrt(var + noise) + mu for q in quantiles]`` on the raw posterior (gp.py:398), then the
        normaliser's inverse_mean on each quantile (gp.py:403-404)."""
        from scipy import stats
        m, v = self._raw_predict(X, full_cov=False)
        qs = [stats.norm.ppf(q / 100.) * np.sqrt(v + self.noise_var) + m for q in quantiles]
        if self.normalizer is not None:
            qs = [self.normalizer.inverse_mean(q) for q in qs]
        return qs

    def posterior_covariance_between_points(self, X1, X2):
        """gp.py:714-721 -> Posterior.covariance_between_points, posterior.py:109-128."""
        p = self.posterior
        Kx1 = self.kern.K(self.X, X1)
        Kx2 = self.kern.K(self.X, X2)
        K12 = self.kern.K(X1, X2)
        tmp1 = dtrtrs(p["L"], Kx1)[0]
        tmp2 = dtrtrs(p["L"], Kx2)[0]
        return K12 - tmp1.T.dot(tmp2)

    def predictive_gradients(self, Xnew):
        """gp.py:407-454."""
        p = self.posterior
        mean_jac = np.empty((Xnew.shape[0], Xnew.shape[1], self.output_dim))
        for i in range(self.output_dim):
            mean_jac[:, :, i] = self.kern.gradients_X(p["alpha"][:, i:i + 1].T, Xnew, self.X)
        dv_dX = self.kern.gradients_X_diag(np.ones(Xnew.shape[0]), Xnew)
        Wi = self._Wi if self._Wi is not None else woodbury_inv(p["L"])
        alpha = -2.0 * np.dot(self.kern.K(Xnew, self.X), Wi)
        dv_dX = dv_dX + self.kern.gradients_X(alpha, Xnew, self.X)
        return mean_jac, dv_dX


# ----------------------------------------------------------------------------
# GPyOpt model adapter + acquisitions
# ----------------------------------------------------------------------------
class OracleGPModel(object):
    """GPModel restated at fixed hyper-parameters: GPyOpt/GPyOpt/models/gpmodel.py:95-142."""

    def __init__(self, gp):
        self.model = gp

    def _predict(self, X, full_cov, include_likelihood):
        if X.ndim == 1:
            X = X[None, :]
        m, v = self.model.predict(X, full_cov=full_cov, include_likelihood=include_likelihood)
        v = np.clip(v, 1e-10, np.inf)
        return m, v

    def predict(self, X, with_noise=True):
        m, v = self._predict(X, False, with_noise)
        return m, np.sqrt(v)

    def get_fmin(self):
        return self.model.predict(self.model.X)[0].min()

    def predict_withGradients(self, X):
        if X.ndim == 1:
            X = X[None, :]
        m, v = self.model.predict(X)
        v = np.clip(v, 1e-10, np.inf)
        dmdx, dvdx = self.model.predictive_gradients(X)
        dmdx = dmdx[:, :, 0]
        dsdx = dvdx / (2 * np.sqrt(v))
        return m, np.sqrt(v), dmdx, dsdx


def acq_EI(model, x, jitter=0.01, fmin=None):
    """AcquisitionEI._compute_acq, GPyOpt/GPyOpt/acquisitions/EI.py:32-40."""
    m, s = model.predict(x)
    if fmin is None:
        fmin = model.get_fmin()
    phi, Phi, u = get_quantiles(jitter, fmin, m, s)
    return s * (u * Phi + phi)


def acq_EI_withGradients(model, x, jitter=0.01, fmin=None):
    """EI.py:42-51."""
    if fmin is None:
        fmin = model.get_fmin()
    m, s, dmdx, dsdx = model.predict_withGradients(x)
    phi, Phi, u = get_quantiles(jitter, fmin, m, s)
    f_acqu = s * (u * Phi + phi)
    df_acqu = dsdx * phi - Phi * dmdx
    return f_acqu, df_acqu


def acq_LCB(model, x, exploration_weight=2.0):
    """AcquisitionLCB._compute_acq, LCB.py:31-37."""
    m, s = model.predict(x)
    return -m + exploration_weight * s


def acq_LCB_withGradients(model, x, exploration_weight=2.0):
    """LCB.py:39-46."""
    m, s, dmdx, dsdx = model.predict_withGradients(x)
    return -m + exploration_weight * s, -dmdx + exploration_weight * dsdx


def acq_MPI(model, x, jitter=0.01, fmin=None):
    """AcquisitionMPI._compute_acq, MPI.py:32-40."""
    m, s = model.predict(x)
    if fmin is None:
        fmin = model.get_fmin()
    _, Phi, _ = get_quantiles(jitter, fmin, m, s)
    return Phi


def acq_MPI_withGradients(model, x, jitter=0.01, fmin=None):
    """MPI.py:42-51."""
    if fmin is None:
        fmin = model.get_fmin()
    m, s, dmdx, dsdx = model.predict_withGradients(x)
    phi, Phi, u = get_quantiles(jitter, fmin, m, s)
    return Phi, -(phi / s) * (dmdx + dsdx * u)


def acquisition_function(f_acqu, indicator=1.0, cost=1.0):
    """AcquisitionBase.acquisition_function, acquisitions/base.py:33-39 (negated)."""
    return -(f_acqu * indicator) / cost


# ----------------------------------------------------------------------------
# Synthetic workloads of SURVEY.md 8(d) (shared by tests and bench.py)
# ----------------------------------------------------------------------------
def usable_cpus():
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup CPU quota (cpu.max / cfs_quota) when
    there is one.  A GPU box reports 64+ cores through os.cpu_count() while the job's share is 16: BLAS with 64 threads
    on 16 CPUs oversubscribes (dpotrf at 30 GFLOP/s in BENCH_r02) -- timing legs limit their thread pools to this."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = float(f1.read()), float(f2.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    env = os.environ.get("GPHIP_CPU_THREADS")
    if env:
        n = max(1, int(env))
    return n


def limit_blas_threads(n=None):
    """threadpoolctl limit of every BLAS / OpenMP pool to ``n`` (default usable_cpus()); returns (limiter, n).  Keep the
    returned object alive for as long as the limit should hold."""
    n = int(n or usable_cpus())
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n), n
    except Exception:  # noqa: BLE001
        return None, n


def synthetic_problem(N, D, M, seed=1234, standardize=True):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (N, D))
    f = np.sin(2 * np.pi * X).sum(1, keepdims=True) / np.sqrt(D)
    Y = f + 0.05 * np.random.default_rng(seed + 1).standard_normal((N, 1))
    if standardize:
        Y = normalize(Y, "stats")
    Xs = np.random.default_rng(seed + 2).uniform(0, 1, (M, D))
    return X, Y, Xs


def default_lengthscale(D, ARD):
    if ARD:
        return 0.2 + 0.04 * np.arange(D)
    return np.array([0.25 * np.sqrt(D)])


def fit_predict_iteration(kern, X, Y, noise_var, Xs, as_gpy=False):
    """One "fit + predict" unit of the headline metric on the CPU.

    as_gpy=False: K, dpotrf, alpha, lml, K(X,Xs), dtrtrs, mean/var -- only what
    is algebraically needed.  as_gpy=True additionally does what pdinv does
    (dtrtri, dpotri + symmetrify, dL_dK) as GP.parameters_changed would.
    Returns (lml, mean, var, phase_times dict).
    """
    import time
    t = {}
    t0 = time.perf_counter()
    K = kern.K(X)
    t["K_build"] = time.perf_counter() - t0
    if as_gpy:
        t0 = time.perf_counter()
        post = exact_gaussian_inference(kern, X, Y, noise_var, K=K)
        t["inference_pdinv"] = time.perf_counter() - t0
        L, alpha, lml = post["L"], post["alpha"], post["lml"]
    else:
        t0 = time.perf_counter()
        Ky = K
        diag_add(Ky, noise_var + 1e-8)
        L, _ = jitchol(Ky)
        t["dpotrf"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        alpha, _ = dpotrs(L, Y, lower=1)
        logdet = 2.0 * np.sum(np.log(np.diag(L)))
        lml = 0.5 * (-Y.size * LOG_2_PI - Y.shape[1] * logdet - np.sum(alpha * Y))
        t["alpha_lml"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    Kx = kern.K(X, Xs)
    t["K_cross"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mu = np.dot(Kx.T, alpha)
    tmp = dtrtrs(L, Kx)[0]
    var = (kern.Kdiag(Xs) - np.square(tmp).sum(0))[:, None]
    t["dtrtrs_var"] = time.perf_counter() - t0
    return float(lml), mu, var, t


# ----------------------------------------------------------------------------
# Local penalisation (GPyOpt/GPyOpt/acquisitions/LP.py)
# ----------------------------------------------------------------------------
def lp_hammer_precompute(model, x0, L, Min):
    """AcquisitionLP._hammer_function_precompute, LP.py:49-62."""
    if x0.ndim == 1:
        x0 = x0[None, :]
    m = model.predict(x0)[0]
    pred = model.predict(x0)[1].copy()
    pred[pred < 1e-16] = 1e-16
    s = np.sqrt(pred)
    return ((m - Min) / L).flatten(), (s / L).flatten()


def lp_penalized_acquisition(neg_acq, x, X_batch, r_x0, s_x0, transform="none"):
    """AcquisitionLP._penalized_acquisition, LP.py:70-89.  neg_acq = base acquisition_function(x) ([M,1], negated)."""
    from scipy.stats import norm
    fval = -neg_acq[:, 0].copy()
    if transform == "softplus":
        fval_org = fval.copy()
        fval[fval_org >= 40.] = np.log(fval_org[fval_org >= 40.])
        fval[fval_org < 40.] = np.log(np.log1p(np.exp(fval_org[fval_org < 40.])))
    elif transform == "none":
        fval = np.log(fval + 1e-50)
    fval = -fval
    if X_batch is not None:
        h = norm.logcdf((np.sqrt((np.square(np.atleast_2d(x)[:, None, :] - np.atleast_2d(X_batch)[None, :, :])).sum(-1))
                         - r_x0) / s_x0)
        fval += -h.sum(axis=-1)
    return fval


def lp_d_acquisition(neg_acq, neg_dacq, x, X_batch, r_x0, s_x0, transform="none"):
    """AcquisitionLP.d_acquisition_function, LP.py:112-133, with _d_hammer_function (LP.py:91-103) verbatim --
    including its omission of the direction (x - x0) / |x - x0|: the reference sums a scalar per point over the
    batch and broadcasts it over the input dimensions.  Restated as is (bug-compatible drop-in)."""
    from scipy.stats import norm
    x = np.atleast_2d(x)
    fval = -neg_acq[:, 0]
    if transform == "softplus":
        scale = 1. / (np.log1p(np.exp(fval)) * (1. + np.exp(-fval)))
    elif transform == "none":
        scale = 1. / fval
    else:
        scale = np.ones_like(fval)
    g = scale[:, None] * neg_dacq
    if X_batch is None:
        return g
    dx = x[:, None, :] - np.atleast_2d(X_batch)[None, :, :]
    nm = np.sqrt((np.square(dx)).sum(-1))
    z = (nm - r_x0) / s_x0
    h_func = norm.cdf(z)
    d = 1. / (s_x0 * np.sqrt(2 * np.pi) * h_func) * np.exp(-np.square(z) / 2) / nm
    d[h_func < 1e-50] = 0.
    return g - d[:, :, None].sum(axis=1)


# ----------------------------------------------------------------------------
# Local-penalisation evaluator (GPyOpt/GPyOpt/core/evaluators/batch_local_penalization.py)
# ----------------------------------------------------------------------------
def samples_multidimensional_uniform(bounds, num_data):
    """GPyOpt/GPyOpt/util/general.py:63-73 (one np.random.uniform call per dimension, in this order)."""
    dim = len(bounds)
    Z_rand = np.zeros(shape=(num_data, dim))
    for k in range(0, dim):
        Z_rand[:, k] = np.random.uniform(low=bounds[k][0], high=bounds[k][1], size=num_data)
    return Z_rand


def estimate_L(model, bounds):
    """batch_local_penalization.py:52-70: max over the box of |d mean / dx|, 500 uniform samples + the training
    inputs as starting candidates, then L-BFGS-B from the best.  ``model`` is the GP (predictive_gradients, X)."""
    import scipy.optimize

    def df(x, model, x0):
        x = np.atleast_2d(x)
        dmdx, _ = model.predictive_gradients(x)
        res = np.sqrt((dmdx * dmdx).sum(1))
        return -res

    samples = samples_multidimensional_uniform(bounds, 500)
    samples = np.vstack([samples, model.X])
    pred_samples = df(samples, model, 0)
    x0 = samples[np.argmin(pred_samples)]
    res = scipy.optimize.minimize(lambda x: float(np.ravel(df(x, model, x0))[0]), x0, method='L-BFGS-B',
                                  bounds=bounds, options={'maxiter': 200})
    L = -float(res.fun)
    if L < 1e-7:
        L = 10
    return L


def lp_compute_batch(acquisition, batch_size):
    """LocalPenalization.compute_batch, batch_local_penalization.py:22-49; ``acquisition`` offers update_batches,
    optimize, model.model (the GP) and space.get_bounds()."""
    acquisition.update_batches(None, None, None)
    X_batch = acquisition.optimize()[0]
    k = 1
    if batch_size > 1:
        L = estimate_L(acquisition.model.model, acquisition.space.get_bounds())
        Min = acquisition.model.model.Y.min()
    while k < batch_size:
        acquisition.update_batches(X_batch, L, Min)
        new_sample = acquisition.optimize()[0]
        X_batch = np.vstack((X_batch, new_sample))
        k += 1
    acquisition.update_batches(None, None, None)
    return X_batch


# ----------------------------------------------------------------------------
# The fork's mixed-variable design space and the thesis driver's table loop
# ----------------------------------------------------------------------------
class MixedSpace(object):
    """The four ``Design_space`` methods the Gower path calls, for one-dimensional continuous / discrete variables
    given as GPyOpt domain dictionaries: ``lengthscales`` (GPyOpt/GPyOpt/core/task/space.py:351-362: range of every
    continuous variable, in order), ``get_continuous_dims`` (:436-445), ``get_discrete_dims`` (:483-492),
    ``get_bounds`` (:263-272 over core/task/variables.py:91-92,169-170: a discrete variable's bounds are (min, max) of its
    domain)."""

    def __init__(self, domain):
        self.domain = [dict(d) for d in domain]
        for d in self.domain:
            assert d["type"] in ("continuous", "discrete") and int(d.get("dimensionality", 1)) == 1
        self.dimensionality = len(self.domain)

    def lengthscales(self):
        return [d["domain"][-1] - d["domain"][0] for d in self.domain if d["type"] == "continuous"]

    def get_continuous_dims(self):
        return [i for i, d in enumerate(self.domain) if d["type"] == "continuous"]

    def get_discrete_dims(self):
        return [i for i, d in enumerate(self.domain) if d["type"] == "discrete"]

    def get_bounds(self):
        return [(min(d["domain"]), max(d["domain"])) if d["type"] == "discrete" else tuple(d["domain"])
                for d in self.domain]

    def draw(self, rng, n):
        """n uniform rows of the mixed domain (test inputs; not a reference function)."""
        cols = [rng.choice(np.asarray(d["domain"], dtype=float), n) if d["type"] == "discrete"
                else rng.uniform(d["domain"][0], d["domain"][1], n) for d in self.domain]
        return np.stack(cols, axis=1)


class OracleLP(object):
    """AcquisitionLP over the oracle model (GPyOpt/GPyOpt/acquisitions/LP.py:26-140) with EI / LCB / MPI as the base
    acquisition: ``update_batches`` :41-47, ``acquisition_function`` :105-110, ``acquisition_function_withGradients``
    :135-140.  ``base`` in {"EI", "LCB", "MPI"}; LCB switches the transform to softplus (:31-32)."""

    def __init__(self, model, space, base="EI", par=None, transform="none"):
        self.model, self.space, self.base = model, space, base
        self.par = par if par is not None else {"EI": 0.01, "LCB": 2.0, "MPI": 0.01}[base]
        self.transform = "softplus" if (base == "LCB" and transform == "none") else transform
        self.X_batch = self.r_x0 = self.s_x0 = None

    def _neg_base(self, x):
        f = {"EI": acq_EI, "LCB": acq_LCB, "MPI": acq_MPI}[self.base]
        return acquisition_function(f(self.model, x, self.par))

    def _neg_base_withGradients(self, x):
        f = {"EI": acq_EI_withGradients, "LCB": acq_LCB_withGradients, "MPI": acq_MPI_withGradients}[self.base]
        a, da = f(self.model, x, self.par)
        return -a, -da     # base.py:42-50 with unit cost and no constraints

    def update_batches(self, X_batch, L, Min):
        self.X_batch = X_batch
        if X_batch is not None:
            self.r_x0, self.s_x0 = lp_hammer_precompute(self.model, X_batch, L, Min)

    def acquisition_function(self, x):
        return lp_penalized_acquisition(self._neg_base(x), x, self.X_batch, self.r_x0, self.s_x0, self.transform)

    def acquisition_function_withGradients(self, x):
        x = np.atleast_2d(x)
        neg, dneg = self._neg_base_withGradients(x)
        return self.acquisition_function(x), lp_d_acquisition(neg, dneg, x, self.X_batch, self.r_x0, self.s_x0,
                                                              self.transform)


def lp_table_batch(lp, configurations, batch_size):
    """The thesis driver's batch over a table of feasible configurations, run.py:1234-1258: the un-penalised arg-MAX of
    ``AcquisitionLP.acquisition_function`` (:1239-1241), then ``estimate_L`` on the GP and ``Min = model.Y.min()``
    (:1244-1245), then ``batch_size - 1`` rounds of update_batches / masked arg-max (:1246-1256).  Returns
    (row indices, L, Min).  ``lp`` offers update_batches, acquisition_function, model.model and space.get_bounds()."""
    suggested = []
    lp.update_batches(None, None, None)
    acquisition_values = lp.acquisition_function(configurations)
    index_best = int(np.argmax(acquisition_values))
    suggested.append(index_best)
    X_batch = configurations[index_best]
    get = 1
    L = estimate_L(lp.model.model, lp.space.get_bounds())
    Min = lp.model.model.Y.min()
    while get < batch_size:
        lp.update_batches(X_batch, L, Min)
        acquisition_values = lp.acquisition_function(configurations)
        masked = np.ma.array(acquisition_values, mask=False)
        for existing_index in suggested:
            masked.mask[existing_index] = True
        index_best = int(np.argmax(masked))
        suggested.append(index_best)
        X_batch = np.vstack((X_batch, configurations[index_best]))
        get += 1
    return suggested, L, Min

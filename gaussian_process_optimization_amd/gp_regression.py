"""``GPRegression`` -- host mirror of ``GPy.models.GPRegression`` backed by libgphip.

Reference: GPy/GPy/models/gp_regression.py:29-36 (constructor), GPy/GPy/core/gp.py
(GP.__init__ :38-110, set_XY :202-238, parameters_changed :258-271, log_likelihood
:273-277, predict :297-354, predict_noiseless :356-379, predictive_gradients :407-454,
optimize :643-664), GPy/GPy/likelihoods/gaussian.py (variance :43, predictive_values
:102-110), GPy/GPy/util/normalizer.py:85-108 (Standardize).

All numerics (K build, Cholesky, solves, LML, gradients, posterior) run on the
GPU through the C-ABI; this module is bookkeeping: hyper-parameters, the Y
normaliser, lazy re-fit on parameter change, and the L-BFGS driver.
"""
import numpy as np
from scipy import optimize as _sopt

from . import _lib
from .kern import RBF, Stationary, gower_config
from .parameterization import Param, Parameterized


class Gaussian(Parameterized):
    """GPy.likelihoods.Gaussian: one positive ``variance`` parameter (gaussian.py:31-47)."""

    def __init__(self, variance=1., name="Gaussian_noise"):
        super(Gaussian, self).__init__(name)
        self.variance = Param("variance", np.atleast_1d(float(variance)))
        self.link_parameter(self.variance)

    def gaussian_variance(self, Y_metadata=None):
        return self.variance  # gaussian.py:69-70


class Standardize(object):
    """GPy/GPy/util/normalizer.py:85-108."""

    def __init__(self):
        self.mean = None

    def scale_by(self, Y):
        Y = np.ma.masked_invalid(Y, copy=False)
        self.mean = Y.mean(0).view(np.ndarray)
        self.std = Y.std(0).view(np.ndarray)

    def normalize(self, Y):
        if not self.scaled():
            raise AttributeError("Norm object not initialized yet, try calling scale_by(data) first.")
        return (Y - self.mean) / self.std

    def inverse_mean(self, X):
        return (X * self.std) + self.mean

    def inverse_variance(self, var):
        return var * (self.std ** 2)

    def inverse_covariance(self, covariance):
        return covariance[..., np.newaxis] * (self.std ** 2)

    def scaled(self):
        return self.mean is not None


class _PosteriorView(object):
    """Read-only stand-in for GPy's Posterior object (posterior.py:9-270): the fields gp.py uses."""

    def __init__(self, model):
        self._m = model

    @property
    def woodbury_vector(self):
        self._m._ensure_fit()
        return self._m._h.alpha()

    @property
    def woodbury_chol(self):
        self._m._ensure_fit()
        return self._m._h.chol()

    @property
    def woodbury_inv(self):
        self._m._ensure_fit()
        return self._m._h.woodbury_inv()


class GPRegression(Parameterized):
    """Gaussian Process model for regression with a Gaussian likelihood, on one MI355X.

    :param X: input observations [N, D]
    :param Y: observed values [N, P]
    :param kernel: ``kern.RBF`` / ``kern.Matern52`` (defaults to RBF, gp_regression.py:31-32)
    :param normalizer: ``True`` standardises Y (gp.py:73-84)
    :param noise_var: Gaussian noise variance (default 1)
    :param device: HIP device ordinal
    """

    def __init__(self, X, Y, kernel=None, Y_metadata=None, normalizer=None, noise_var=1., mean_function=None,
                 device=0, name="GP regression"):
        super(GPRegression, self).__init__(name)
        if mean_function is not None:
            raise NotImplementedError("mean functions are outside the accelerated path")
        X = np.asarray(X, dtype=float)
        if kernel is None:
            kernel = RBF(X.shape[1])
        if not isinstance(kernel, Stationary):
            raise TypeError("kernel must be kern.RBF or kern.Matern52")
        self.kern = kernel
        self.kern._device = int(device)     # kern.K evaluates on the owning model's device (kern.py, _scratch_handle)
        self.likelihood = Gaussian(variance=noise_var)
        self.Gaussian_noise = self.likelihood
        self.link_parameters(self.kern, self.likelihood)
        if normalizer is True:
            self.normalizer = Standardize()
        elif normalizer in (False, None):
            self.normalizer = None
        else:
            self.normalizer = normalizer
        self.Y_metadata = Y_metadata
        self.max_jitter_tries = 5  # jitchol default, linalg.py:56
        # hyper-parameter search of a Gower model.  'exact' (default): the LML's true gradient from ONE device call -- the Gower
        # K is variance^D times a product that depends on neither hyper-parameter, so d LML / d variance = D x the fork's
        # variance entry and d LML / d lengthscale = 0 (stationary.py:116-135); 'differences': forward differences of the
        # device LML (the same numbers, D + 2 fits per evaluation); 'fork': the fork's own update_gradients_full values
        # (stationary.py:218-238 on a Gower K: not derivatives of the objective -- what the reference's optimiser follows)
        self.gower_gradients = 'exact'
        self._h = _lib.Handle(device)
        self._groups = {}        # replica groups over several devices, keyed by the device tuple (_device_group)
        self._data_epoch = 0
        self._dirty = True
        self._lml = None
        self._jitter = 0.0
        self.posterior = _PosteriorView(self)
        self.set_XY(X, Y)

    # -- data -------------------------------------------------------------------------
    def set_XY(self, X=None, Y=None):
        """GP.set_XY, gp.py:202-238."""
        if Y is not None:
            Y = np.asarray(Y, dtype=float)
            if Y.ndim == 1:
                Y = Y[:, None]
            if self.normalizer is not None:
                self.normalizer.scale_by(Y)
                self.Y_normalized = self.normalizer.normalize(Y)
            else:
                self.Y_normalized = Y
            self.Y = Y
        if X is not None:
            X = np.asarray(X, dtype=float)
            assert X.ndim == 2
            self.X = X
        assert self.X.shape[0] == self.Y.shape[0]
        assert self.X.shape[1] == self.kern.input_dim
        self.num_data, self.input_dim = self.X.shape
        self.output_dim = self.Y.shape[1]
        self._h.set_data(self.X, self.Y_normalized)
        self._data_epoch += 1
        self._dirty = True

    def set_X(self, X):
        self.set_XY(X=X)

    def set_Y(self, Y):
        self.set_XY(Y=Y)

    # -- (re)fit ----------------------------------------------------------------------
    def _on_change(self):
        self._dirty = True

    def _ensure_fit(self):
        """GP.parameters_changed, gp.py:258-271: inference on the device when anything changed."""
        if not self._dirty:
            return
        self._push_params()
        self._lml, self._logdet, self._jitter = self._h.fit(self.max_jitter_tries)
        self._dirty = False

    def _push_params(self):
        k = self.kern
        self._h.set_params(k._kernel_id, k.ARD, float(k.variance), k.lengthscale.values,
                           float(self.likelihood.variance))
        if k.Gower and k.space is not None:
            self._h.set_gower(*gower_config(k.space, k.input_dim))
        else:
            self._h.set_gower()

    def _device_group(self, devices):
        """The model replicated on ``devices`` (HIP ordinals; one may repeat) for scoring ONE candidate table on all of them
        from this single process (include/gphip.h, gp_group_*; run.py:1240-1241 over the GPUs of the node).  Created on first
        use, and brought in step -- data, hyper-parameters, Gower set-up, refit of every replica -- whenever the model has
        changed since the group last fitted."""
        key = tuple(int(d) for d in devices)
        grp = self._groups.get(key)
        if grp is None:
            grp = self._groups[key] = _lib.Group(key)
            grp._data_epoch = grp._signature = None
        k = self.kern
        gower = gower_config(k.space, k.input_dim) if (k.Gower and k.space is not None) else None
        signature = (self._data_epoch, k._kernel_id, bool(k.ARD), float(k.variance), tuple(np.ravel(k.lengthscale.values)),
                     float(self.likelihood.variance), None if gower is None else tuple(map(tuple, gower)))
        if grp._signature != signature:
            if grp._data_epoch != self._data_epoch:
                grp.set_data(self.X, self.Y_normalized)
                grp._data_epoch = self._data_epoch
            grp.set_params(k._kernel_id, k.ARD, float(k.variance), k.lengthscale.values, float(self.likelihood.variance))
            if gower is None:
                grp.set_gower()
            else:
                grp.set_gower(*gower)
            grp.fit(self.max_jitter_tries)
            grp._signature = signature
        return grp

    def _predict_resident(self, include_noise):
        """Mean / variance at the staged candidates.  When the model has to be (re)fitted first -- new data or new
        hyper-parameters, the state every BO iteration starts in (core/bo.py:236-254 then acquisitions/base.py:33-39)
        -- fit and predict go down as ONE call, gp_fit_predict (bitwise the results of the two calls)."""
        if self._dirty:
            self._push_params()
            (self._lml, self._logdet, self._jitter), mean, var = self._h.fit_predict(include_noise, self.max_jitter_tries)
            self._dirty = False
            return mean, var
        return self._h.predict(include_noise=include_noise)

    def parameters_changed(self):
        self._dirty = True
        self._ensure_fit()

    def log_likelihood(self):
        """gp.py:273-277."""
        self._ensure_fit()
        return self._lml

    def objective_function(self):
        """Model.objective_function, core/model.py:96-110 (no priors on this path)."""
        return -float(self.log_likelihood())

    def _log_likelihood_gradients_natural(self):
        nls = self.kern.lengthscale.size
        if self._dirty:
            # objective and gradients of a new parameter vector (every L-BFGS evaluation, core/model.py:96-127):
            # fit and the Ky^-1 solve go down as ONE call, gp_fit_grad (bitwise the results of the two calls)
            self._push_params()
            (self._lml, self._logdet, self._jitter), (dv, dl, dn) = self._h.fit_grad(nls, self.max_jitter_tries)
            self._dirty = False
        else:
            self._ensure_fit()
            dv, dl, dn = self._h.lml_grad(nls)
        if self._uses_gower() and self.gower_gradients != 'fork':
            # the device returns the fork's values; the LML's own: K = variance^D x (a product free of both parameters)
            dv, dl = dv * self.kern.input_dim, np.zeros_like(dl)
        self.kern.variance.gradient = np.atleast_1d(dv)
        self.kern.lengthscale.gradient = dl
        self.likelihood.variance.gradient = np.atleast_1d(dn)
        return [(self.kern.variance, dv), (self.kern.lengthscale, dl), (self.likelihood.variance, dn)]

    def objective_function_gradients(self):
        """Model.objective_function_gradients, core/model.py:112-127: d(-lml)/d(optimizer_array)."""
        return -self._transform_gradients(self._log_likelihood_gradients_natural())

    @property
    def gradient(self):
        g = self._log_likelihood_gradients_natural()
        return np.concatenate([np.atleast_1d(np.asarray(x, dtype=float)).reshape(-1) for _, x in g])

    # -- prediction ---------------------------------------------------------------------
    def _stage(self, Xnew, fit=True):
        Xnew = np.asarray(Xnew, dtype=float)
        if Xnew.ndim == 1:
            Xnew = Xnew[None, :]
        if fit:
            self._ensure_fit()
        self._h.set_candidates(Xnew)
        return Xnew

    def _few_rows(self, Xnew, limit=8):
        """``Xnew`` as a 2-D float array when it is a handful of locations of a fitted single-output model -- the calls an
        optimiser makes one location at a time: they go down as ONE gp_predict_rows call, locations by value -- else None."""
        Xnew = np.asarray(Xnew, dtype=float)
        if Xnew.ndim == 1:
            Xnew = Xnew[None, :]
        if self.output_dim != 1 or not (1 <= Xnew.shape[0] <= limit):
            return None
        if Xnew.shape[1] != self.input_dim:
            raise ValueError("candidates have %d columns, model has %d" % (Xnew.shape[1], self.input_dim))
        self._ensure_fit()
        return Xnew

    def _empty(self, Xnew, full_cov):
        """Zero prediction locations: the shapes NumPy gives the reference (posterior.py:273-302 on a (0, D) array)."""
        Xnew = np.asarray(Xnew, dtype=float)
        if Xnew.ndim == 2 and Xnew.shape[0] == 0:
            return np.empty((0, self.output_dim)), (np.empty((0, 0)) if full_cov else np.empty((0, 1)))
        return None

    def _raw_predict(self, Xnew, full_cov=False, kern=None):
        """gp.py:279-295 -> PosteriorExact._raw_predict (posterior.py:273-302)."""
        if kern is not None and kern is not self.kern:
            raise NotImplementedError("prediction with a foreign kernel is outside the accelerated path")
        e = self._empty(Xnew, full_cov)
        if e is not None:
            return e
        few = None if full_cov else self._few_rows(Xnew)
        if few is not None:
            return self._h.predict_rows(few, include_noise=False)
        self._stage(Xnew, fit=full_cov)
        if full_cov:
            return self._h.predict_full_cov(include_noise=False)
        return self._predict_resident(False)

    def predict(self, Xnew, full_cov=False, Y_metadata=None, kern=None, likelihood=None, include_likelihood=True):
        """gp.py:297-354."""
        if kern is not None and kern is not self.kern:
            raise NotImplementedError("prediction with a foreign kernel is outside the accelerated path")
        e = self._empty(Xnew, full_cov)
        if e is not None:
            return e
        few = None if full_cov else self._few_rows(Xnew)
        if few is not None:
            mean, var = self._h.predict_rows(few, include_noise=include_likelihood)
        else:
            self._stage(Xnew, fit=full_cov)
            if full_cov:
                mean, var = self._h.predict_full_cov(include_noise=include_likelihood)
            else:
                mean, var = self._predict_resident(include_likelihood)
        if self.normalizer is not None:
            mean = self.normalizer.inverse_mean(mean)
            if full_cov and mean.shape[1] > 1:
                var = self.normalizer.inverse_covariance(var)
            else:
                var = self.normalizer.inverse_variance(var)
        return mean, var

    def predict_noiseless(self, Xnew, full_cov=False, Y_metadata=None, kern=None):
        """gp.py:356-379."""
        return self.predict(Xnew, full_cov, Y_metadata, kern, None, False)

    def predict_quantiles(self, X, quantiles=(2.5, 97.5), Y_metadata=None, kern=None, likelihood=None):
        """gp.py:384-405: predictive quantiles around the prediction at X, one [Nnew, output_dim] array per quantile.
        The Gaussian likelihood's ``predictive_quantiles`` (gaussian.py:118-119) is
        ``norm.ppf(q / 100) * sqrt(var + noise) + mu`` on the raw (noise-free, normalised) posterior; the normaliser's
        ``inverse_mean`` is then applied to each quantile, exactly as the reference does (gp.py:403-404)."""
        if likelihood is not None and likelihood is not self.likelihood:
            raise NotImplementedError("a foreign likelihood is outside the accelerated path")
        from scipy import stats
        m, v = self._raw_predict(X, full_cov=False, kern=kern)
        noise = float(self.likelihood.variance)
        qs = [stats.norm.ppf(q / 100.) * np.sqrt(v + noise) + m for q in quantiles]
        if self.normalizer is not None:
            qs = [self.normalizer.inverse_mean(q) for q in qs]
        return qs

    def predictive_gradients(self, Xnew, kern=None):
        """gp.py:407-454: (dmu_dX [M, D, P], dv_dX [M, D])."""
        Xn = np.asarray(Xnew, dtype=float)
        if Xn.ndim == 2 and Xn.shape[0] == 0:
            return np.empty((0, self.input_dim, self.output_dim)), np.empty((0, self.input_dim))
        few = self._few_rows(Xnew)
        if few is not None:
            return self._h.predict_rows(few, grad=True)[2:]
        self._stage(Xnew)
        return self._h.predict_grad()

    def mean_gradients(self, Xnew):
        """d mean / dx [M, D, P] alone -- the first output of ``predictive_gradients`` (gp.py:433-438) without the work of
        the second: what ``estimate_L`` maximises over 500 + N points (batch_local_penalization.py:55-64)."""
        Xn = np.asarray(Xnew, dtype=float)
        if Xn.ndim == 2 and Xn.shape[0] == 0:
            return np.empty((0, self.input_dim, self.output_dim))
        few = self._few_rows(Xnew)
        if few is not None:
            return self._h.mean_grad_rows(few)
        self._stage(Xnew)
        return self._h.predict_grad(mean_only=True)

    def posterior_covariance_between_points(self, X1, X2):
        """gp.py:714-721 -> Posterior.covariance_between_points (posterior.py:109-128):
        K(X1, X2) - (L^-1 K(X, X1))^T (L^-1 K(X, X2)), in the model's (normalised) output space.  Served by the
        device's full-covariance path on the stacked points; the answer is its off-diagonal block."""
        X1 = np.atleast_2d(np.asarray(X1, dtype=float))
        X2 = np.atleast_2d(np.asarray(X2, dtype=float))
        _, C = self._raw_predict(np.vstack([X1, X2]), full_cov=True)
        return C[:X1.shape[0], X1.shape[0]:]

    def posterior_samples_f(self, X, size=10, normals=None, **kw):
        """gp.py:581-609: draws of the latent function at X, [Nnew, output_dim, size].

        The Nnew x Nnew posterior covariance (posterior.py:280-284) is built and Cholesky-factored on the device
        (gp_posterior_samples, GPy's jitter ladder) and applied to standard normals: ``normals`` [output_dim, size,
        Nnew] if given (reproducible draws), else ``np.random.standard_normal`` -- the global generator the
        reference's ``np.random.multivariate_normal`` consumes too (its SVD factor gives other draws of the same law)."""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        M, P = X.shape[0], self.output_dim
        if normals is None:
            normals = np.random.standard_normal((P, size, M))
        normals = np.asarray(normals, dtype=float).reshape(P, size, M)
        self._stage(X)
        # jitter of the M x M factorisation (jitchol's ladder on the posterior covariance: mean(diag) * 1e-6 * 10^k,
        # linalg.py:62-75; 0 when the first attempt succeeds) is kept for the caller: ``posterior_samples_jitter_``.
        # An exhausted ladder raises LinAlgError as jitchol does -- there is no host fallback.
        m, dev, self.posterior_samples_jitter_ = self._h.posterior_samples(normals.reshape(P * size, M), include_noise=False,
                                                                           maxtries=self.max_jitter_tries)
        dev = dev.reshape(P, size, M)
        if self.normalizer is not None:
            # inverse_mean / inverse_variance (normalizer.py:85-108): deviations scale with std, like sqrt(variance)
            m = self.normalizer.inverse_mean(m)
            dev = dev * np.asarray(self.normalizer.std, dtype=float).reshape(-1, 1, 1)
        fsim = np.empty((M, P, size))
        for d in range(P):
            fsim[:, d, :] = m[:, d:d + 1] + dev[d].T
        return fsim

    # -- optimisation -------------------------------------------------------------------
    def _uses_gower(self):
        return bool(self.kern.Gower and self.kern.space is not None)

    def _obj_grad(self, x):
        try:
            self.optimizer_array = x
            if not self._uses_gower() or self.gower_gradients in ('fork', 'exact'):
                g = self.objective_function_gradients()   # gp_fit_grad: leaves the LML of this x behind
                return self.objective_function(), g
            # 'differences': forward differences of the device LML (the Gower K paired with the fork's Euclidean gradient
            # formulas, stationary.py:218-238, is not differentiated by them)
            f = self.objective_function()
            g = np.empty_like(x)
            for i in range(x.size):
                e = np.zeros_like(x)
                e[i] = 1e-6
                self.optimizer_array = x + e
                g[i] = (self.objective_function() - f) / 1e-6
            self.optimizer_array = x
            return f, g
        except np.linalg.LinAlgError:
            return 1e10, np.zeros_like(x)  # paramz Model._objective_grads: failed evaluations are walls

    def optimize(self, optimizer=None, start=None, messages=False, max_iters=1000, ipython_notebook=False,
                 clear_after_finish=False, **kwargs):
        """GP.optimize (gp.py:643-664) -> scipy L-BFGS-B over the transformed parameters."""
        if optimizer not in (None, "lbfgs", "lbfgsb", "bfgs", "scg"):
            raise ValueError("unknown optimizer %r" % optimizer)
        x0 = self.optimizer_array if start is None else np.asarray(start, dtype=float)
        if x0.size == 0:
            return None
        # paramz' opt_lbfgsb spells the two stopping tolerances ``bfgs_factor`` (factr) and ``gtol`` (pgtol)
        extra = {}
        if kwargs.get("bfgs_factor") is not None:
            extra["factr"] = float(kwargs["bfgs_factor"])
        if kwargs.get("gtol") is not None:
            extra["pgtol"] = float(kwargs["gtol"])
        res = _sopt.fmin_l_bfgs_b(self._obj_grad, x0, maxiter=int(max_iters), maxfun=int(max_iters), **extra)
        xbest = res[0]
        self.optimizer_array = xbest
        self._ensure_fit()
        return res

    def checkgrad(self, verbose=False, step=1e-6, tolerance=1e-3):
        """paramz ``Model.checkgrad`` as the reference's tests call it (GPy/GPy/testing/model_tests.py:684-723,
        kernel_tests.py:414-422: ``assert m.checkgrad()``), restated from its published behaviour: in the optimiser's
        (transformed) space, the objective's central difference along ONE random sign vector of length ``step`` against the
        analytic gradient's projection on it; True when their ratio is within ``tolerance`` of 1 (or both vanish).
        ``verbose`` prints the per-parameter table instead and returns whether every parameter passes."""
        x = self.optimizer_array.copy()
        if x.size == 0:
            return True
        try:
            if not verbose:
                dx = step * np.sign(np.random.uniform(-1, 1, x.size))
                f1 = self._obj_grad(x + dx)[0]
                f2 = self._obj_grad(x - dx)[0]
                grad = self._obj_grad(x)[1]
                denominator = 2.0 * np.dot(dx, grad)
                ratio = (f1 - f2) / (denominator if denominator != 0.0 else 1e-32)
                both_flat = abs(f1 - f2) < tolerance and np.allclose(grad, 0, atol=tolerance)
                return bool(abs(1.0 - ratio) < tolerance or both_flat)
            grad = self._obj_grad(x)[1]
            names = self.parameter_names_flat()
            ok = True
            print("%-32s | %12s | %12s | %12s" % ("Name", "Ratio", "Analytical", "Numerical"))
            for i in range(x.size):
                e = np.zeros_like(x)
                e[i] = step
                num = (self._obj_grad(x + e)[0] - self._obj_grad(x - e)[0]) / (2.0 * step)
                ratio = num / grad[i] if grad[i] != 0.0 else np.inf
                good = abs(1.0 - ratio) < tolerance or abs(num - grad[i]) < tolerance
                ok = ok and good
                print("%-32s | %12.6f | %12.6f | %12.6f%s" % (names[i] if i < len(names) else i, ratio, grad[i], num,
                                                           "" if good else "   <-- FAIL"))
            return bool(ok)
        finally:
            self.optimizer_array = x

    def randomize(self):
        """paramz Parameterized.randomize: N(0,1) draws in the optimiser space."""
        x = np.random.normal(size=self.optimizer_array.size)
        self.optimizer_array = x

    def optimize_restarts(self, num_restarts=10, robust=False, verbose=True, parallel=False, num_processes=None,
                          **kwargs):
        """paramz Model.optimize_restarts: keep the best of ``num_restarts`` L-BFGS runs."""
        initial = self.optimizer_array.copy()
        runs = []
        for i in range(num_restarts):
            try:
                if i > 0:
                    self.randomize()
                self.optimize(**kwargs)
                runs.append((self.objective_function(), self.optimizer_array.copy()))
                if verbose:
                    print("Optimization restart %d/%d, f = %s" % (i + 1, num_restarts, runs[-1][0]))
            except Exception as e:  # noqa: BLE001  (robust mode of the reference swallows failures)
                if robust:
                    if verbose:
                        print("Warning - optimization restart %d/%d failed: %s" % (i + 1, num_restarts, e))
                else:
                    raise
        if runs:
            best = min(runs, key=lambda t: t[0])
            self.optimizer_array = best[1]
        else:
            self.optimizer_array = initial
        self._ensure_fit()
        return runs

    # -- misc ---------------------------------------------------------------------------
    def __str__(self):
        self_lml = "?" if self._dirty else "%.6f" % self._lml
        rows = ["Name : %s" % self.name, "Objective : -%s" % self_lml]
        for p in self.flattened_parameters():
            rows.append("  %s" % p)
        return "\n".join(rows)

    def close(self):
        for grp in self._groups.values():
            grp.close()
        self._groups = {}
        self._h.close()

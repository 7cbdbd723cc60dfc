"""``GPModel`` -- host mirror of ``GPyOpt.models.GPModel`` on top of the HIP ``GPRegression``.

Reference: GPyOpt/GPyOpt/models/base.py:7-33 (BOModel contract),
GPyOpt/GPyOpt/models/gpmodel.py:9-177 (GPModel).  Same constructor keywords, same
return conventions: ``predict`` returns (mean, **std**) with the variance clipped at
1e-10 (:95-112), ``get_fmin`` is the minimum posterior mean over the training inputs
(:125-129; cached per fit on the device -- the reference recomputes the identical
value on every acquisition call), ``predict_withGradients`` (:131-142).
"""
import numpy as np

from . import kern as _kern
from .gp_regression import GPRegression


_VAR_FLOOR = 1e-10      # the reference clips predictive variances here before taking the root (gpmodel.py:99)


class BOModel(object):
    """What a surrogate offers the BO loop (GPyOpt/GPyOpt/models/base.py:7-33): subclasses fill these four in."""
    MCMC_sampler = False
    analytical_gradient_prediction = False

    def updateModel(self, X_all, Y_all, X_new, Y_new):
        """Take the data set (``X_all``, ``Y_all``) over and re-estimate the hyper-parameters."""

    def predict(self, X):
        """(mean, standard deviation) at the rows of ``X``."""

    def predict_withGradients(self, X):
        """(mean, standard deviation, d mean / dx, d std / dx)."""

    def get_fmin(self):
        """Smallest posterior mean over the training inputs."""


class GPModel(BOModel):
    """Exact GP surrogate on the device.  Keyword arguments, attributes and return conventions are GPyOpt's
    (gpmodel.py:9-177); ``device`` (HIP ordinal) is the one addition."""
    analytical_gradient_prediction = True

    def __init__(self, kernel=None, noise_var=None, exact_feval=False, optimizer='bfgs', max_iters=1000,
                 optimize_restarts=5, sparse=False, num_inducing=10, verbose=True, ARD=False, Gower=False,
                 space=None, device=0):
        if sparse:
            raise NotImplementedError("sparse GP is a different model family (out of scope)")
        vars(self).update(kernel=kernel, noise_var=noise_var, exact_feval=exact_feval, optimizer=optimizer,
                          max_iters=max_iters, optimize_restarts=optimize_restarts, sparse=sparse,
                          num_inducing=num_inducing, verbose=verbose, ARD=ARD, Gower=Gower, space=space, device=device,
                          model=None)

    @staticmethod
    def fromConfig(config):
        return GPModel(**config)

    def _gower(self):
        return bool(self.Gower and self.space is not None)

    def _create_model(self, X, Y):
        """First data: build the GP.  Default kernel Matern-5/2 (with the fork's Gower option passed through); default noise
        1 % of Var(Y); ``exact_feval`` pins the noise at 1e-6, otherwise it is kept inside [1e-9, 1e6] (gpmodel.py:50-76)."""
        self.input_dim = X.shape[1]
        chosen, self.kernel = self.kernel, None      # a user kernel is consumed by the first model, as in the reference
        if chosen is None:
            chosen = _kern.Matern52(self.input_dim, variance=1., ARD=self.ARD, Gower=self.Gower, space=self.space)
        noise = 0.01 * Y.var() if self.noise_var is None else self.noise_var
        gp = GPRegression(X, Y, kernel=chosen, noise_var=noise, device=self.device)
        if self.exact_feval:
            gp.Gaussian_noise.constrain_fixed(1e-6, warning=False)
        else:
            gp.Gaussian_noise.constrain_bounded(1e-9, 1e6, warning=False)
        self.model = gp

    def updateModel(self, X_all, Y_all, X_new, Y_new):
        """New data in, then the hyper-parameter search: one L-BFGS run, or ``optimize_restarts`` of them (gpmodel.py:78-93);
        ``max_iters = 0`` keeps the hyper-parameters."""
        if self.model is not None:
            self.model.set_XY(X_all, Y_all)
        else:
            self._create_model(X_all, Y_all)
        if self.max_iters <= 0:
            return
        search = dict(optimizer=self.optimizer, max_iters=self.max_iters)
        if self.optimize_restarts == 1:
            self.model.optimize(messages=False, ipython_notebook=False, **search)
        else:
            self.model.optimize_restarts(num_restarts=self.optimize_restarts, verbose=self.verbose, **search)

    def _predict(self, X, full_cov, include_likelihood):
        mean, var = self.model.predict(np.atleast_2d(X), full_cov=full_cov, include_likelihood=include_likelihood)
        return mean, np.maximum(var, _VAR_FLOOR)

    def predict(self, X, with_noise=True):
        mean, var = self._predict(X, False, with_noise)
        return mean, np.sqrt(var)

    def predict_covariance(self, X, with_noise=True):
        return self._predict(X, True, with_noise)[1]

    def get_fmin(self):
        """``self.model.predict(self.model.X)[0].min()`` (gpmodel.py:125-129), evaluated on the device and cached per fit."""
        gp = self.model
        gp._ensure_fit()
        lowest = gp._h.fmin()
        if gp.normalizer is not None:
            lowest = float(gp.normalizer.inverse_mean(np.array([[lowest]]))[0, 0])
        return lowest

    def predict_withGradients(self, X):
        """Mean, std and their input gradients (gpmodel.py:131-142); with a normaliser the gradients are those of the
        un-normalised mean / variance."""
        X = np.atleast_2d(X)
        gp = self.model
        few = gp._few_rows(X)
        if few is not None:     # posterior and gradients of a handful of locations in ONE device call (gp_predict_rows)
            mean, var, jac_mean, jac_var = gp._h.predict_rows(few, include_noise=True, grad=True)
            if gp.normalizer is not None:
                mean, var = gp.normalizer.inverse_mean(mean), gp.normalizer.inverse_variance(var)
        else:
            mean, var = gp.predict(X)
            jac_mean, jac_var = gp.predictive_gradients(X)
        std = np.sqrt(np.maximum(var, _VAR_FLOOR))
        jac_mean = jac_mean[..., 0]
        if gp.normalizer is not None:
            jac_mean = jac_mean * gp.normalizer.std
            jac_var = jac_var * gp.normalizer.std ** 2
        return mean, std, jac_mean, jac_var / (2 * std)

    def copy(self):
        twin = GPModel(kernel=self.model.kern.copy(), noise_var=self.noise_var, exact_feval=self.exact_feval,
                       optimizer=self.optimizer, max_iters=self.max_iters, optimize_restarts=self.optimize_restarts,
                       verbose=self.verbose, ARD=self.ARD, Gower=self.Gower, space=self.space, device=self.device)
        twin._create_model(self.model.X, self.model.Y)
        twin.updateModel(self.model.X, self.model.Y, None, None)
        return twin

    def get_model_parameters(self):
        return np.atleast_2d(self.model[:])

    def get_model_parameters_names(self):
        return self.model.parameter_names_flat().tolist()

    def get_covariance_between_points(self, x1, x2):
        """Posterior covariance between two point sets (gpmodel.py:173-177)."""
        return self.model.posterior_covariance_between_points(x1, x2)

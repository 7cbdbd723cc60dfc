"""Acquisition functions mirroring GPyOpt's (EI / LCB / MPI) with device-side batched scoring.

Reference: GPyOpt/GPyOpt/acquisitions/base.py:7-68 (AcquisitionBase: ``acquisition_function``
returns the NEGATED value weighted by constraints and cost), EI.py:7-51, LCB.py:6-46,
MPI.py:7-51, GPyOpt/GPyOpt/util/general.py:113-129 (get_quantiles).

``_compute_acq`` / ``_compute_acq_withGradients`` keep the reference's host formulas on top of
``model.predict`` (so any BOModel works); when the model is the HIP ``GPModel`` and there are no
constraints or cost, ``acquisition_function`` and ``argbest`` score the whole candidate table
inside libgphip (gp_acq / gp_acq_argbest) -- the batched call pattern of
GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59,96-98 and run.py:1240-1241.
"""
import numpy as np
from scipy.special import erfc

from . import _lib
from .gpmodel import GPModel


def get_quantiles(acquisition_par, fmin, m, s):
    """GPyOpt/GPyOpt/util/general.py:113-129."""
    if isinstance(s, np.ndarray):
        s[s < 1e-10] = 1e-10
    elif s < 1e-10:
        s = 1e-10
    u = (fmin - m - acquisition_par) / s
    phi = np.exp(-0.5 * u ** 2) / np.sqrt(2 * np.pi)
    Phi = 0.5 * erfc(-u / np.sqrt(2))
    return (phi, Phi, u)


def constant_cost_withGradients(x):
    """GPyOpt/GPyOpt/core/task/cost.py:76."""
    return np.ones(x.shape[0])[:, None], np.zeros(x.shape)


class _NoConstraints(object):
    def indicator_constraints(self, x):
        return np.ones((np.atleast_2d(x).shape[0], 1))

    def has_constraints(self):
        return False


class AcquisitionBase(object):
    """acquisitions/base.py:7-68."""
    analytical_gradient_prediction = False
    _acq_id = None

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None):
        self.model = model
        self.space = space if space is not None else _NoConstraints()
        self.optimizer = optimizer
        self.analytical_gradient_acq = self.analytical_gradient_prediction and self.model.analytical_gradient_prediction
        self.cost_withGradients = constant_cost_withGradients if cost_withGradients is None else cost_withGradients

    # -- device fast path ---------------------------------------------------------------
    def _device_ok(self):
        if not isinstance(self.model, GPModel) or self._acq_id is None or self.model.model is None:
            return False
        if self.cost_withGradients is not constant_cost_withGradients:
            return False
        has = getattr(self.space, "has_constraints", None)
        if has is not None and has():
            return False
        return self.model.model.output_dim == 1

    def _par(self):
        raise NotImplementedError

    def _device_stage(self, x):
        gp = self.model.model
        x = np.atleast_2d(np.asarray(x, dtype=float))
        if gp._dirty:
            # a refit is pending (new data / hyper-parameters): fit and the posterior at x go down as one call
            gp._stage(x, fit=False)
            gp._predict_resident(True)   # GPModel.predict: with_noise=True (gpmodel.py:102); kept resident on the device
        else:
            gp._stage(x)
        nz = gp.normalizer
        y_mean = float(nz.mean[0]) if nz is not None else 0.0
        y_std = float(nz.std[0]) if nz is not None else 1.0
        fmin = self.model.get_fmin() if self._acq_id != _lib.GP_ACQ_LCB else 0.0
        return gp, fmin, y_mean, y_std

    def acquisition_function(self, x):
        """base.py:33-39: -(acq * indicator_constraints) / cost."""
        if self._device_ok():
            gp, fmin, y_mean, y_std = self._device_stage(x)
            return gp._h.acq(self._acq_id, self._par(), fmin, y_mean, y_std)
        f_acqu = self._compute_acq(x)
        cost_x, _ = self.cost_withGradients(x)
        return -(f_acqu * self.space.indicator_constraints(x)) / cost_x

    def acquisition_function_withGradients(self, x):
        """base.py:42-50."""
        f_acqu, df_acqu = self._compute_acq_withGradients(x)
        cost_x, cost_grad_x = self.cost_withGradients(x)
        f_acq_cost = f_acqu / cost_x
        df_acq_cost = (df_acqu * cost_x - f_acqu * cost_grad_x) / (cost_x ** 2)
        ind = self.space.indicator_constraints(x)
        return -f_acq_cost * ind, -df_acq_cost * ind

    def argbest(self, x, sense=-1):
        """Index and value of the best row of ``acquisition_function(x)``.

        sense=-1: the smallest (GPyOpt's convention, anchor_points_generator.py:61);
        sense=+1: the largest (run.py:1241 takes ``np.argmax``).  Ties -> lowest index.
        """
        if self._device_ok():
            gp, fmin, y_mean, y_std = self._device_stage(x)
            return gp._h.acq_argbest(self._acq_id, self._par(), fmin, sense, y_mean, y_std)
        a = self.acquisition_function(x)[:, 0]
        i = int(np.argmin(a) if sense < 0 else np.argmax(a))
        return i, float(a[i])

    def topk(self, x, k, sense=-1):
        """The ``k`` best rows of ``acquisition_function(x)`` in order, (indices, values): what
        ``AnchorPointsGenerator.get`` keeps (anchor_points_generator.py:59-61, ``argsort(scores)[:num_anchor]``).
        Equal scores come out lowest index first; fewer than ``k`` rows -> the tail is index -1."""
        if self._device_ok() and k <= 64:
            gp, fmin, y_mean, y_std = self._device_stage(x)
            return gp._h.acq_topk(self._acq_id, self._par(), fmin, sense, k, y_mean, y_std)
        a = self.acquisition_function(x)[:, 0]
        order = np.argsort(a if sense < 0 else -a, kind="stable")[:k]
        idx = np.full(k, -1, dtype=np.int64)
        val = np.full(k, np.inf if sense < 0 else -np.inf)
        idx[:order.size], val[:order.size] = order, a[order]
        return idx, val

    def optimize(self, duplicate_manager=None):
        """base.py:52-60."""
        if not self.analytical_gradient_acq:
            return self.optimizer.optimize(f=self.acquisition_function, duplicate_manager=duplicate_manager)
        return self.optimizer.optimize(f=self.acquisition_function, f_df=self.acquisition_function_withGradients,
                                       duplicate_manager=duplicate_manager)

    def _compute_acq(self, x):
        raise NotImplementedError('')

    def _compute_acq_withGradients(self, x):
        raise NotImplementedError('')


class AcquisitionEI(AcquisitionBase):
    """Expected improvement, EI.py:7-51."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_EI

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        self.optimizer = optimizer
        super(AcquisitionEI, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionEI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        fmin = self.model.get_fmin()
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        return s * (u * Phi + phi)

    def _compute_acq_withGradients(self, x):
        fmin = self.model.get_fmin()
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        f_acqu = s * (u * Phi + phi)
        df_acqu = dsdx * phi - Phi * dmdx
        return f_acqu, df_acqu


class AcquisitionLCB(AcquisitionBase):
    """GP lower confidence bound, LCB.py:6-46."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_LCB

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, exploration_weight=2):
        self.optimizer = optimizer
        super(AcquisitionLCB, self).__init__(model, space, optimizer)
        self.exploration_weight = exploration_weight
        if cost_withGradients is not None:
            print('The set cost function is ignored! LCB acquisition does not make sense with cost.')

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionLCB(model, space, optimizer, cost_withGradients, exploration_weight=config['weight'])

    def _par(self):
        return self.exploration_weight

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        return -m + self.exploration_weight * s

    def _compute_acq_withGradients(self, x):
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        return -m + self.exploration_weight * s, -dmdx + self.exploration_weight * dsdx


class AcquisitionMPI(AcquisitionBase):
    """Maximum probability of improvement, MPI.py:7-51."""
    analytical_gradient_prediction = True
    _acq_id = _lib.GP_ACQ_MPI

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        self.optimizer = optimizer
        super(AcquisitionMPI, self).__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionMPI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter

    def _compute_acq(self, x):
        m, s = self.model.predict(x)
        fmin = self.model.get_fmin()
        _, Phi, _ = get_quantiles(self.jitter, fmin, m, s)
        return Phi

    def _compute_acq_withGradients(self, x):
        fmin = self.model.get_fmin()
        m, s, dmdx, dsdx = self.model.predict_withGradients(x)
        phi, Phi, u = get_quantiles(self.jitter, fmin, m, s)
        return Phi, -(phi / s) * (dmdx + dsdx * u)


class AcquisitionLP(AcquisitionBase):
    """Local-penalisation acquisition for batch design, GPyOpt/GPyOpt/acquisitions/LP.py:10-140.

    Always in log space: ``-T(acq(x)) - sum_k log Phi((|x - x0_k| - r_k) / s_k)``.  On the HIP path the whole
    candidate table is scored on the device (gp_acq_lp: the base EI / LCB / MPI kernel plus the hammer-function
    epilogue); the reference's NumPy formulas remain as the path for foreign models and for gradients.
    """
    analytical_gradient_prediction = True

    def __init__(self, model, space=None, optimizer=None, acquisition=None, transform='none'):
        super(AcquisitionLP, self).__init__(model, space, optimizer)
        self.acq = acquisition
        self.transform = transform.lower()
        if isinstance(acquisition, AcquisitionLCB) and self.transform == 'none':
            self.transform = 'softplus'                      # LP.py:32-35
        self.X_batch = None
        self.r_x0 = None
        self.s_x0 = None

    def update_batches(self, X_batch, L, Min):
        """LP.py:41-47."""
        self.X_batch = X_batch
        if X_batch is not None:
            self.r_x0, self.s_x0 = self._hammer_function_precompute(X_batch, L, Min, self.model)

    def _hammer_function_precompute(self, x0, L, Min, model):
        """LP.py:49-62 (the reference feeds the model's *std* into ``pred`` and takes its square root again)."""
        if x0 is None:
            return None, None
        if len(x0.shape) == 1:
            x0 = x0[None, :]
        m = model.predict(x0)[0]
        pred = model.predict(x0)[1].copy()
        pred[pred < 1e-16] = 1e-16
        s = np.sqrt(pred)
        r_x0 = (m - Min) / L
        s_x0 = s / L
        return r_x0.flatten(), s_x0.flatten()

    def _hammer_function(self, x, x0, r_x0, s_x0):
        """LP.py:64-68."""
        from scipy.stats import norm
        return norm.logcdf((np.sqrt((np.square(np.atleast_2d(x)[:, None, :] - np.atleast_2d(x0)[None, :, :])).sum(-1))
                            - r_x0) / s_x0)

    def _penalized_acquisition(self, x, model, X_batch, r_x0, s_x0):
        """LP.py:70-89 (host formulas)."""
        fval = -self.acq.acquisition_function(x)[:, 0]
        if self.transform == 'softplus':
            fval_org = fval.copy()
            fval[fval_org >= 40.] = np.log(fval_org[fval_org >= 40.])
            fval[fval_org < 40.] = np.log(np.log1p(np.exp(fval_org[fval_org < 40.])))
        elif self.transform == 'none':
            fval = np.log(fval + 1e-50)
        fval = -fval
        if X_batch is not None:
            h_vals = self._hammer_function(x, X_batch, r_x0, s_x0)
            fval += -h_vals.sum(axis=-1)
        return fval

    def _d_hammer_function(self, x, X_batch, r_x0, s_x0):
        """LP.py:91-103."""
        from scipy.stats import norm
        dx = np.atleast_2d(x)[:, None, :] - np.atleast_2d(X_batch)[None, :, :]
        nm = np.sqrt((np.square(dx)).sum(-1))
        z = (nm - r_x0) / s_x0
        h_func = norm.cdf(z)
        d = 1. / (s_x0 * np.sqrt(2 * np.pi) * h_func) * np.exp(-np.square(z) / 2) / nm
        d[h_func < 1e-50] = 0.
        d = d[:, :, None]
        return d.sum(axis=1)

    # -- device fast path -----------------------------------------------------------------
    def _lp_device_ok(self):
        return (self.acq is not None and self.acq.model is self.model and self.acq._device_ok()
                and self.transform in ('none', 'softplus'))

    def _lp_device_args(self, x):
        gp, fmin, y_mean, y_std = self.acq._device_stage(x)
        tr = 1 if self.transform == 'softplus' else 0
        return gp, (self.acq._acq_id, self.acq._par(), fmin, tr), dict(Xb=self.X_batch, r_x0=self.r_x0, s_x0=self.s_x0,
                                                                      y_mean=y_mean, y_std=y_std)

    def acquisition_function(self, x):
        """LP.py:105-110.  Returns a 1-D array like the reference."""
        if self._lp_device_ok():
            gp, a, kw = self._lp_device_args(x)
            return gp._h.acq_lp(*a, **kw)
        return self._penalized_acquisition(x, self.model, self.X_batch, self.r_x0, self.s_x0)

    def argbest(self, x, sense=+1, exclude=()):
        """Arg-best of ``acquisition_function(x)`` with already-chosen rows masked (run.py:1241,1249-1252)."""
        if self._lp_device_ok():
            gp, a, kw = self._lp_device_args(x)
            return gp._h.acq_lp_argbest(a[0], a[1], a[2], a[3], sense, exclude=exclude, **kw)
        v = np.ma.array(self.acquisition_function(x), mask=False)
        for e in exclude:
            v.mask[e] = True
        i = int(np.argmax(v) if sense > 0 else np.argmin(v))
        return i, float(v[i])

    def d_acquisition_function(self, x):
        """LP.py:112-133."""
        x = np.atleast_2d(x)
        if self.transform == 'softplus':
            fval = -self.acq.acquisition_function(x)[:, 0]
            scale = 1. / (np.log1p(np.exp(fval)) * (1. + np.exp(-fval)))
        elif self.transform == 'none':
            fval = -self.acq.acquisition_function(x)[:, 0]
            scale = 1. / fval
        else:
            scale = 1.
        # the reference multiplies a length-M vector with an [M, D] gradient, which only broadcasts for the
        # single-row calls L-BFGS makes; a column keeps those values and also serves M > 1
        scale = np.atleast_1d(scale)[:, None] if np.ndim(scale) else scale
        _, grad_acq_x = self.acq.acquisition_function_withGradients(x)
        if self.X_batch is None:
            return scale * grad_acq_x
        return scale * grad_acq_x - self._d_hammer_function(x, self.X_batch, self.r_x0, self.s_x0)

    def acquisition_function_withGradients(self, x):
        """LP.py:135-140."""
        return self.acquisition_function(x), self.d_acquisition_function(x)


def estimate_L(model, bounds, storehistory=True):
    """Lipschitz constant of the posterior mean, GPyOpt/GPyOpt/core/evaluators/batch_local_penalization.py:52-70.

    ``model`` is the GP (``GPModel.model``); its predictive gradients over the 500 + N sample points come from
    one batched device call."""
    from scipy import optimize as _sopt

    def df(x, model, x0):
        x = np.atleast_2d(x)
        dmdx, _ = model.predictive_gradients(x)
        res = np.sqrt((dmdx * dmdx).sum(1))
        return -res

    bounds = list(bounds)
    lo = np.array([b[0] for b in bounds], dtype=float)
    hi = np.array([b[1] for b in bounds], dtype=float)
    samples = np.random.uniform(size=(500, len(bounds))) * (hi - lo) + lo     # samples_multidimensional_uniform
    samples = np.vstack([samples, model.X])
    pred_samples = df(samples, model, 0)
    x0 = samples[np.argmin(pred_samples)]
    res = _sopt.minimize(lambda x: float(df(x, model, x0).ravel()[0]), x0, method='L-BFGS-B', bounds=bounds,
                         options={'maxiter': 200})
    L = -float(res.fun)
    if L < 1e-7:
        L = 10  # flat model
    return L


class LocalPenalization(object):
    """Batch evaluator of Gonzalez et al. 2016, core/evaluators/batch_local_penalization.py:7-49."""

    def __init__(self, acquisition, batch_size):
        self.acquisition = acquisition
        self.batch_size = batch_size

    def compute_batch(self, duplicate_manager=None, context_manager=None):
        assert isinstance(self.acquisition, AcquisitionLP)
        self.acquisition.update_batches(None, None, None)
        X_batch = self.acquisition.optimize()[0]
        k = 1
        if self.batch_size > 1:
            L = estimate_L(self.acquisition.model.model, self.acquisition.space.get_bounds())
            Min = self.acquisition.model.model.Y.min()
        while k < self.batch_size:
            self.acquisition.update_batches(X_batch, L, Min)
            new_sample = self.acquisition.optimize()[0]
            X_batch = np.vstack((X_batch, new_sample))
            k += 1
        self.acquisition.update_batches(None, None, None)
        return X_batch

    def compute_batch_from_table(self, table, sense=+1):
        """The candidate-table variant the thesis driver uses (run.py:1234-1258): pick ``batch_size`` rows of
        ``table`` by repeated arg-best of the penalised acquisition, never the same row twice."""
        acq = self.acquisition
        acq.update_batches(None, None, None)
        i, _ = acq.argbest(table, sense)
        chosen = [i]
        X_batch = table[i]
        if self.batch_size > 1:
            L = estimate_L(acq.model.model, acq.space.get_bounds())
            Min = acq.model.model.Y.min()
        while len(chosen) < self.batch_size:
            acq.update_batches(np.atleast_2d(X_batch), L, Min)
            i, _ = acq.argbest(table, sense, exclude=chosen)
            chosen.append(i)
            X_batch = np.vstack((X_batch, table[i]))
        acq.update_batches(None, None, None)
        return chosen

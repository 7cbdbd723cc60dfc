"""EI / LCB / MPI and the local-penalisation batch acquisition, scored on the device.

Contract mirrored (names, arguments, return shapes and signs): GPyOpt/GPyOpt/acquisitions/base.py:7-68
(``acquisition_function`` returns the NEGATED value weighted by constraints and cost), EI.py:7-51, LCB.py:6-46, MPI.py:7-51,
LP.py:10-140, GPyOpt/GPyOpt/core/evaluators/batch_local_penalization.py:7-70, GPyOpt/GPyOpt/util/general.py:113-129.

Where the work runs.  With the HIP ``GPModel``, one output column, constant cost and no constraints, EVERY entry point goes to
libgphip on the whole candidate block at once: values (``gp_acq``), values + x-gradients (``gp_acq_grad``), arg-best / top-k
(``gp_acq_argbest`` / ``gp_acq_topk``), the penalised batch acquisition and its gradient (``gp_acq_lp`` / ``gp_acq_lp_grad``).
Anything else (a foreign ``BOModel``, a cost model, string constraints, several outputs) is scored by ``_Rule`` below from
``model.predict[_withGradients]``: each acquisition is one rule giving its value and its two partial derivatives with respect
to the posterior mean and standard deviation, and every class shares the chain rule built on them.
"""
import numpy as np
from scipy.special import erfc, log_ndtr, ndtr

from . import _lib
from .gpmodel import GPModel

_FEW_ROWS = 8            # up to this many locations go down as ONE gp_*_rows call (include/gphip.h): the L-BFGS inner loop
_ROOT_2 = np.sqrt(2)
_ROOT_2PI = np.sqrt(2 * np.pi)
_STD_FLOOR = 1e-10


def get_quantiles(acquisition_par, fmin, m, s):
    """Standardised improvement ``z = (fmin - m - par) / s`` with its normal density and distribution value, returned as
    ``(pdf, cdf, z)``; ``s`` is floored at 1e-10 (in place for arrays).  Same contract as general.py:113-129."""
    if isinstance(s, np.ndarray):
        np.maximum(s, _STD_FLOOR, out=s)
    else:
        s = max(s, _STD_FLOOR)
    z = (fmin - m - acquisition_par) / s
    pdf = np.exp(-0.5 * z ** 2) / _ROOT_2PI
    cdf = 0.5 * erfc(-z / _ROOT_2)
    return pdf, cdf, z


def constant_cost_withGradients(x):
    """Unit cost and zero cost gradient (core/task/cost.py:76)."""
    rows = x.shape[0]
    return np.ones((rows, 1)), np.zeros(x.shape)


class _Unconstrained(object):
    """Stand-in design space when none is given: every point feasible."""

    def indicator_constraints(self, x):
        return np.ones((np.atleast_2d(x).shape[0], 1))

    def has_constraints(self):
        return False


class _Rule(object):
    """One acquisition as a function of the posterior mean ``mu`` and standard deviation ``sd`` (column vectors):
    ``terms`` returns (value, d value / d mu, d value / d sd); the x-gradient follows by the chain rule in ``gradient``."""

    def __init__(self, device_id, needs_fmin):
        self.device_id = device_id
        self.needs_fmin = needs_fmin

    def terms(self, par, fmin, mu, sd):
        raise NotImplementedError

    def value(self, par, fmin, mu, sd):
        return self.terms(par, fmin, mu, sd)[0]

    def gradient(self, par, fmin, mu, sd, dmu_dx, dsd_dx):
        val, wrt_mu, wrt_sd = self.terms(par, fmin, mu, sd)
        return val, wrt_mu * dmu_dx + wrt_sd * dsd_dx


class _ExpectedImprovement(_Rule):
    def terms(self, par, fmin, mu, sd):
        pdf, cdf, z = get_quantiles(par, fmin, mu, sd)
        return sd * (z * cdf + pdf), -cdf, pdf


class _LowerConfidenceBound(_Rule):
    def terms(self, par, fmin, mu, sd):
        return -mu + par * sd, -1.0, par


class _ProbabilityOfImprovement(_Rule):
    def terms(self, par, fmin, mu, sd):
        pdf, cdf, z = get_quantiles(par, fmin, mu, sd)
        slope = -(pdf / sd)
        return cdf, slope, slope * z


_RULES = {"EI": _ExpectedImprovement(_lib.GP_ACQ_EI, True),
          "LCB": _LowerConfidenceBound(_lib.GP_ACQ_LCB, False),
          "MPI": _ProbabilityOfImprovement(_lib.GP_ACQ_MPI, True)}


def _pick(values, sense):
    """Lowest-index arg-best of a 1-D score vector (NumPy argmin / argmax semantics)."""
    i = int(np.argmin(values) if sense < 0 else np.argmax(values))
    return i, float(values[i])


class AcquisitionBase(object):
    """base.py:7-68.  Subclasses either name a ``_rule`` (EI / LCB / MPI) or override ``_compute_acq`` /
    ``_compute_acq_withGradients`` as in GPyOpt."""
    analytical_gradient_prediction = False
    _rule = None

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None):
        self.model = model
        self.space = _Unconstrained() if space is None else space
        self.optimizer = optimizer
        self.analytical_gradient_acq = self.analytical_gradient_prediction and self.model.analytical_gradient_prediction
        self.cost_withGradients = cost_withGradients or constant_cost_withGradients

    # ---- what the rule needs ---------------------------------------------------------------------------------------
    @property
    def _acq_id(self):
        return None if self._rule is None else self._rule.device_id

    def _par(self):
        raise NotImplementedError

    def _fmin(self):
        return self.model.get_fmin() if self._rule.needs_fmin else 0.0

    # ---- device route ----------------------------------------------------------------------------------------------
    def _device_ok(self):
        """True when libgphip can score this acquisition by itself: our GPModel, one output, unit cost, no constraints."""
        if self._rule is None or not isinstance(self.model, GPModel) or self.model.model is None:
            return False
        if self.cost_withGradients is not constant_cost_withGradients:
            return False
        constrained = getattr(self.space, "has_constraints", None)
        if constrained is not None and constrained():
            return False
        return self.model.model.output_dim == 1

    def _device_stage(self, x):
        """Make ``x`` the resident candidate block (refitting first if data or hyper-parameters changed) and return what
        every device scoring call takes: the handle owner, fmin and the normaliser's mean / std."""
        gp = self.model.model
        x = np.atleast_2d(np.asarray(x, dtype=float))
        if gp._dirty:
            gp._stage(x, fit=False)        # pending refit: fit + posterior at x go down as ONE call
            gp._predict_resident(True)     # GPModel.predict: with_noise=True (gpmodel.py:102)
        else:
            gp._stage(x)
        nz = gp.normalizer
        shift, scale = (0.0, 1.0) if nz is None else (float(nz.mean[0]), float(nz.std[0]))
        return gp, self._fmin(), shift, scale

    def _device_few(self, x):
        """The handful-of-locations route (what scipy's L-BFGS-B issues, optimizer.py:36-61): ``x`` as a 2-D float array when it
        has at most ``_FEW_ROWS`` rows -- the caller then makes ONE gp_acq_rows call, locations by value -- else None.
        Also returns what that call takes: the handle owner, fmin and the normaliser's mean / std."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        if x.shape[0] > _FEW_ROWS:
            return None
        gp = self.model.model
        gp._ensure_fit()
        nz = gp.normalizer
        shift, scale = (0.0, 1.0) if nz is None else (float(nz.mean[0]), float(nz.std[0]))
        return x, gp, self._fmin(), shift, scale

    # ---- the contract ----------------------------------------------------------------------------------------------
    def acquisition_function(self, x):
        """-(acq(x) * feasibility(x)) / cost(x), [M, 1]  (base.py:33-39)."""
        if self._device_ok():
            few = self._device_few(x)
            if few is not None:
                x, gp, fmin, shift, scale = few
                return gp._h.acq_rows(x, self._acq_id, self._par(), fmin, shift, scale)
            gp, fmin, shift, scale = self._device_stage(x)
            return gp._h.acq(self._acq_id, self._par(), fmin, shift, scale)
        price, _ = self.cost_withGradients(x)
        return -(self._compute_acq(x) * self.space.indicator_constraints(x)) / price

    def acquisition_function_withGradients(self, x):
        """The same value and its x-gradient [M, D]  (base.py:42-50)."""
        if self._device_ok():
            few = self._device_few(x)
            if few is not None:
                x, gp, fmin, shift, scale = few
                return gp._h.acq_rows(x, self._acq_id, self._par(), fmin, shift, scale, grad=True)
            gp, fmin, shift, scale = self._device_stage(x)
            return gp._h.acq_grad(self._acq_id, self._par(), fmin, shift, scale)
        val, dval = self._compute_acq_withGradients(x)
        price, dprice = self.cost_withGradients(x)
        feasible = self.space.indicator_constraints(x)
        quotient = val / price
        dquotient = (dval * price - val * dprice) / (price ** 2)
        return -quotient * feasible, -dquotient * feasible

    def _group_stage(self, x, devices):
        """``x`` split over the replicas of the model on ``devices`` (gp_group_*): the group, fmin, normaliser mean / std."""
        gp = self.model.model
        grp = gp._device_group(devices)
        grp.set_candidates(np.atleast_2d(np.asarray(x, dtype=float)))
        nz = gp.normalizer
        shift, scale = (0.0, 1.0) if nz is None else (float(nz.mean[0]), float(nz.std[0]))
        fmin = grp.fmin() if self._rule.needs_fmin else 0.0
        if self._rule.needs_fmin and nz is not None:
            fmin = float(nz.inverse_mean(np.array([[fmin]]))[0, 0])
        return grp, fmin, shift, scale

    def argbest(self, x, sense=-1, devices=None):
        """Row index and value of the best entry of ``acquisition_function(x)``: sense=-1 the smallest (GPyOpt's convention,
        anchor_points_generator.py:61), sense=+1 the largest (run.py:1241 takes ``np.argmax``).  Ties -> lowest index.
        ``devices=[0, 1, ...]``: the table is split over replicas of the model on those GPUs, still from this one process."""
        if self._device_ok():
            if devices is not None:
                grp, fmin, shift, scale = self._group_stage(x, devices)
                return grp.acq_argbest(self._acq_id, self._par(), fmin, sense, shift, scale)
            gp, fmin, shift, scale = self._device_stage(x)
            return gp._h.acq_argbest(self._acq_id, self._par(), fmin, sense, shift, scale)
        return _pick(self.acquisition_function(x)[:, 0], sense)

    def topk(self, x, k, sense=-1, devices=None):
        """The ``k`` best rows in order, (indices, values): what ``AnchorPointsGenerator.get`` keeps
        (anchor_points_generator.py:59-61).  Equal scores lowest index first; fewer than ``k`` rows -> index -1 in the tail.
        ``devices``: as in ``argbest``."""
        if self._device_ok() and k <= 64:
            if devices is not None:
                grp, fmin, shift, scale = self._group_stage(x, devices)
                return grp.acq_topk(self._acq_id, self._par(), fmin, sense, k, shift, scale)
            gp, fmin, shift, scale = self._device_stage(x)
            return gp._h.acq_topk(self._acq_id, self._par(), fmin, sense, k, shift, scale)
        scores = self.acquisition_function(x)[:, 0]
        ranked = np.argsort(scores if sense < 0 else -scores, kind="stable")[:k]
        idx = np.full(k, -1, dtype=np.int64)
        val = np.full(k, np.inf if sense < 0 else -np.inf)
        idx[:ranked.size] = ranked
        val[:ranked.size] = scores[ranked]
        return idx, val

    def optimize(self, duplicate_manager=None):
        """Hand the (negated) acquisition to the acquisition optimiser (base.py:52-60)."""
        kw = dict(f=self.acquisition_function, duplicate_manager=duplicate_manager)
        if self.analytical_gradient_acq:
            kw["f_df"] = self.acquisition_function_withGradients
        return self.optimizer.optimize(**kw)

    # ---- host scoring from model.predict (foreign models, cost, constraints) -----------------------------------------
    def _compute_acq(self, x):
        if self._rule is None:
            raise NotImplementedError('a subclass without a _rule scores itself')
        mu, sd = self.model.predict(x)
        return self._rule.value(self._par(), self._fmin(), mu, sd)

    def _compute_acq_withGradients(self, x):
        if self._rule is None:
            raise NotImplementedError('a subclass without a _rule scores itself')
        fmin = self._fmin()
        mu, sd, dmu_dx, dsd_dx = self.model.predict_withGradients(x)
        return self._rule.gradient(self._par(), fmin, mu, sd, dmu_dx, dsd_dx)


class AcquisitionEI(AcquisitionBase):
    """Expected improvement (EI.py:7-51): ``jitter`` is the improvement margin."""
    analytical_gradient_prediction = True
    _rule = _RULES["EI"]

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        super().__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionEI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter


class AcquisitionLCB(AcquisitionBase):
    """GP lower confidence bound (LCB.py:6-46); a cost model is ignored, as in the reference."""
    analytical_gradient_prediction = True
    _rule = _RULES["LCB"]

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, exploration_weight=2):
        super().__init__(model, space, optimizer)
        self.exploration_weight = exploration_weight
        if cost_withGradients is not None:
            print('The set cost function is ignored! LCB acquisition does not make sense with cost.')

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionLCB(model, space, optimizer, cost_withGradients, exploration_weight=config['weight'])

    def _par(self):
        return self.exploration_weight


class AcquisitionMPI(AcquisitionBase):
    """Maximum probability of improvement (MPI.py:7-51)."""
    analytical_gradient_prediction = True
    _rule = _RULES["MPI"]

    def __init__(self, model, space=None, optimizer=None, cost_withGradients=None, jitter=0.01):
        super().__init__(model, space, optimizer, cost_withGradients=cost_withGradients)
        self.jitter = jitter

    @staticmethod
    def fromConfig(model, space, optimizer, cost_withGradients, config):
        return AcquisitionMPI(model, space, optimizer, cost_withGradients, jitter=config['jitter'])

    def _par(self):
        return self.jitter


# ---- local penalisation ---------------------------------------------------------------------------------------------
def _log_transform(acq, transform):
    """(log T(acq), d log T / d acq) of the positive base acquisition: T = identity ('none': log(acq + 1e-50), slope
    1 / acq) or softplus (log of log(1 + e^acq), taken as log(acq) from 40 on; slope 1 / (softplus (1 + e^-acq)))."""
    acq = np.asarray(acq, dtype=float)
    if transform == 'softplus':
        big = acq >= 40.
        soft = np.log1p(np.exp(acq))
        return np.where(big, np.log(np.where(big, acq, 1.0)), np.log(soft)), 1. / (soft * (1. + np.exp(-acq)))
    if transform == 'none':
        return np.log(acq + 1e-50), 1. / acq
    return acq, np.ones_like(acq)


def _exclusion(x, centres, radius, width):
    """Per (point, centre): z = (|x - centre| - radius) / width and the distance itself, [M, nb] each."""
    gap = np.atleast_2d(x)[:, None, :] - np.atleast_2d(centres)[None, :, :]
    dist = np.sqrt(np.square(gap).sum(-1))
    return (dist - radius) / width, dist


class AcquisitionLP(AcquisitionBase):
    """Local-penalisation wrapper of a base acquisition for batch design (Gonzalez et al. 2016; LP.py:10-140), always in log
    space: ``-log T(acq(x)) - sum_k log Phi((|x - x_k| - r_k) / s_k)`` over the batch points x_k chosen so far."""
    analytical_gradient_prediction = True

    def __init__(self, model, space=None, optimizer=None, acquisition=None, transform='none'):
        super().__init__(model, space, optimizer)
        self.acq = acquisition
        kind = str(transform).lower()
        # LCB can be negative, so its plain logarithm is replaced by log(softplus) (LP.py:32-35)
        self.transform = 'softplus' if (kind == 'none' and isinstance(acquisition, AcquisitionLCB)) else kind
        self.X_batch = self.r_x0 = self.s_x0 = None
        self._lp_packed = None

    def update_batches(self, X_batch, L, Min):
        """Set (or with None: clear) the batch chosen so far; radii and widths of its exclusion balls come from the model's
        prediction at the batch, the Lipschitz constant ``L`` and the best observed value ``Min`` (LP.py:41-62)."""
        self.X_batch, have_batch = X_batch, X_batch is not None
        if have_batch:
            self.r_x0, self.s_x0 = self._ball_parameters(X_batch, L, Min)
        self._lp_packed = None          # (transform, Xb, r, s) as contiguous float64, built once per batch (see _lp_few)

    def _ball_parameters(self, centres, L, Min):
        mu, spread = self.model.predict(np.atleast_2d(centres))
        # the reference floors what ``predict`` returns as its second output (a standard deviation here) at 1e-16 and takes
        # the square root of it once more; kept, since r / s define the batch the reference would pick
        width = np.sqrt(np.maximum(spread, 1e-16)) / L
        return ((mu - Min) / L).ravel(), width.ravel()

    # -- host scoring (foreign models) -------------------------------------------------------------------------------
    def _score_on_host(self, x):
        with np.errstate(over='ignore'):
            logt, _ = _log_transform(-self.acq.acquisition_function(x)[:, 0], self.transform)
        score = -logt
        if self.X_batch is not None:
            z, _ = _exclusion(x, self.X_batch, self.r_x0, self.s_x0)
            score = score - log_ndtr(z).sum(axis=-1)
        return score

    def _penalty_slope(self, x):
        """What the reference subtracts from every gradient component (LP.py:91-103): sum over the batch of
        pdf(z) / (s Phi(z) |x - x_k|) -- the direction factor is absent there, and so here."""
        z, dist = _exclusion(x, self.X_batch, self.r_x0, self.s_x0)
        mass = ndtr(z)
        with np.errstate(divide='ignore', invalid='ignore'):
            term = 1. / (self.s_x0 * _ROOT_2PI * mass) * np.exp(-np.square(z) / 2) / dist
        term[mass < 1e-50] = 0.
        return term.sum(axis=1)[:, None]

    # -- device route ------------------------------------------------------------------------------------------------
    def _lp_device_ok(self):
        base = self.acq
        return base is not None and base.model is self.model and base._device_ok() and self.transform in ('none', 'softplus')

    def _lp_call(self, x):
        base = self.acq
        gp, fmin, shift, scale = base._device_stage(x)
        head = (base._acq_id, base._par(), fmin, 1 if self.transform == 'softplus' else 0)
        batch = dict(Xb=self.X_batch, r_x0=self.r_x0, s_x0=self.s_x0, y_mean=shift, y_std=scale)
        return gp._h, head, batch

    def _lp_few(self, x, grad):
        """ONE gp_acq_rows call for up to ``_FEW_ROWS`` locations (None otherwise): the L-BFGS runs of compute_batch."""
        few = self.acq._device_few(x)
        if few is None:
            return None
        x, gp, fmin, shift, scale = few
        src = (self.X_batch, self.r_x0, self.s_x0, self.transform)
        held = getattr(self, "_lp_packed", None)
        lp = held[1] if held is not None and all(a is b for a, b in zip(held[0], src)) else None
        if lp is None:      # an L-BFGS run makes hundreds of calls with one batch: convert it once (per object identity of the
            tr = 1 if self.transform == 'softplus' else 0          # batch attributes: assigning new arrays renews it)
            if self.X_batch is None:
                lp = (tr, None, None, None)
            else:
                lp = (tr, np.ascontiguousarray(np.atleast_2d(self.X_batch), dtype=float),
                      np.ascontiguousarray(np.atleast_1d(self.r_x0), dtype=float),
                      np.ascontiguousarray(np.atleast_1d(self.s_x0), dtype=float))
            self._lp_packed = (src, lp)
        return gp._h.acq_rows(x, self.acq._acq_id, self.acq._par(), fmin, shift, scale, grad=grad, lp=lp)

    def acquisition_function(self, x):
        """1-D array like the reference's (LP.py:105-110)."""
        if self._lp_device_ok():
            few = self._lp_few(x, False)
            if few is not None:
                return few
            h, head, batch = self._lp_call(x)
            return h.acq_lp(*head, **batch)
        return self._score_on_host(x)

    def d_acquisition_function(self, x):
        return self.acquisition_function_withGradients(x)[1]

    def acquisition_function_withGradients(self, x):
        """(value [M], gradient [M, D])  (LP.py:112-140)."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        if self._lp_device_ok():
            few = self._lp_few(x, True)
            if few is not None:
                return few
            h, head, batch = self._lp_call(x)
            return h.acq_lp_grad(*head, **batch)
        neg, dneg = self.acq.acquisition_function_withGradients(x)
        with np.errstate(over='ignore', divide='ignore'):
            _, slope = _log_transform(-neg[:, 0], self.transform)
        grad = slope[:, None] * dneg
        if self.X_batch is not None:
            grad = grad - self._penalty_slope(x)
        return self.acquisition_function(x), grad

    def argbest(self, x, sense=+1, exclude=(), devices=None):
        """Arg-best of ``acquisition_function(x)`` with the rows in ``exclude`` masked out (run.py:1241,1249-1252).
        ``devices``: the table split over replicas of the model on those GPUs, from this one process (gp_group_*)."""
        if self._lp_device_ok():
            if devices is not None:
                base = self.acq
                grp, fmin, shift, scale = base._group_stage(x, devices)
                return grp.acq_lp_argbest(base._acq_id, base._par(), fmin, 1 if self.transform == 'softplus' else 0, sense,
                                          Xb=self.X_batch, r_x0=self.r_x0, s_x0=self.s_x0, exclude=exclude, y_mean=shift,
                                          y_std=scale)
            h, head, batch = self._lp_call(x)
            return h.acq_lp_argbest(head[0], head[1], head[2], head[3], sense, exclude=exclude, **batch)
        scores = np.array(self.acquisition_function(x), dtype=float)
        scores[list(exclude)] = -np.inf if sense > 0 else np.inf
        return _pick(scores, sense)


def estimate_L(model, bounds, storehistory=True):
    """Lipschitz constant of the posterior mean over the box ``bounds``: the largest |d mean / dx| over 500 uniform draws
    plus the training inputs, polished by L-BFGS-B from the best of them; 10 for a flat model
    (core/evaluators/batch_local_penalization.py:52-70).  ``model`` is the GP (``GPModel.model``): all 500 + N predictive
    gradients come from ONE batched device call."""
    from scipy.optimize import minimize

    mean_jac = getattr(model, "mean_gradients", None)      # the HIP model: d mean / dx without the variance's share of the work

    def neg_slope(pts):
        jac = mean_jac(np.atleast_2d(pts)) if mean_jac is not None else model.predictive_gradients(np.atleast_2d(pts))[0]
        return -np.sqrt((jac * jac).sum(1))

    box = np.asarray(list(bounds), dtype=float)
    # the global generator is consumed as the reference consumes it -- 500 values for the first variable, then 500 for the
    # second, ... (util/general.py:63-73) -- so that a seeded run starts L-BFGS from the reference's point
    unit = np.random.uniform(size=(box.shape[0], 500))
    draws = (box[:, :1] + (box[:, 1:] - box[:, :1]) * unit).T
    pool = np.vstack([draws, model.X])
    start = pool[np.argmin(neg_slope(pool))]
    polished = minimize(lambda p: float(neg_slope(p).ravel()[0]), start, method='L-BFGS-B', bounds=[tuple(b) for b in box],
                        options={'maxiter': 200})
    steepest = -float(polished.fun)
    return steepest if steepest >= 1e-7 else 10


class LocalPenalization(object):
    """Batch evaluator: pick ``batch_size`` points one after the other, each time penalising the acquisition around the
    points already chosen (core/evaluators/batch_local_penalization.py:7-49)."""

    def __init__(self, acquisition, batch_size):
        self.acquisition = acquisition
        self.batch_size = batch_size

    def _constants(self):
        gp = self.acquisition.model.model
        return estimate_L(gp, self.acquisition.space.get_bounds()), gp.Y.min()

    def compute_batch(self, duplicate_manager=None, context_manager=None):
        """Batch by optimising the (penalised) acquisition over the domain."""
        lp = self.acquisition
        assert isinstance(lp, AcquisitionLP)
        lp.update_batches(None, None, None)
        batch = lp.optimize()[0]
        if self.batch_size >= 2:
            lipschitz, best_seen = self._constants()
            for _ in range(self.batch_size - 1):
                lp.update_batches(batch, lipschitz, best_seen)
                batch = np.vstack((batch, lp.optimize()[0]))
        lp.update_batches(None, None, None)
        return batch

    def compute_batch_from_table(self, table, sense=+1, devices=None, lipschitz=None):
        """The candidate-table variant the thesis driver uses (run.py:1234-1258): ``batch_size`` distinct rows of ``table`` by
        repeated arg-best of the penalised acquisition; ``devices``: scored on several GPUs from this one process.
        ``lipschitz``: a constant estimated earlier for the same model (run.py re-estimates the same L for each of its
        three acquisitions, :1244); None estimates it here, after the first row, as the driver does."""
        lp = self.acquisition
        lp.update_batches(None, None, None)
        rows = [lp.argbest(table, sense, devices=devices)[0]]
        if self.batch_size >= 2:
            if lipschitz is None:
                lipschitz, best_seen = self._constants()
            else:
                best_seen = self.acquisition.model.model.Y.min()
            while len(rows) < self.batch_size:
                lp.update_batches(np.atleast_2d(table[rows]), lipschitz, best_seen)
                rows.append(lp.argbest(table, sense, exclude=rows, devices=devices)[0])
        lp.update_batches(None, None, None)
        return rows

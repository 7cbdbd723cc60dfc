"""Minimal parameter plumbing standing in for paramz on this path.

The reference's kernels and likelihood declare ``Param(name, value, Logexp())``
(GPy/GPy/kern/src/stationary.py:80-81, GPy/GPy/likelihoods/gaussian.py:43) and
GPyOpt constrains the noise with ``constrain_fixed`` / ``constrain_bounded``
(GPyOpt/GPyOpt/models/gpmodel.py:72-76).  paramz itself is an un-vendored
dependency of the reference (absent from /root/reference); only what the L-BFGS
loop needs is restated here: a flat parameter vector, the Logexp / Logistic
transforms with their chain-rule factors, and change notification.  Optimiser
trajectories are therefore "parity unpinned" (SURVEY.md 8c); LML, gradients and
posteriors at fixed hyper-parameters do not depend on this module.
"""
import numpy as np

_LIM_VAL = 36.0
_EPS = np.finfo(np.float64).resolution


class Logexp(object):
    """Positive transform: f(x) = log(1 + exp(x))."""

    def f(self, x):
        x = np.asarray(x, dtype=float)
        return np.where(x > _LIM_VAL, x, np.log1p(np.exp(np.clip(x, -np.log(np.finfo(float).max) + 2, _LIM_VAL)))) + _EPS

    def finv(self, f):
        f = np.asarray(f, dtype=float)
        return np.where(f > _LIM_VAL, f, np.log(np.expm1(f)))

    def gradfactor(self, f, df):
        f = np.asarray(f, dtype=float)
        return df * np.where(f > _LIM_VAL, 1.0, -np.expm1(-f))


class Logistic(object):
    """Bounded transform onto (lower, upper)."""

    def __init__(self, lower, upper):
        assert lower < upper
        self.lower, self.upper = float(lower), float(upper)
        self.difference = self.upper - self.lower

    def f(self, x):
        x = np.asarray(x, dtype=float)
        return self.lower + self.difference / (1.0 + np.exp(-x))

    def finv(self, f):
        f = np.clip(np.asarray(f, dtype=float), self.lower + 1e-10 * self.difference, self.upper - 1e-10 * self.difference)
        return np.log((f - self.lower) / (self.upper - f))

    def gradfactor(self, f, df):
        f = np.asarray(f, dtype=float)
        return df * (f - self.lower) * (self.upper - f) / self.difference


class Param(object):
    """A named positive parameter vector with a gradient slot and a constraint."""

    def __init__(self, name, values, transform=None):
        self.name = name
        self._v = np.array(values, dtype=float).reshape(-1)
        self.gradient = np.zeros_like(self._v)
        self.transform = transform if transform is not None else Logexp()
        self.is_fixed = False
        self._parent = None

    # -- value access -----------------------------------------------------------
    @property
    def values(self):
        return self._v

    @property
    def size(self):
        return self._v.size

    def _changed(self):
        if self._parent is not None:
            self._parent._notify()

    def __setitem__(self, idx, val):
        self._v[idx] = val
        self._changed()

    def __getitem__(self, idx):
        return self._v[idx]

    def set(self, val):
        self._v[:] = np.asarray(val, dtype=float).reshape(-1)
        self._changed()

    def __float__(self):
        assert self._v.size == 1
        return float(self._v[0])

    def __array__(self, dtype=None, copy=None):
        return self._v if dtype is None else self._v.astype(dtype)

    def __len__(self):
        return self._v.size

    def __repr__(self):
        return "%s = %s%s" % (self.name, self._v, " (fixed)" if self.is_fixed else "")

    def _op(self, other, fn):
        return fn(self._v, np.asarray(other))

    def __mul__(self, o): return self._op(o, np.multiply)
    __rmul__ = __mul__
    def __add__(self, o): return self._op(o, np.add)
    __radd__ = __add__
    def __sub__(self, o): return self._op(o, np.subtract)
    def __rsub__(self, o): return np.subtract(np.asarray(o), self._v)
    def __truediv__(self, o): return self._op(o, np.divide)
    def __rtruediv__(self, o): return np.divide(np.asarray(o), self._v)
    def __pow__(self, o): return self._op(o, np.power)
    def __neg__(self): return -self._v

    # -- constraints (paramz API names) ---------------------------------------------
    def constrain_fixed(self, value=None, warning=True):
        if value is not None:
            self._v[:] = value
        self.is_fixed = True
        self._changed()
    fix = constrain_fixed

    def unconstrain_fixed(self):
        self.is_fixed = False
    unfix = unconstrain_fixed

    def constrain_bounded(self, lower, upper, warning=True):
        self.transform = Logistic(lower, upper)
        self._v[:] = np.clip(self._v, lower, upper)
        self.is_fixed = False
        self._changed()

    def constrain_positive(self, warning=True):
        self.transform = Logexp()
        self.is_fixed = False


class Parameterized(object):
    """A node holding Params and child nodes; the root is told when anything changes."""

    def __init__(self, name):
        self.name = name
        self._params = []
        self._children = []
        self._parent = None

    def link_parameters(self, *ps):
        for p in ps:
            p._parent = self
            if isinstance(p, Param):
                self._params.append(p)
            else:
                self._children.append(p)
    link_parameter = link_parameters

    def _notify(self):
        if self._parent is not None:
            self._parent._notify()
        else:
            self._on_change()

    def _on_change(self):
        pass

    def flattened_parameters(self):
        out = []
        for c in self._children:
            out.extend(c.flattened_parameters())
        out.extend(self._params)
        return out

    # paramz names ------------------------------------------------------------------
    def parameter_names_flat(self):
        names = []
        for c in self._children:
            names.extend("%s.%s" % (self.name, n.split(".", 1)[-1] if False else n) for n in c.parameter_names_flat())
        for p in self._params:
            if p.size == 1:
                names.append("%s.%s" % (self.name, p.name))
            else:
                names.extend("%s.%s[[%d]]" % (self.name, p.name, i) for i in range(p.size))
        return np.array(names)

    @property
    def param_array(self):
        ps = self.flattened_parameters()
        return np.concatenate([p.values for p in ps]) if ps else np.zeros(0)

    def __getitem__(self, idx):
        return self.param_array[idx]

    def __setitem__(self, idx, val):
        arr = self.param_array.copy()
        arr[idx] = val
        i = 0
        for p in self.flattened_parameters():
            p._v[:] = arr[i:i + p.size]
            i += p.size
        self._notify()

    @property
    def optimizer_array(self):
        ps = [p for p in self.flattened_parameters() if not p.is_fixed]
        return np.concatenate([p.transform.finv(p.values) for p in ps]) if ps else np.zeros(0)

    @optimizer_array.setter
    def optimizer_array(self, x):
        x = np.asarray(x, dtype=float)
        i = 0
        for p in self.flattened_parameters():
            if p.is_fixed:
                continue
            p._v[:] = p.transform.f(x[i:i + p.size])
            i += p.size
        self._notify()

    def _transform_gradients(self, natural_grads):
        """Chain rule through the transforms (paramz Model._transform_gradients)."""
        out = []
        for p, g in natural_grads:
            if p.is_fixed:
                continue
            out.append(p.transform.gradfactor(p.values, np.asarray(g, dtype=float).reshape(-1)))
        return np.concatenate(out) if out else np.zeros(0)

    def constrain_fixed(self, value=None, warning=True):
        for p in self.flattened_parameters():
            p.constrain_fixed(value, warning)
    fix = constrain_fixed

    def constrain_bounded(self, lower, upper, warning=True):
        for p in self.flattened_parameters():
            p.constrain_bounded(lower, upper, warning)

    def constrain_positive(self, warning=True):
        for p in self.flattened_parameters():
            p.constrain_positive(warning)

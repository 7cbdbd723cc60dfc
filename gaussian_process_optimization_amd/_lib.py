"""ctypes binding of libgphip.so (include/gphip.h).  No torch, no CPU fallback.

The product path fails loudly when the HIP library is missing or no GPU is
visible: ``load()`` raises ``RuntimeError``; nothing here (or anywhere in this
package) imports ``oracle/``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgphip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int_p = ctypes.POINTER(ctypes.c_int)

GP_KERNEL_RBF, GP_KERNEL_MATERN52 = 0, 1
GP_ACQ_EI, GP_ACQ_LCB, GP_ACQ_MPI = 0, 1, 2
GP_ERR_ARG, GP_ERR_HIP, GP_ERR_STATE, GP_ERR_RCCL, GP_ERR_NOT_PD_DIAG = -1, -2, -3, -4, -5

# every symbol include/gphip.h declares: (name, restype, argtypes)
_vp = ctypes.c_void_p
SIGNATURES = [
    ("gp_last_error", ctypes.c_char_p, []),
    ("gp_version", ctypes.c_char_p, []),
    ("gp_device_count", ctypes.c_int, [c_int_p]),
    ("gp_device_info", ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, c_int_p, c_int64_p]),
    ("gp_create", ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int]),
    ("gp_destroy", ctypes.c_int, [_vp]),
    ("gp_shutdown", ctypes.c_int, []),
    ("gp_get_fit_state", ctypes.c_int, [_vp, c_double_p, c_double_p, c_double_p]),
    ("gp_get_dl_dk", ctypes.c_int, [_vp, c_double_p]),
    ("gp_posterior_samples", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p,
                                            c_double_p, c_double_p]),
    ("gp_acq_topk", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_int, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_comm_allgather_topk", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_int64_p, c_double_p, c_int64_p]),
    ("gp_set_data", ctypes.c_int, [_vp, c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    ("gp_set_params", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_double]),
    ("gp_set_gower", ctypes.c_int, [_vp, ctypes.c_int, c_int_p, c_double_p]),
    ("gp_fit", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_double_p, c_double_p]),
    ("gp_fit_predict", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                      c_double_p]),
    ("gp_get_alpha", ctypes.c_int, [_vp, c_double_p]),
    ("gp_get_chol", ctypes.c_int, [_vp, c_double_p]),
    ("gp_get_woodbury_inv", ctypes.c_int, [_vp, c_double_p]),
    ("gp_kernel_matrix", ctypes.c_int, [_vp, c_double_p]),
    ("gp_cross_kernel_matrix", ctypes.c_int, [_vp, c_double_p, ctypes.c_int64, c_double_p]),
    ("gp_lml_grad", ctypes.c_int, [_vp, c_double_p, c_double_p, c_double_p]),
    ("gp_fit_grad", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                   c_double_p]),
    ("gp_set_candidates", ctypes.c_int, [_vp, c_double_p, ctypes.c_int64]),
    ("gp_predict", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_double_p]),
    ("gp_predict_full_cov", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_double_p]),
    ("gp_predict_grad", ctypes.c_int, [_vp, c_double_p, c_double_p]),
    ("gp_fmin", ctypes.c_int, [_vp, c_double_p]),
    ("gp_acq", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, c_double_p]),
    ("gp_acq_argbest", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_acq_grad", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, c_double_p, c_double_p]),
    ("gp_acq_lp", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                 ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p]),
    ("gp_acq_lp_argbest", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p,
                                         ctypes.c_int, c_int64_p, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_acq_lp_grad", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    # (the double* arguments of these two are declared void*: the caller passes ndarray.ctypes.data, an integer -- building a
    # typed ctypes pointer costs ~4 us apiece, a tenth of the whole call)
    ("gp_predict_rows", ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, _vp]),
    ("gp_acq_rows", ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int,
                                   _vp, _vp, _vp, _vp]),
    ("gp_rows_stats", ctypes.c_int, [_vp, c_int64_p, c_int64_p]),
    ("gp_comm_unique_id", ctypes.c_int, [ctypes.c_char_p]),
    ("gp_comm_init", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    ("gp_comm_destroy", ctypes.c_int, [_vp]),
    ("gp_comm_info", ctypes.c_int, [_vp, c_int_p, c_int_p]),
    ("gp_comm_allgather_best", ctypes.c_int, [_vp, ctypes.c_double, ctypes.c_int64, c_double_p, c_int64_p]),
    ("gp_comm_bcast_fit", ctypes.c_int, [_vp, ctypes.c_int]),
    ("gp_comm_version", ctypes.c_int, [c_int_p]),
    ("gp_comm_selftest_fit_record", ctypes.c_int, [c_double_p, c_double_p, c_int_p]),
    ("gp_group_create", ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, c_int_p]),
    ("gp_group_destroy", ctypes.c_int, [_vp]),
    ("gp_group_info", ctypes.c_int, [_vp, c_int_p, c_int_p, ctypes.c_char_p, ctypes.c_int]),
    ("gp_group_member", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(_vp)]),
    ("gp_group_set_data", ctypes.c_int, [_vp, c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    ("gp_group_set_params", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_double]),
    ("gp_group_set_gower", ctypes.c_int, [_vp, ctypes.c_int, c_int_p, c_double_p]),
    ("gp_group_set_option", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
    ("gp_group_fit", ctypes.c_int, [_vp, ctypes.c_int, c_double_p, c_double_p, c_double_p]),
    ("gp_group_fmin", ctypes.c_int, [_vp, c_double_p]),
    ("gp_group_set_candidates", ctypes.c_int, [_vp, c_double_p, ctypes.c_int64]),
    ("gp_group_acq_argbest", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_group_acq_lp_argbest", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                               ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_int, c_double_p, c_double_p,
                                               ctypes.c_int, c_int64_p, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_group_acq_topk", ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_merge_best", ctypes.c_int, [ctypes.c_int, c_double_p, c_int64_p, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_merge_topk", ctypes.c_int, [ctypes.c_int, c_double_p, c_int64_p, ctypes.c_int, ctypes.c_int, c_int64_p, c_double_p]),
    ("gp_last_phases", ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), c_double_p, c_double_p,
                                      c_double_p]),
    ("gp_profile", ctypes.c_int, [_vp, ctypes.c_int]),
    ("gp_gemm_stats", ctypes.c_int, [_vp, c_int64_p, c_double_p, c_double_p]),
    ("gp_rns_stats", ctypes.c_int, [_vp, c_int64_p, c_double_p, c_double_p]),
    ("gp_gemm_busy", ctypes.c_int, [_vp, c_double_p]),
    ("gp_gemm_trace", ctypes.c_int, [_vp, ctypes.c_int, c_int64_p, c_int_p, c_double_p]),
    ("gp_synchronize", ctypes.c_int, [_vp]),
    ("gp_set_option", ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int64]),
]

_lib = None


def load_library():
    """dlopen libgphip.so and bind every declared symbol (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libgphip.so is missing at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def load():
    """Library handle, insisting on a visible GPU."""
    lib = load_library()
    n = ctypes.c_int(0)
    rc = lib.gp_device_count(ctypes.byref(n))
    if rc != 0 or n.value < 1:
        raise RuntimeError("gphip: no HIP device visible (%s). The GP path runs on MI355X only; "
                           "there is no CPU fallback." % lib.gp_last_error().decode())
    return lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def as_f64(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError("expected a %d-D array, got shape %s" % (ndim, a.shape))
    return a


def check(lib, rc, what):
    """Map C return codes to the exceptions the reference raises (linalg.py:62-75, bo.py:134-137)."""
    if rc == 0:
        return
    msg = lib.gp_last_error().decode()
    if rc > 0:
        raise np.linalg.LinAlgError("not positive definite, even with jitter.")
    if rc == GP_ERR_NOT_PD_DIAG:
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    if rc == GP_ERR_ARG:
        raise ValueError("%s: %s" % (what, msg))
    raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


class Handle(object):
    """Owns one gp_t (one device)."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        check(self.lib, self.lib.gp_create(ctypes.byref(h), int(device)), "gp_create")
        self.h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- thin typed wrappers -------------------------------------------------
    def set_data(self, X, Y):
        X = as_f64(X, 2)
        Y = as_f64(Y, 2)
        if X.shape[0] != Y.shape[0]:
            raise ValueError("X and Y row counts differ")
        check(self.lib, self.lib.gp_set_data(self.h, dptr(X), dptr(Y), X.shape[0], X.shape[1], Y.shape[1]),
              "gp_set_data")
        self.N, self.D, self.P = X.shape[0], X.shape[1], Y.shape[1]

    def set_params(self, kernel, ard, variance, lengthscale, noise):
        ls = as_f64(np.atleast_1d(lengthscale), 1)
        if ls.size != (self.D if ard else 1):
            raise ValueError("lengthscale has %d entries, expected %d" % (ls.size, self.D if ard else 1))
        check(self.lib, self.lib.gp_set_params(self.h, int(kernel), int(bool(ard)), float(variance), dptr(ls),
                                               float(noise)), "gp_set_params")
        self.n_ls = ls.size

    def _grad_ls(self, nls):
        """The lengthscale-gradient buffer for gp_lml_grad / gp_fit_grad: the library writes one entry per lengthscale of the
        LAST gp_set_params (1, or D with ard) whatever the caller expects, so a wrong count is refused here."""
        want = getattr(self, "n_ls", None)
        if want is None:
            raise ValueError("set_params has not been called on this handle")
        if int(nls) != want:
            raise ValueError("the model has %d lengthscale(s), the caller asked for %d gradient entries" % (want, int(nls)))
        return np.empty(want)

    def set_gower(self, is_discrete=None, ranges=None):
        """Enable (or, with no arguments, disable) the fork's Gower product kernel."""
        if is_discrete is None:
            check(self.lib, self.lib.gp_set_gower(self.h, 0, None, None), "gp_set_gower")
            return
        disc = np.ascontiguousarray(is_discrete, dtype=np.int32)
        rng = as_f64(ranges, 1)
        if disc.size != self.D or rng.size != self.D:
            raise ValueError("is_discrete / ranges need one entry per input dimension")
        check(self.lib, self.lib.gp_set_gower(self.h, 1, disc.ctypes.data_as(c_int_p), dptr(rng)), "gp_set_gower")

    def fit(self, maxtries=5):
        lml, logdet, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        rc = self.lib.gp_fit(self.h, int(maxtries), ctypes.byref(lml), ctypes.byref(logdet), ctypes.byref(jit))
        check(self.lib, rc, "gp_fit")
        return lml.value, logdet.value, jit.value

    def fit_predict(self, include_noise=True, maxtries=5):
        """gp_fit + gp_predict on the resident candidates as one pipelined pass."""
        lml, logdet, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        mean = np.empty((self.M, self.P))
        var = np.empty((self.M, 1))
        rc = self.lib.gp_fit_predict(self.h, int(maxtries), int(bool(include_noise)), ctypes.byref(lml),
                                     ctypes.byref(logdet), ctypes.byref(jit), dptr(mean), dptr(var))
        check(self.lib, rc, "gp_fit_predict")
        return (lml.value, logdet.value, jit.value), mean, var

    def fit_grad(self, nls, maxtries=5):
        """gp_fit + gp_lml_grad as one call: ((lml, logdet, jitter), (dvariance, dlengthscale[nls], dnoise))."""
        lml, logdet, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        dv, dn = ctypes.c_double(), ctypes.c_double()
        dl = self._grad_ls(nls)
        rc = self.lib.gp_fit_grad(self.h, int(maxtries), ctypes.byref(lml), ctypes.byref(logdet), ctypes.byref(jit),
                                  ctypes.byref(dv), dptr(dl), ctypes.byref(dn))
        check(self.lib, rc, "gp_fit_grad")
        return (lml.value, logdet.value, jit.value), (dv.value, dl, dn.value)

    def fit_state(self):
        lml, logdet, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(self.lib, self.lib.gp_get_fit_state(self.h, ctypes.byref(lml), ctypes.byref(logdet), ctypes.byref(jit)),
              "gp_get_fit_state")
        return lml.value, logdet.value, jit.value

    def dL_dK(self):
        out = np.empty((self.N, self.N))
        check(self.lib, self.lib.gp_get_dl_dk(self.h, dptr(out)), "gp_get_dl_dk")
        return out

    def posterior_samples(self, Z, include_noise=False, maxtries=5):
        """Z[S, M] standard normals -> (mean[M, P], dev[S, M], jitter): draw s of output d = mean[:, d] + dev[s]."""
        Z = as_f64(Z, 2)
        if Z.shape[1] != self.M:
            raise ValueError("Z needs one column per resident candidate")
        mean = np.empty((self.M, self.P))
        dev = np.empty((Z.shape[0], self.M))
        jit = ctypes.c_double()
        rc = self.lib.gp_posterior_samples(self.h, int(bool(include_noise)), dptr(Z), Z.shape[0], int(maxtries),
                                           dptr(mean), dptr(dev), ctypes.byref(jit))
        check(self.lib, rc, "gp_posterior_samples")
        return mean, dev, jit.value

    def acq_topk(self, type_, par, fmin, sense, k, y_mean=0.0, y_std=1.0):
        idx = np.empty(k, dtype=np.int64)
        val = np.empty(k)
        check(self.lib, self.lib.gp_acq_topk(self.h, int(type_), float(par), float(fmin), float(y_mean), float(y_std),
                                             int(sense), int(k), idx.ctypes.data_as(c_int64_p), dptr(val)),
              "gp_acq_topk")
        return idx, val

    def alpha(self):
        out = np.empty((self.N, self.P))
        check(self.lib, self.lib.gp_get_alpha(self.h, dptr(out)), "gp_get_alpha")
        return out

    def chol(self):
        out = np.empty((self.N, self.N))
        check(self.lib, self.lib.gp_get_chol(self.h, dptr(out)), "gp_get_chol")
        return out

    def woodbury_inv(self):
        out = np.empty((self.N, self.N))
        check(self.lib, self.lib.gp_get_woodbury_inv(self.h, dptr(out)), "gp_get_woodbury_inv")
        return out

    def kernel_matrix(self):
        out = np.empty((self.N, self.N))
        check(self.lib, self.lib.gp_kernel_matrix(self.h, dptr(out)), "gp_kernel_matrix")
        return out

    def cross_kernel_matrix(self, X2):
        """kern.K(X, X2): [N, M2] (stationary.py:107-140 with X2 given)."""
        X2 = as_f64(X2, 2)
        if X2.shape[1] != self.D:
            raise ValueError("X2 has %d columns, model has %d" % (X2.shape[1], self.D))
        out = np.empty((self.N, X2.shape[0]))
        check(self.lib, self.lib.gp_cross_kernel_matrix(self.h, dptr(X2), X2.shape[0], dptr(out)),
              "gp_cross_kernel_matrix")
        return out

    def lml_grad(self, nls):
        dv, dn = ctypes.c_double(), ctypes.c_double()
        dl = self._grad_ls(nls)
        check(self.lib, self.lib.gp_lml_grad(self.h, ctypes.byref(dv), dptr(dl), ctypes.byref(dn)), "gp_lml_grad")
        return dv.value, dl, dn.value

    def set_candidates(self, Xs):
        Xs = as_f64(Xs, 2)
        if Xs.shape[1] != self.D:
            raise ValueError("candidates have %d columns, model has %d" % (Xs.shape[1], self.D))
        check(self.lib, self.lib.gp_set_candidates(self.h, dptr(Xs), Xs.shape[0]), "gp_set_candidates")
        self.M = Xs.shape[0]

    def predict(self, include_noise=True):
        mean = np.empty((self.M, self.P))
        var = np.empty((self.M, 1))
        check(self.lib, self.lib.gp_predict(self.h, int(bool(include_noise)), dptr(mean), dptr(var)), "gp_predict")
        return mean, var

    def predict_full_cov(self, include_noise=True):
        mean = np.empty((self.M, self.P))
        cov = np.empty((self.M, self.M))
        check(self.lib, self.lib.gp_predict_full_cov(self.h, int(bool(include_noise)), dptr(mean), dptr(cov)),
              "gp_predict_full_cov")
        return mean, cov

    def predict_grad(self, mean_only=False):
        dm = np.empty((self.M, self.D, self.P))
        if mean_only:      # the mean's gradients alone: no Ky^-1, no K* Ky^-1 (include/gphip.h)
            check(self.lib, self.lib.gp_predict_grad(self.h, dptr(dm), None), "gp_predict_grad")
            return dm
        dv = np.empty((self.M, self.D))
        check(self.lib, self.lib.gp_predict_grad(self.h, dptr(dm), dptr(dv)), "gp_predict_grad")
        return dm, dv

    def fmin(self):
        v = ctypes.c_double()
        check(self.lib, self.lib.gp_fmin(self.h, ctypes.byref(v)), "gp_fmin")
        return v.value

    def acq(self, type_, par, fmin, y_mean=0.0, y_std=1.0):
        out = np.empty((self.M, 1))
        check(self.lib, self.lib.gp_acq(self.h, int(type_), float(par), float(fmin), float(y_mean), float(y_std),
                                        dptr(out)), "gp_acq")
        return out

    def acq_grad(self, type_, par, fmin, y_mean=0.0, y_std=1.0):
        out = np.empty((self.M, 1))
        dout = np.empty((self.M, self.D))
        check(self.lib, self.lib.gp_acq_grad(self.h, int(type_), float(par), float(fmin), float(y_mean),
                                             float(y_std), dptr(out), dptr(dout)), "gp_acq_grad")
        return out, dout

    def acq_argbest(self, type_, par, fmin, sense, y_mean=0.0, y_std=1.0):
        idx, val = ctypes.c_int64(), ctypes.c_double()
        check(self.lib, self.lib.gp_acq_argbest(self.h, int(type_), float(par), float(fmin), float(y_mean),
                                                float(y_std), int(sense), ctypes.byref(idx), ctypes.byref(val)),
              "gp_acq_argbest")
        return idx.value, val.value

    # -- a handful of locations per call (include/gphip.h, gp_*_rows): set_candidates + the batched call in ONE entry point
    def predict_rows(self, Xs, include_noise=True, grad=False):
        """(mean [M, P], var [M, 1]) and, with ``grad``, (dmdx [M, D, P], dvdx [M, D]) as well."""
        Xs = as_f64(Xs, 2)
        M = Xs.shape[0]
        mean, var = np.empty((M, self.P)), np.empty((M, 1))
        if grad:
            dm, dv = np.empty((M, self.D, self.P)), np.empty((M, self.D))
            rc = self.lib.gp_predict_rows(self.h, Xs.ctypes.data, M, 1 if include_noise else 0, mean.ctypes.data,
                                          var.ctypes.data, dm.ctypes.data, dv.ctypes.data)
        else:
            rc = self.lib.gp_predict_rows(self.h, Xs.ctypes.data, M, 1 if include_noise else 0, mean.ctypes.data,
                                          var.ctypes.data, None, None)
        if rc:
            check(self.lib, rc, "gp_predict_rows")
        return (mean, var, dm, dv) if grad else (mean, var)

    def mean_grad_rows(self, Xs):
        """d mean / dx [M, D, P] of a handful of locations, alone: one pass over the training points (include/gphip.h)."""
        Xs = as_f64(Xs, 2)
        M = Xs.shape[0]
        dm = np.empty((M, self.D, self.P))
        rc = self.lib.gp_predict_rows(self.h, Xs.ctypes.data, M, 0, None, None, dm.ctypes.data, None)
        if rc:
            check(self.lib, rc, "gp_predict_rows")
        return dm

    def acq_rows(self, Xs, type_, par, fmin, y_mean=0.0, y_std=1.0, grad=False, lp=None):
        """Negated acquisition [M, 1] (and its gradient [M, D]); ``lp`` = (transform, Xb, r_x0, s_x0) adds the local
        penalisation, in which case the value comes back 1-D as AcquisitionLP returns it."""
        Xs = as_f64(Xs, 2)
        M = Xs.shape[0]
        out = np.empty(M) if lp is not None else np.empty((M, 1))
        dout = np.empty((M, self.D)) if grad else None
        on = tr = nb = 0
        pX = pr = ps = keep = None
        if lp is not None:
            on, tr = 1, int(lp[0])
            if lp[1] is not None:
                keep = (as_f64(np.atleast_2d(lp[1]), 2), as_f64(np.atleast_1d(lp[2]), 1), as_f64(np.atleast_1d(lp[3]), 1))
                nb, pX, pr, ps = keep[0].shape[0], keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data
        rc = self.lib.gp_acq_rows(self.h, Xs.ctypes.data, M, int(type_), par, fmin, y_mean, y_std, on, tr, pX, nb, pr, ps,
                                  out.ctypes.data, dout.ctypes.data if grad else None)
        if rc:
            check(self.lib, rc, "gp_acq_rows")
        return (out, dout) if grad else out

    def rows_stats(self):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        check(self.lib, self.lib.gp_rows_stats(self.h, ctypes.byref(a), ctypes.byref(b)), "gp_rows_stats")
        return dict(fused=a.value, fallback=b.value)

    def _lp_args(self, Xb, r_x0, s_x0):
        if Xb is None:
            return None, 0, None, None, None
        Xb = as_f64(np.atleast_2d(Xb), 2)
        r = as_f64(np.atleast_1d(r_x0), 1)
        s = as_f64(np.atleast_1d(s_x0), 1)
        return (Xb, r, s), Xb.shape[0], dptr(Xb), dptr(r), dptr(s)

    def acq_lp(self, type_, par, fmin, transform, Xb=None, r_x0=None, s_x0=None, y_mean=0.0, y_std=1.0):
        keep, nb, pX, pr, ps = self._lp_args(Xb, r_x0, s_x0)
        out = np.empty(self.M)
        check(self.lib, self.lib.gp_acq_lp(self.h, int(type_), float(par), float(fmin), float(y_mean), float(y_std),
                                           int(transform), pX, nb, pr, ps, dptr(out)), "gp_acq_lp")
        return out

    def acq_lp_grad(self, type_, par, fmin, transform, Xb=None, r_x0=None, s_x0=None, y_mean=0.0, y_std=1.0):
        """Penalised acquisition and its gradient at the resident candidates: (value[M], gradient[M, D])."""
        keep, nb, pX, pr, ps = self._lp_args(Xb, r_x0, s_x0)
        out = np.empty(self.M)
        dout = np.empty((self.M, self.D))
        check(self.lib, self.lib.gp_acq_lp_grad(self.h, int(type_), float(par), float(fmin), float(y_mean), float(y_std),
                                                int(transform), pX, nb, pr, ps, dptr(out), dptr(dout)), "gp_acq_lp_grad")
        return out, dout

    def acq_lp_argbest(self, type_, par, fmin, transform, sense, Xb=None, r_x0=None, s_x0=None, exclude=(),
                       y_mean=0.0, y_std=1.0):
        keep, nb, pX, pr, ps = self._lp_args(Xb, r_x0, s_x0)
        ex = np.asarray(list(exclude), dtype=np.int64)
        idx, val = ctypes.c_int64(), ctypes.c_double()
        check(self.lib, self.lib.gp_acq_lp_argbest(self.h, int(type_), float(par), float(fmin), float(y_mean),
                                                   float(y_std), int(transform), pX, nb, pr, ps, int(sense),
                                                   ex.ctypes.data_as(c_int64_p), int(ex.size), ctypes.byref(idx),
                                                   ctypes.byref(val)), "gp_acq_lp_argbest")
        return idx.value, val.value

    def phases(self):
        cap = 16
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_double * cap)()
        fl = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        n = self.lib.gp_last_phases(self.h, cap, names, ms, fl, by)
        return [dict(name=names[i].decode(), ms=ms[i], flops=fl[i], bytes=by[i]) for i in range(max(n, 0))]

    def profile(self, on=True):
        """on: False/0 off, True/1 events around every launch of the dominant GEMM symbol (8 waves, 128-tiles), 2 around
        every launch of the 64 x 64 work-unit symbol instead (stalls the latency chain: for an un-timed pass only)."""
        check(self.lib, self.lib.gp_profile(self.h, int(on)), "gp_profile")

    def gemm_stats(self):
        n, ms, fl = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        check(self.lib, self.lib.gp_gemm_stats(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)),
              "gp_gemm_stats")
        return dict(launches=n.value, ms=ms.value, flops=fl.value)

    def rns_stats(self):
        n, ms, ops = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        check(self.lib, self.lib.gp_rns_stats(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(ops)), "gp_rns_stats")
        return dict(launches=n.value, ms=ms.value, ops=ops.value)

    def gemm_busy(self):
        b = ctypes.c_double()
        check(self.lib, self.lib.gp_gemm_busy(self.h, ctypes.byref(b)), "gp_gemm_busy")
        return b.value

    def gemm_trace(self, cap=100000):
        tiles = np.empty(cap, dtype=np.int64)
        K = np.empty(cap, dtype=np.int32)
        ms = np.empty(cap)
        n = self.lib.gp_gemm_trace(self.h, cap, tiles.ctypes.data_as(c_int64_p), K.ctypes.data_as(c_int_p), dptr(ms))
        return tiles[:n], K[:n], ms[:n]

    def synchronize(self):
        check(self.lib, self.lib.gp_synchronize(self.h), "gp_synchronize")

    def set_option(self, name, value):
        check(self.lib, self.lib.gp_set_option(self.h, name.encode(), int(value)), "gp_set_option")

    # -- RCCL ------------------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        check(self.lib, self.lib.gp_comm_unique_id(buf), "gp_comm_unique_id")
        return buf.raw

    def comm_version(self):
        v = ctypes.c_int()
        check(self.lib, self.lib.gp_comm_version(ctypes.byref(v)), "gp_comm_version")
        return v.value

    def comm_init(self, uid, rank, nranks):
        check(self.lib, self.lib.gp_comm_init(self.h, uid, int(rank), int(nranks)), "gp_comm_init")

    def comm_info(self):
        r, n = ctypes.c_int(), ctypes.c_int()
        check(self.lib, self.lib.gp_comm_info(self.h, ctypes.byref(r), ctypes.byref(n)), "gp_comm_info")
        return r.value, n.value

    def comm_allgather_best(self, val, idx, nranks):
        vals = np.empty(nranks)
        idxs = np.empty(nranks, dtype=np.int64)
        check(self.lib, self.lib.gp_comm_allgather_best(self.h, float(val), int(idx), dptr(vals),
                                                        idxs.ctypes.data_as(c_int64_p)), "gp_comm_allgather_best")
        return vals, idxs

    def comm_allgather_topk(self, vals, idxs, nranks):
        vals = as_f64(vals, 1)
        idxs = np.ascontiguousarray(idxs, dtype=np.int64)
        k = vals.size
        av = np.empty(nranks * k)
        ai = np.empty(nranks * k, dtype=np.int64)
        check(self.lib, self.lib.gp_comm_allgather_topk(self.h, k, dptr(vals), idxs.ctypes.data_as(c_int64_p), dptr(av),
                                                        ai.ctypes.data_as(c_int64_p)), "gp_comm_allgather_topk")
        return av, ai

    def comm_bcast_fit(self, root=0):
        check(self.lib, self.lib.gp_comm_bcast_fit(self.h, int(root)), "gp_comm_bcast_fit")


class Group(object):
    """Owns one gp_group_t: the model replicated on ``devices`` (a device may appear more than once), one candidate table
    split over them, winners merged with NumPy's lowest-index rule (include/gphip.h, gp_group_*)."""

    def __init__(self, devices):
        self.lib = load()
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        if devs.ndim != 1 or devs.size < 1:
            raise ValueError("devices must be a non-empty list of HIP device ordinals")
        h = _vp()
        check(self.lib, self.lib.gp_group_create(ctypes.byref(h), int(devs.size), devs.ctypes.data_as(c_int_p)),
              "gp_group_create")
        self.h = h
        self.devices = tuple(int(d) for d in devs)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gp_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, r = ctypes.c_int(), ctypes.c_int()
        buf = ctypes.create_string_buffer(256)
        check(self.lib, self.lib.gp_group_info(self.h, ctypes.byref(n), ctypes.byref(r), buf, 256), "gp_group_info")
        return dict(ndev=n.value, rccl=bool(r.value), note=buf.value.decode())

    def set_data(self, X, Y):
        X, Y = as_f64(X, 2), as_f64(Y, 2)
        if X.shape[0] != Y.shape[0]:
            raise ValueError("X and Y row counts differ")
        check(self.lib, self.lib.gp_group_set_data(self.h, dptr(X), dptr(Y), X.shape[0], X.shape[1], Y.shape[1]),
              "gp_group_set_data")
        self.N, self.D, self.P = X.shape[0], X.shape[1], Y.shape[1]

    def set_params(self, kernel, ard, variance, lengthscale, noise):
        ls = as_f64(np.atleast_1d(lengthscale), 1)
        if ls.size != (self.D if ard else 1):
            raise ValueError("lengthscale has %d entries, expected %d" % (ls.size, self.D if ard else 1))
        check(self.lib, self.lib.gp_group_set_params(self.h, int(kernel), int(bool(ard)), float(variance), dptr(ls),
                                                     float(noise)), "gp_group_set_params")

    def set_gower(self, is_discrete=None, ranges=None):
        if is_discrete is None:
            check(self.lib, self.lib.gp_group_set_gower(self.h, 0, None, None), "gp_group_set_gower")
            return
        disc = np.ascontiguousarray(is_discrete, dtype=np.int32)
        rng = as_f64(ranges, 1)
        check(self.lib, self.lib.gp_group_set_gower(self.h, 1, disc.ctypes.data_as(c_int_p), dptr(rng)), "gp_group_set_gower")

    def set_option(self, name, value):
        check(self.lib, self.lib.gp_group_set_option(self.h, name.encode(), int(value)), "gp_group_set_option")

    def fit(self, maxtries=5):
        lml, logdet, jit = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(self.lib, self.lib.gp_group_fit(self.h, int(maxtries), ctypes.byref(lml), ctypes.byref(logdet),
                                              ctypes.byref(jit)), "gp_group_fit")
        return lml.value, logdet.value, jit.value

    def fmin(self):
        v = ctypes.c_double()
        check(self.lib, self.lib.gp_group_fmin(self.h, ctypes.byref(v)), "gp_group_fmin")
        return v.value

    def set_candidates(self, Xs):
        Xs = as_f64(Xs, 2)
        if Xs.shape[1] != self.D:
            raise ValueError("candidates have %d columns, model has %d" % (Xs.shape[1], self.D))
        check(self.lib, self.lib.gp_group_set_candidates(self.h, dptr(Xs), Xs.shape[0]), "gp_group_set_candidates")
        self.M = Xs.shape[0]

    def acq_argbest(self, type_, par, fmin, sense, y_mean=0.0, y_std=1.0):
        idx, val = ctypes.c_int64(), ctypes.c_double()
        check(self.lib, self.lib.gp_group_acq_argbest(self.h, int(type_), float(par), float(fmin), float(y_mean),
                                                      float(y_std), int(sense), ctypes.byref(idx), ctypes.byref(val)),
              "gp_group_acq_argbest")
        return idx.value, val.value

    def acq_lp_argbest(self, type_, par, fmin, transform, sense, Xb=None, r_x0=None, s_x0=None, exclude=(), y_mean=0.0,
                       y_std=1.0):
        if Xb is None:
            keep, nb, pX, pr, ps = None, 0, None, None, None
        else:
            Xb, r, sc = as_f64(np.atleast_2d(Xb), 2), as_f64(np.atleast_1d(r_x0), 1), as_f64(np.atleast_1d(s_x0), 1)
            keep, nb, pX, pr, ps = (Xb, r, sc), Xb.shape[0], dptr(Xb), dptr(r), dptr(sc)
        ex = np.asarray(list(exclude), dtype=np.int64)
        idx, val = ctypes.c_int64(), ctypes.c_double()
        check(self.lib, self.lib.gp_group_acq_lp_argbest(self.h, int(type_), float(par), float(fmin), float(y_mean),
                                                         float(y_std), int(transform), pX, nb, pr, ps, int(sense),
                                                         ex.ctypes.data_as(c_int64_p), int(ex.size), ctypes.byref(idx),
                                                         ctypes.byref(val)), "gp_group_acq_lp_argbest")
        return idx.value, val.value

    def acq_topk(self, type_, par, fmin, sense, k, y_mean=0.0, y_std=1.0):
        idx = np.empty(k, dtype=np.int64)
        val = np.empty(k)
        check(self.lib, self.lib.gp_group_acq_topk(self.h, int(type_), float(par), float(fmin), float(y_mean), float(y_std),
                                                   int(sense), int(k), idx.ctypes.data_as(c_int64_p), dptr(val)),
              "gp_group_acq_topk")
        return idx, val

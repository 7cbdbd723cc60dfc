"""GPU: the cooperative tail of the factorisation (option "tail_tiles": the trailing tile columns in ONE persistent
launch, potrf.hip chol_tail_kernel) against the stream version it replaces and against the oracle."""
import numpy as np
import pytest

from gaussian_process_optimization_amd import _lib
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def h():
    hd = _lib.Handle(0)
    yield hd
    hd.close()


@pytest.mark.parametrize("N,pt,tail,wgs,la", [(100, 6, 64, 0, 1), (128, 6, 64, 1, 1), (700, 2, 3, 0, 1), (700, 6, 64, 7, 0),
                                              (1500, 2, 5, 0, 1), (1500, 3, 64, 0, 1), (2048, 4, 8, 33, 1), (2048, 6, 64, 0, 0),
                                              (4096, 6, 20, 0, 1), (4096, 6, 64, 0, 1), (5000, 6, 16, 256, 1)])
def test_tail_kernel_equals_stream_version(h, N, pt, tail, wgs, la):
    X, Y, Xs = O.synthetic_problem(N, 4, 50, seed=N)
    P = 2 if N == 700 else 1
    if P == 2:
        Y = np.c_[Y, np.cos(3 * X[:, :1])]
    h.set_option("panel_tiles", pt)
    h.set_option("lookahead", la)
    try:
        h.set_data(X, Y)
        h.set_params(1, 0, 1.3, [0.6], 1e-2)
        h.set_candidates(Xs)
        f0 = h.fit()
        L0, a0 = h.chol(), h.alpha()
        m0, v0 = h.predict(True)
        h.set_option("tail_tiles", tail)
        h.set_option("tail_wgs", wgs)
        f1 = h.fit()
        L1, a1 = h.chol(), h.alpha()
        m1, v1 = h.predict(True)
        # same tile routines, same k order per product; only the summation grouping of the rank-128 updates differs
        assert abs(f1[0] - f0[0]) <= 1e-12 * abs(f0[0])
        assert np.max(np.abs(L1 - L0)) <= 1e-12 * np.max(np.abs(L0))
        assert np.max(np.abs(a1 - a0)) <= 1e-9 * np.max(np.abs(a0))
        assert np.max(np.abs(m1 - m0)) <= 1e-9 * max(1.0, np.max(np.abs(m0))) and np.max(np.abs(v1 - v0)) <= 1e-9
        f2 = h.fit()
        assert f2 == f1 and np.array_equal(h.chol(), L1)      # repeatable
        if N <= 2048:
            gp = O.OracleGP(X, Y, O.Matern52(4, 1.3, 0.6), 1e-2)
            p = gp.posterior
            assert abs(f1[0] - p["lml"]) <= 1e-8 * abs(p["lml"])
            assert np.max(np.abs(a1 - p["alpha"])) <= 1e-6 * np.max(np.abs(p["alpha"]))
        # gradients and the one-call entry points on top of a tail-factored matrix
        g1 = h.lml_grad(1)
        h.set_option("tail_tiles", 0)
        h.fit()
        g0 = h.lml_grad(1)
        assert abs(g1[0] - g0[0]) <= 1e-8 * max(1.0, abs(g0[0])) and abs(g1[2] - g0[2]) <= 1e-8 * max(1.0, abs(g0[2]))
    finally:
        h.set_option("tail_tiles", 0)
        h.set_option("tail_wgs", 0)
        h.set_option("panel_tiles", 6)
        h.set_option("lookahead", 1)


def test_tail_kernel_reports_a_non_positive_pivot(golden, h):
    h.set_option("tail_tiles", 64)
    try:
        h.set_data(golden["jit3/X"], golden["jit3/Y"])
        h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jit3/noise"]))
        lml, logdet, jit = h.fit(5)
        assert jit == pytest.approx(float(golden["jit3/jitter"]), rel=1e-12)
        h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jitfail/noise"]))
        with pytest.raises(np.linalg.LinAlgError):
            h.fit(5)
    finally:
        h.set_option("tail_tiles", 0)

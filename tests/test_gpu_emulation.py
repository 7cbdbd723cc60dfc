"""GPU: the fp64-emulation prototype (option "emulate_fp64", default OFF): the candidate solve's updates run as int8
MFMA contractions in residue form (csrc/rns.hip).  It only counts if it meets the north-star tolerances wherever the
true-fp64 path does: golden vectors, the extended-precision truth of the stress cases, ragged shapes, and the headline
configuration against the independent full-size oracle."""
import os

import numpy as np
import pytest

from conftest import Case, relmax
from gaussian_process_optimization_amd import _lib
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def h():
    hd = _lib.Handle(0)
    yield hd
    hd.close()


def _tags():
    g = np.load(os.path.join(HERE, "golden", "gp_golden.npz"))
    return sorted({k.split("/")[0] for k in g.files if k.startswith("N512") or k.startswith("N300")})


@pytest.mark.parametrize("tag", _tags())
def test_emulated_candidate_solve_on_golden_cases(golden, h, tag):
    c = Case(golden, tag)
    noise = float(c.noise)
    h.set_option("panel_tiles", 1)     # N <= 512: four one-tile panels, so three residue updates per candidate tile
    h.set_option("emulate_fp64", 1)
    try:
        h.set_data(c.X, c.Y)
        h.set_params(int(c.kernel), int(c.ard), float(c.variance), c.lengthscale, noise)
        h.fit()
        h.set_candidates(c.Xs)
        mu, var = h.predict(True)
        _, var0 = h.predict(False)
        if noise >= 1e-4:
            assert relmax(mu, c.mu) < 1e-6
            assert np.max(np.abs(var - c.var) / np.abs(c.var)) < 1e-6
            assert np.max(np.abs(var0 - c.var_noiseless)) < 1e-6 * float(c.variance)
            a = h.acq(_lib.GP_ACQ_EI, 0.01, float(c.fmin))
            assert np.max(np.abs(a - c.neg_EI)) <= 1e-6 * np.max(np.abs(c.neg_EI))
        else:   # stress: against the extended-precision truth, like the fp64 path (test_gpu_parity.py)
            T = np.load(os.path.join(HERE, "golden", "gp_truth.npz"))
            t = Case(T, tag)
            for hip, ref, truth, scale in ((mu, c.mu, t.mu, float(np.max(np.abs(t.mu)))),
                                           (var / t.var, c.var / t.var, t.var / t.var, 1.0),
                                           (var0, c.var_noiseless, t.var_noiseless, float(c.variance))):
                e_hip = float(np.max(np.abs(hip - truth))) / scale
                e_ref = float(np.max(np.abs(ref - truth))) / scale
                assert e_hip <= max(1e-6, 4 * e_ref), (e_hip, e_ref)
        # against the true-fp64 device path: the residue arithmetic is exact, only the operands are rounded to
        # fixed point (one ulp of the largest entry)
        h.set_option("emulate_fp64", 0)
        mu2, var2 = h.predict(True)
        assert np.max(np.abs(mu - mu2)) <= (1e-9 if noise >= 1e-4 else 1e-6) * max(1.0, np.max(np.abs(mu2)))
        assert np.max(np.abs(var - var2)) <= (1e-9 if noise >= 1e-4 else 1e-6) * float(c.variance)
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


@pytest.mark.parametrize("N,D,M,pt,variance", [(1500, 3, 700, 2, 1.7), (2048, 5, 300, 4, 0.3), (1100, 2, 129, 3, 40.0),
                                               (900, 4, 1, 1, 1.0), (3000, 8, 1000, 6, 1.0), (2500, 6, 333, 7, 2.5)])
def test_emulated_equals_fp64_path_on_ragged_shapes(h, N, D, M, pt, variance):
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=N + D)
    h.set_option("panel_tiles", pt)
    try:
        h.set_data(X, Y)
        h.set_params(1, 0, variance, [0.7], 1e-2)
        h.fit()
        h.set_candidates(Xs)
        m0, v0 = h.predict(True)
        h.set_option("emulate_fp64", 1)
        m1, v1 = h.predict(True)
        assert np.max(np.abs(m1 - m0)) <= 1e-10 * max(1.0, np.max(np.abs(m0)))
        assert np.max(np.abs(v1 - v0)) <= 1e-10 * variance
        gp = O.OracleGP(X, Y, O.Matern52(D, variance, 0.7), 1e-2)
        mo, vo = gp.predict(Xs)
        assert relmax(m1, mo) < 1e-6 and np.max(np.abs(v1 - vo) / vo) < 1e-6
        # the one-call entry point falls back to (emulated) fit + emulated predict: the same numbers as the two calls
        h.fit()
        m1b, v1b = h.predict(True)
        assert np.max(np.abs(m1b - m1)) <= 1e-9 * max(1.0, np.max(np.abs(m1))) and np.max(np.abs(v1b - v1)) <= 1e-9 * variance
        f1, m2, v2 = h.fit_predict(True)
        assert np.array_equal(m2, m1b) and np.array_equal(v2, v1b)
        # the number of panels one residue launch contracts changes the launches, not the integers they sum
        for grp in (1, 3):
            h.set_option("rns_group", grp)
            m3, v3 = h.predict(True)
            assert np.array_equal(m3, m1b) and np.array_equal(v3, v1b), grp
        h.set_option("rns_group", 8)
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)
        h.set_option("rns_group", 8)


def test_emulated_wide_panels(h):
    """Panels wider than the 896 bytes whose int32 sums an f32 holds exactly: the fold of rns_reduce_f (csrc/rns.hip)
    brings any sum of a contraction of up to 8192 bytes into range first."""
    N, D, M = 3072, 4, 500
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=11)
    h.set_option("panel_tiles", 8)
    try:
        h.set_data(X, Y)
        h.set_params(0, 0, 1.3, [0.5], 1e-2)
        h.set_candidates(Xs)
        f0 = h.fit()
        m0, v0 = h.predict(True)
        h.set_option("emulate_fp64", 1)
        f1 = h.fit()
        m1, v1 = h.predict(True)
        assert abs(f1[0] - f0[0]) <= 1e-10 * abs(f0[0])
        assert np.max(np.abs(m1 - m0)) <= 1e-9 * max(1.0, np.max(np.abs(m0)))
        assert np.max(np.abs(v1 - v0)) <= 1e-9 * 1.3
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


def test_emulated_headline_configuration_full_size(h):
    """C3 (N=16384, D=8, M=10^4) with the emulated candidate solve: mean / variance of 64 candidates against the
    independent full-size oracle at 1e-6, and all 10^4 against the true-fp64 device path."""
    from test_gpu_fullsize import _oracle_full_size
    N, D, M = 16384, 8, 10000
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=1234)
    ls = O.default_lengthscale(D, False)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.0, ls, 1e-2)
    h.fit()
    h.set_candidates(Xs)
    m0, v0 = h.predict(True)
    h.set_option("emulate_fp64", 1)
    try:
        m1, v1 = h.predict(True)
        ph = {p["name"]: p["ms"] for p in h.phases()}
    finally:
        h.set_option("emulate_fp64", 0)
    assert np.max(np.abs(m1 - m0)) <= 1e-9 * np.max(np.abs(m0))
    assert np.max(np.abs(v1 - v0) / v0) <= 1e-9
    pick = np.unique(np.r_[np.linspace(0, M - 1, 62).astype(int), int(np.argmax(v1)), int(np.argmin(v1))])
    lml0, _, alpha0, mu_o, var_o = _oracle_full_size(O.RBF(D, 1.0, ls), X, Y, 1e-2, Xs[pick], "c3")
    assert np.max(np.abs(m1[pick] - mu_o)) <= 1e-6 * np.max(np.abs(mu_o))
    assert np.max(np.abs(v1[pick] - var_o) / var_o) <= 1e-6
    print("emulated candidate solve phases (ms):", ph)


@pytest.mark.parametrize("N,D,pt,noise,P", [(1500, 3, 2, 1e-2, 1), (2048, 5, 4, 1e-2, 2), (3000, 8, 6, 1e-2, 1), (2600, 4, 2, 1e-6, 1),
                                            (4096, 4, 6, 1e-4, 1)])
def test_emulated_trailing_update_of_the_factorisation(h, N, D, pt, noise, P):
    """emulate_fp64 also moves the Cholesky's trailing update (the bulk stream's launches) onto the int8 matrix cores:
    factor, LML, alpha and the posterior against the true-fp64 device path and the oracle."""
    X, Y, Xs = O.synthetic_problem(N, D, 200, seed=N)
    if P == 2:
        Y = np.c_[Y, np.sin(5 * X[:, :1])]
    h.set_option("panel_tiles", pt)
    try:
        h.set_data(X, Y)
        h.set_params(0, 0, 1.4, [0.25 * np.sqrt(D)], noise)
        h.set_candidates(Xs)
        f0 = h.fit()
        L0, a0 = h.chol(), h.alpha()
        m0, v0 = h.predict(True)
        h.set_option("emulate_fp64", 1)
        f1 = h.fit()
        L1, a1 = h.chol(), h.alpha()
        m1, v1 = h.predict(True)
        assert f1[2] == f0[2]
        tol = 1e-10 if noise >= 1e-4 else 1e-7      # the stress case amplifies one-ulp operand rounding by cond(Ky) ~ 1e9
        assert abs(f1[0] - f0[0]) <= max(tol, 1e-12) * abs(f0[0])
        assert np.max(np.abs(L1 - L0)) <= tol * np.max(np.abs(L0))
        assert np.max(np.abs(a1 - a0)) <= (1e-8 if noise >= 1e-4 else 1e-5) * np.max(np.abs(a0))
        assert np.max(np.abs(m1 - m0)) <= (1e-8 if noise >= 1e-4 else 1e-5) * max(1.0, np.max(np.abs(m0)))
        assert np.max(np.abs(v1 - v0)) <= (1e-8 if noise >= 1e-4 else 1e-6) * 1.4
        if noise >= 1e-4:
            gp = O.OracleGP(X, Y, O.RBF(D, 1.4, 0.25 * np.sqrt(D)), noise)
            p = gp.posterior
            assert abs(f1[0] - p["lml"]) <= 1e-8 * abs(p["lml"])
            assert np.max(np.abs(a1 - p["alpha"])) <= 1e-6 * np.max(np.abs(p["alpha"]))
            mo, vo = gp.predict(Xs)
            assert relmax(m1, mo) < 1e-6 and np.max(np.abs(v1 - vo) / vo) < 1e-6
        # gradients on top of an emulated factor
        g1 = h.lml_grad(1)
        h.set_option("emulate_fp64", 0)
        h.fit()
        g0 = h.lml_grad(1)
        sc = max(1.0, abs(g0[0]), abs(g0[2]), float(np.max(np.abs(g0[1]))))
        assert abs(g1[0] - g0[0]) <= 1e-6 * sc and abs(g1[2] - g0[2]) <= 1e-6 * sc and np.max(np.abs(g1[1] - g0[1])) <= 1e-6 * sc
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


@pytest.mark.parametrize("N,pt", [(4096, 2), (5000, 2), (6144, 4)])
def test_emulated_factorisation_does_not_depend_on_the_grouping(h, N, pt):
    """rns_group_fit panels per residue launch (near / mid / far launches on two streams, csrc/api.hip): the integers
    summed are the same, so the factor must be BITWISE the one of one launch per panel."""
    X, Y, Xs = O.synthetic_problem(N, 5, 64, seed=N + pt)
    h.set_option("panel_tiles", pt)
    h.set_option("emulate_fp64", 1)
    try:
        h.set_data(X, Y)
        h.set_params(1, 0, 1.1, [0.6], 1e-2)
        h.set_candidates(Xs)
        h.set_option("rns_group_fit", 1)
        f1 = h.fit()
        L1, a1 = h.chol(), h.alpha()
        for grp in (2, 3, 4, 8):
            h.set_option("rns_group_fit", grp)
            f = h.fit()
            assert f == f1, grp
            assert np.array_equal(h.chol(), L1), grp
            assert np.array_equal(h.alpha(), a1), grp
    finally:
        h.set_option("rns_group_fit", 8)
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


def test_emulated_row_products_at_the_bound(h):
    """The 14 moduli rest on |sum_k a_k b_k| <= max diag Ky for rows of L and of L^-1 k* (csrc/rns.hip).  Here the bound is
    attained: max diag Ky = 1 exactly (no slack in the power-of-two scale), a long lengthscale makes all rows of L nearly
    parallel (every accumulated product ~ Ky_rc ~ 1 = 2^102 in integer units), and the candidates include training points
    themselves (rows of L^-1 k* of norm ~ sqrt(variance))."""
    N, D, M = 2048, 3, 300
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=77)
    Xs[:100] = X[:100]
    noise = 1e-3
    variance = 1.0 - noise - 1e-8
    h.set_option("panel_tiles", 2)
    try:
        h.set_data(X, Y)
        h.set_params(0, 0, variance, [6.0], noise)
        h.set_candidates(Xs)
        f0 = h.fit()
        L0 = h.chol()
        m0, v0 = h.predict(True)
        assert float(np.min(np.abs(L0[:, 0]))) > 0.97          # rows nearly parallel to the first column
        h.set_option("emulate_fp64", 1)
        f1 = h.fit()
        L1 = h.chol()
        m1, v1 = h.predict(True)
        assert f1[2] == f0[2]
        # (cond(Ky) ~ 1e6..1e7 here: operand rounding of one ulp is amplified accordingly)
        assert abs(f1[0] - f0[0]) <= 1e-9 * abs(f0[0])
        assert np.max(np.abs(L1 - L0)) <= 1e-9
        assert np.max(np.abs(m1 - m0)) <= 1e-7 * max(1.0, np.max(np.abs(m0)))
        assert np.max(np.abs(v1 - v0)) <= 1e-8
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


def test_emulated_predict_falls_back_to_fp64_on_non_finite_candidates(h):
    """A NaN candidate cannot be put into fixed point: the chunk is solved again in true fp64, where the NaN stays in its
    own row as it does in the reference's dtrtrs (posterior.py:294)."""
    N, D, M = 1500, 3, 260
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=5)
    Xs[17, 1] = np.nan
    h.set_option("panel_tiles", 2)
    try:
        h.set_data(X, Y)
        h.set_params(1, 0, 1.0, [0.5], 1e-2)
        h.set_candidates(Xs)
        h.fit()
        m0, v0 = h.predict(True)
        h.set_option("emulate_fp64", 1)
        h.fit()
        m1, v1 = h.predict(True)
        assert np.isnan(m1[17]).all() and np.isnan(v1[17]).all()
        ok = np.ones(M, bool); ok[17] = False
        assert np.isfinite(m1[ok]).all() and np.isfinite(v1[ok]).all()
        assert np.max(np.abs(m1[ok] - m0[ok])) <= 1e-9 and np.max(np.abs(v1[ok] - v0[ok])) <= 1e-9
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


@pytest.mark.parametrize("N,D,pt,noise,ard", [(2048, 4, 2, 1e-2, 0), (2560, 6, 4, 1e-2, 1), (3000, 3, 6, 1e-4, 0), (1536, 2, 2, 1e-6, 0)])
def test_emulated_woodbury_inverse_and_hyper_gradients(h, N, D, pt, noise, ard):
    """emulate_fp64 also covers Ky^-1 (dtrtri + dpotri, linalg.py:127-145,193-214): the solve of the identity and the product
    W W^T in residue form, W = L^-T on its own fixed-point scale 2^(eS-1) >= 1 / sqrt(noise + 1e-8).  Ky^-1 and the LML
    gradients against the true-fp64 device path and, for the well-conditioned cases, Ky Ky^-1 = I through the oracle's Ky."""
    X, Y, Xs = O.synthetic_problem(N, D, 8, seed=N + pt)
    ls = (0.3 + 0.1 * np.arange(D)) if ard else [0.35 * np.sqrt(D)]
    h.set_option("panel_tiles", pt)
    try:
        h.set_data(X, Y)
        h.set_params(0, ard, 1.3, ls, noise)
        h.fit()
        W0 = h.woodbury_inv()
        g0 = h.lml_grad(D if ard else 1)
        h.set_option("emulate_fp64", 1)
        h.fit()
        W1 = h.woodbury_inv()
        ph = [p["name"] for p in h.phases()]
        assert any(n.startswith("potri_solve_emulated") for n in ph) and any(n.startswith("potri_lauum_emulated") for n in ph), ph
        g1 = h.lml_grad(D if ard else 1)
        fg = h.fit_grad(D if ard else 1)
        # operand rounding: one ulp of 2^(eS-1) ~ 1/sqrt(noise) per entry of W, 1/noise in Ky^-1
        tol = 1e-9 if noise >= 1e-4 else 1e-6
        assert np.max(np.abs(W1 - W0)) <= tol * np.max(np.abs(W0))
        assert np.allclose(W1, W1.T, rtol=0, atol=0)
        sc = max(1.0, abs(g0[0]), abs(g0[2]), float(np.max(np.abs(g0[1]))))
        gt = 1e-8 if noise >= 1e-4 else 1e-5
        assert abs(g1[0] - g0[0]) <= gt * sc and abs(g1[2] - g0[2]) <= gt * sc and np.max(np.abs(g1[1] - g0[1])) <= gt * sc
        assert fg[1][0] == g1[0] and fg[1][2] == g1[2] and np.array_equal(fg[1][1], g1[1])    # one call == two calls
        if noise >= 1e-4:
            kern = O.RBF(D, 1.3, np.asarray(ls, dtype=float), ARD=bool(ard))
            Ky = kern.K(X) + (noise + 1e-8) * np.eye(N)
            R = Ky[:256] @ W1 - np.eye(N)[:256]
            assert np.max(np.abs(R)) <= 1e-8 / noise * 1e-2
    finally:
        h.set_option("emulate_fp64", 0)
        h.set_option("panel_tiles", 6)


def test_emulated_fit_headline_configuration_full_size(h):
    """C3 with the factorisation's trailing update AND the candidate solve emulated: LML 1e-8, alpha / mean / variance
    1e-6 against the independent full-size oracle."""
    from test_gpu_fullsize import _oracle_full_size
    N, D, M = 16384, 8, 10000
    X, Y, Xs = O.synthetic_problem(N, D, M, seed=1234)
    ls = O.default_lengthscale(D, False)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.0, ls, 1e-2)
    h.set_candidates(Xs)
    h.set_option("emulate_fp64", 1)
    try:
        lml, logdet, jit = h.fit()
        alpha = h.alpha()
        m1, v1 = h.predict(True)
        ph = {p["name"]: round(p["ms"], 3) for p in h.phases()}
        f2, m2, v2 = h.fit_predict(True)
        assert f2[0] == lml and np.array_equal(m2, m1) and np.array_equal(v2, v1)
    finally:
        h.set_option("emulate_fp64", 0)
    pick = np.unique(np.r_[np.linspace(0, M - 1, 62).astype(int), int(np.argmax(v1)), int(np.argmin(v1))])
    lml0, logdet0, alpha0, mu_o, var_o = _oracle_full_size(O.RBF(D, 1.0, ls), X, Y, 1e-2, Xs[pick], "c3")
    assert abs(lml - lml0) <= 1e-8 * abs(lml0)
    assert abs(logdet - logdet0) <= 1e-8 * abs(logdet0)
    assert np.max(np.abs(alpha - alpha0)) <= 1e-6 * np.max(np.abs(alpha0))
    assert np.max(np.abs(m1[pick] - mu_o)) <= 1e-6 * np.max(np.abs(mu_o))
    assert np.max(np.abs(v1[pick] - var_o) / var_o) <= 1e-6
    print("emulated predict phases (ms):", ph)

"""GPU: parity of the HIP path (through the C-ABI) with the golden vectors and the live oracle.

Tolerances (BASELINE.json:north_star): posterior mean / variance within 1e-6 relative, LML within
1e-8 relative.  Bit-exactness is not expected: the Cholesky, the distance formula and every
reduction run in a different (blocked / fused) order than LAPACK + NumPy.
At noise 1e-6 (cond(Ky) ~ 1e8..1e9) both the reference's LAPACK result and the HIP result are measured against an
extended-precision truth (tests/golden/gp_truth.npz) instead of against each other.
"""
import numpy as np
import pytest

from conftest import Case, golden_tags, relmax, emulation_modes
from gaussian_process_optimization_amd import _lib
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu


def _tags():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gp_golden.npz"))
    return golden_tags(g)


@pytest.fixture(scope="module", params=emulation_modes())
def h(request):
    """One context per arithmetic mode: every test of this module runs in true fp64 AND with the bulk contractions
    emulated on the int8 matrix cores (same tolerances).  Contexts a test creates itself inherit the mode through the
    environment default (gp_create reads GPHIP_EMULATE_FP64)."""
    import os
    prev = os.environ.get("GPHIP_EMULATE_FP64")
    os.environ["GPHIP_EMULATE_FP64"] = str(request.param)
    hd = _lib.Handle(0)
    hd.set_option("emulate_fp64", request.param)
    yield hd
    hd.close()
    if prev is None:
        os.environ.pop("GPHIP_EMULATE_FP64", None)
    else:
        os.environ["GPHIP_EMULATE_FP64"] = prev


@pytest.mark.parametrize("tag", _tags())
def test_golden_case(golden, h, tag):
    """Every golden case against the reference's float64 values at the north-star tolerances (1e-8 LML, 1e-6
    everything else).  The stress cases (noise 1e-6, cond(Ky) 1e8..1e9) check structure and the LML here; their values
    are judged against the extended-precision truth in test_stress_case_against_extended_precision_truth, where the
    reference's own LAPACK result is measured too (comparing two float64 results that each carry cond * eps of error
    with one another would need tolerances chosen by assertion)."""
    c = Case(golden, tag)
    noise = float(c.noise)
    stress = noise < 1e-4
    tol = 1e-6
    h.set_data(c.X, c.Y)
    h.set_params(int(c.kernel), int(c.ard), float(c.variance), c.lengthscale, noise)
    K = h.kernel_matrix()
    assert relmax(K[c.rows], c.K_rows) < 1e-13
    assert np.allclose(K, K.T, rtol=0, atol=0)
    lml, logdet, jit = h.fit()
    assert jit == 0.0
    assert abs(lml - float(c.lml)) <= 1e-8 * abs(float(c.lml))
    assert abs(logdet - float(c.logdet)) <= 1e-10 * abs(float(c.logdet))
    L = h.chol()
    assert np.all(np.triu(L, 1) == 0)
    h.set_candidates(c.Xs)
    Wi = h.woodbury_inv()
    assert np.array_equal(Wi, Wi.T)
    f0 = float(c.fmin)
    acqs = ((_lib.GP_ACQ_EI, 0.01, "EI"), (_lib.GP_ACQ_LCB, 2.0, "LCB"), (_lib.GP_ACQ_MPI, 0.01, "MPI"))
    for t, par, name in acqs:   # device arg-best == numpy's first extremum of the device's own scores
        a = h.acq(t, par, f0)
        for sense, fn in ((-1, np.argmin), (+1, np.argmax)):
            idx, val = h.acq_argbest(t, par, f0, sense)
            assert idx == int(fn(a[:, 0])) and val == a[idx, 0]
    if stress:
        return
    assert relmax(L[c.rows], c.L_rows) < tol
    assert relmax(np.diag(L), c.L_diag) < tol
    assert relmax(h.alpha(), c.alpha) < tol
    mu, var = h.predict(True)
    assert relmax(mu, c.mu) < tol
    assert np.max(np.abs(var - c.var) / np.abs(c.var)) < tol
    mu0, var0 = h.predict(False)
    assert relmax(mu0, c.mu) < tol
    assert np.max(np.abs(var0 - c.var_noiseless)) < tol * float(c.variance)
    if "cov_full" in [k.split("/")[1] for k in golden.files if k.startswith(tag + "/")]:
        _, cov = h.predict_full_cov(True)
        assert relmax(cov, c.cov_full) < tol
    # gradients of the LML (natural space)
    dv, dl, dn = h.lml_grad(c.lengthscale.size)
    scale = max(abs(float(c.dvariance)), np.max(np.abs(c.dlengthscale)), 1.0)
    assert abs(dv - float(c.dvariance)) < tol * scale
    assert np.max(np.abs(dl - c.dlengthscale)) < tol * scale
    assert abs(dn - float(c.dnoise)) < tol * max(abs(float(c.dnoise)), 1.0)
    assert relmax(Wi[c.rows], c.Wi_rows) < tol
    # predictive gradients, fmin, acquisitions (+ gradients), the reference's winner
    dm, dvx = h.predict_grad()
    assert relmax(dm, c.dmdx) < tol
    assert relmax(dvx, c.dvdx) < tol
    fmin = h.fmin()
    assert abs(fmin - float(c.fmin)) < tol * max(1.0, abs(float(c.fmin)))
    for t, par, name in acqs:
        ref = getattr(c, "neg_" + name)
        atol = tol * max(np.max(np.abs(ref)), 1e-300)
        a2, da = h.acq_grad(t, par, f0)
        assert np.max(np.abs(h.acq(t, par, f0) - ref)) <= atol and np.max(np.abs(a2 - ref)) <= atol
        dref = getattr(c, "neg_d" + name)
        assert np.max(np.abs(da - dref)) <= 1e-5 * max(np.max(np.abs(dref)), 1e-300)   # d/dx of a tail probability
        idx, _ = h.acq_argbest(t, par, f0, -1)
        ir = int(getattr(c, "argmin_" + name))   # same winner as the reference unless the top two are within tolerance
        assert idx == ir or abs(ref[idx, 0] - ref[ir, 0]) <= 2 * atol


def _truth_tags():
    import os
    t = np.load(os.path.join(os.path.dirname(__file__), "golden", "gp_truth.npz"))
    return sorted({k.split("/")[0] for k in t.files})


@pytest.mark.parametrize("tag", _truth_tags())
def test_stress_case_against_extended_precision_truth(golden, h, tag):
    """noise = 1e-6 (cond(Ky) 1e8..1e9): the comparisons of test_golden_case at 1e-5 .. 2e-3 are between TWO float64
    results that both carry cond * eps of error.  Here both are measured against tests/golden/gp_truth.npz (the same
    formulas on the same inputs in 80-bit arithmetic, generate_truth.py; itself within 2e-13 of 60-digit arithmetic)
    and the HIP result has to be as close to the truth as the north-star tolerance (1e-8 LML, 1e-6 everything else) OR
    as close as the reference's own LAPACK path gets -- within a factor 4, since the blocked factorisation with
    inverted tiles has other error constants than LAPACK's substitutions.  Measured errors are written to
    gpurun_out/stress_truth.json for DESIGN.md."""
    import json
    import os
    T = np.load(os.path.join(os.path.dirname(__file__), "golden", "gp_truth.npz"))
    t = Case(T, tag)
    c = Case(golden, tag)
    noise = float(c.noise)
    assert noise == 1e-6
    h.set_data(c.X, c.Y)
    h.set_params(int(c.kernel), int(c.ard), float(c.variance), c.lengthscale, noise)
    lml, logdet, jit = h.fit()
    assert jit == 0.0
    h.set_candidates(c.Xs)
    mu, var = h.predict(True)
    _, var0 = h.predict(False)
    dv, dl, dn = h.lml_grad(c.lengthscale.size)
    Wi = h.woodbury_inv()
    dm, dvx = h.predict_grad()
    fmin = h.fmin()
    f0 = float(t.fmin)
    acq = {}
    for typ, par, name in ((_lib.GP_ACQ_EI, 0.01, "EI"), (_lib.GP_ACQ_LCB, 2.0, "LCB"), (_lib.GP_ACQ_MPI, 0.01, "MPI")):
        acq[name] = h.acq_grad(typ, par, f0)

    def sc(x):
        return max(float(np.max(np.abs(x))), 1e-300)
    vscale = float(c.variance)
    gscale = max(abs(float(t.dvariance)), sc(t.dlengthscale), 1.0)
    # (name, hip, reference (golden through LAPACK), truth, scale the error is relative to, north-star tolerance)
    rows = [
        ("lml", lml, float(c.lml), float(t.lml), abs(float(t.lml)), 1e-8),
        ("logdet", logdet, float(c.logdet), float(t.logdet), abs(float(t.logdet)), 1e-8),
        ("alpha", h.alpha(), c.alpha, t.alpha, sc(t.alpha), 1e-6),
        ("L_diag", np.diag(h.chol()), c.L_diag, t.L_diag, sc(t.L_diag), 1e-6),
        ("mean", mu, c.mu, t.mu, sc(t.mu), 1e-6),
        ("var", var / t.var, c.var / t.var, t.var / t.var, 1.0, 1e-6),
        ("var_noiseless", var0, c.var_noiseless, t.var_noiseless, vscale, 1e-6),
        ("dvariance", dv, float(c.dvariance), float(t.dvariance), gscale, 1e-6),
        ("dlengthscale", dl, c.dlengthscale, t.dlengthscale, gscale, 1e-6),
        ("dnoise", dn, float(c.dnoise), float(t.dnoise), max(abs(float(t.dnoise)), 1.0), 1e-6),
        ("Wi_rows", Wi[c.rows], c.Wi_rows, t.Wi_rows, float(t.Wi_absmax), 1e-6),
        ("dmdx", dm, c.dmdx, t.dmdx, sc(t.dmdx), 1e-6),
        ("dvdx", dvx, c.dvdx, t.dvdx, vscale / float(np.min(c.lengthscale)), 1e-6),
        ("fmin", fmin, float(c.fmin), float(t.fmin), max(1.0, abs(float(t.fmin))), 1e-6),
    ]
    # LCB is linear in the posterior: judged against the truth like the posterior itself.  EI and MPI sit ~11 sigma out in
    # the Gaussian tail in these cases (u ~ -11): d ln EI / du ~ -u, so a relative error d of the posterior arrives
    # amplified by ~u^2 ~ 100; their truth errors are RECORDED (report) and the kernels are judged on what they compute:
    # the reference's formulas (general.py:113-129, EI.py:32-51, MPI.py:32-51) on the device's OWN mean / variance /
    # gradients, which are themselves held to the truth above.
    rows.append(("neg_LCB", acq["LCB"][0], c.neg_LCB, t.neg_LCB, sc(t.neg_LCB), 1e-6))
    rows.append(("neg_dLCB", acq["LCB"][1], c.neg_dLCB, t.neg_dLCB, sc(t.neg_dLCB), 1e-6))
    sdev = np.sqrt(np.clip(var, 1e-10, np.inf))
    dsdx = dvx / (2 * sdev)
    phi, Phi, u = O.get_quantiles(0.01, f0, mu, sdev.copy())
    own = {"EI": (-(sdev * (u * Phi + phi)), -(dsdx * phi - Phi * dm[:, :, 0])),
           "MPI": (-Phi, (phi / sdev) * (dm[:, :, 0] + dsdx * u))}
    recorded = {}
    for name in ("EI", "MPI"):
        for k, (dev_val, own_val) in enumerate(zip(acq[name], own[name])):
            key = ("neg_" if k == 0 else "neg_d") + name
            scale = max(sc(own_val), 1e-300)
            assert np.max(np.abs(dev_val - own_val)) <= 1e-9 * scale, key
            truth = getattr(t, key)
            recorded[key] = {"hip": float(np.max(np.abs(dev_val - truth))) / sc(truth),
                             "lapack_reference": float(np.max(np.abs(getattr(c, key) - truth))) / sc(truth),
                             "north_star_tol": None, "amplification_u2": float(np.max(u * u))}
    # the DECISION these cases lead to (anchor_points_generator.py:59-61, run.py:1241): the row the device's arg-best
    # picks, fed by the device's own fmin as in a live loop, against the row the reference picks (fixture argmin_*).
    # Where they differ the reference's own scores must not be able to tell the two rows apart: their gap is below the
    # distance of either float64 path to the extended-precision truth.  Agreement is recorded per case and acquisition
    # (profiles/r05_stress_decisions.txt is the tally).
    decisions = {}
    for typ, par, name in ((_lib.GP_ACQ_EI, 0.01, "EI"), (_lib.GP_ACQ_LCB, 2.0, "LCB"), (_lib.GP_ACQ_MPI, 0.01, "MPI")):
        idx, _ = h.acq_argbest(typ, par, fmin, -1)
        ir = int(getattr(c, "argmin_" + name))
        ref, tr = getattr(c, "neg_" + name)[:, 0], getattr(t, "neg_" + name)[:, 0]
        dev = h.acq(typ, par, fmin)[:, 0]
        e_hip, e_ref = float(np.max(np.abs(dev - tr))), float(np.max(np.abs(ref - tr)))
        gap = float(abs(ref[idx] - ref[ir]))
        decisions[name] = {"device_row": int(idx), "reference_row": ir, "truth_row": int(np.argmin(tr)), "agree": idx == ir,
                           "reference_gap": gap, "hip_to_truth": e_hip, "lapack_to_truth": e_ref}
        assert idx == ir or gap <= min(e_hip, e_ref), (name, decisions[name])
    report, bad = dict(recorded), []
    report["decisions"] = decisions
    for name, hip, ref, truth, scale, tol in rows:
        e_hip = float(np.max(np.abs(np.asarray(hip, dtype=float) - truth))) / scale
        e_ref = float(np.max(np.abs(np.asarray(ref, dtype=float) - truth))) / scale
        report[name] = {"hip": e_hip, "lapack_reference": e_ref, "north_star_tol": tol}
        # two conditions: not worse than 4x LAPACK's own error where LAPACK itself misses the north-star tolerance, AND an
        # absolute ceiling per quantity whatever LAPACK does -- the north-star tolerance itself for every quantity but d(LCB)/dx,
        # where the reference's own float64 result is 6e-6 off the truth (measured worst cases over the 40 cases, both arithmetic
        # modes: profiles/r03_stress_truth.json; the largest is alpha at 5e-8)
        ceiling = {"neg_dLCB": 3e-5}.get(name, tol)
        if not (e_hip <= max(tol, 4.0 * e_ref) and e_hip <= ceiling):
            bad.append((name, e_hip, e_ref))
    for key, rec in recorded.items():   # EI / MPI ~11 sigma out in the tail: recorded above, with a loose backstop (measured 8e-5)
        if not rec["hip"] <= 1e-3:
            bad.append((key, rec["hip"], rec["lapack_reference"]))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "stress_truth.json")
        allr = json.load(open(path)) if os.path.exists(path) else {}
        allr[tag] = report
        json.dump(allr, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert not bad, bad


def test_multi_output_with_normalizer(golden):
    import gaussian_process_optimization_amd as gpo
    m = gpo.models.GPRegression(golden["multi/X"], golden["multi/Y"], gpo.kern.RBF(3, 0.9, [0.3, 0.5, 0.7], ARD=True),
                                normalizer=True, noise_var=0.02)
    mu, var = m.predict(golden["multi/Xs"])
    assert relmax(mu, golden["multi/mu"]) < 1e-6
    assert relmax(var, golden["multi/var"]) < 1e-6
    assert abs(m.log_likelihood() - float(golden["multi/lml"])) <= 1e-8 * abs(float(golden["multi/lml"]))
    assert relmax(m.posterior.woodbury_vector, golden["multi/alpha"]) < 1e-6
    g = m.gradient
    ref = np.r_[golden["multi/dvariance"], golden["multi/dlengthscale"], golden["multi/dnoise"]]
    assert np.max(np.abs(g - ref)) < 1e-6 * np.max(np.abs(ref))
    m.close()


def test_jitter_ladder(golden, h):
    """The reference's ladder (linalg.py:62-75): un-jittered dpotrf, then mean(diag)*1e-6 * 10^k."""
    for name in ("jit1", "jit3"):
        h.set_data(golden[name + "/X"], golden[name + "/Y"])
        h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden[name + "/noise"]))
        lml, logdet, jit = h.fit(5)
        assert jit == pytest.approx(float(golden[name + "/jitter"]), rel=1e-12)
        assert logdet == pytest.approx(float(golden[name + "/logdet"]), rel=1e-4)
    h.set_data(golden["jitfail/X"], golden["jitfail/Y"])
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jitfail/noise"]))
    with pytest.raises(np.linalg.LinAlgError, match="even with jitter"):
        h.fit(5)
    # maxtries bounds the ladder (linalg_test.py:18-37: succeeds with enough tries, raises with fewer)
    h.set_data(golden["jit3/X"], golden["jit3/Y"])
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jit3/noise"]))
    with pytest.raises(np.linalg.LinAlgError):
        h.fit(2)
    h.fit(3)
    # non-positive diagonal (linalg.py:63-64)
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], -2.0)
    with pytest.raises(np.linalg.LinAlgError, match="non-positive diagonal"):
        h.fit(5)


@pytest.mark.parametrize("N,D,M,P", [(1, 1, 1, 1), (2, 3, 5, 1), (127, 2, 3, 2), (128, 2, 128, 1), (129, 4, 129, 1),
                                     (700, 5, 257, 2), (300, 17, 40, 1), (1500, 3, 33, 1)])
@pytest.mark.parametrize("kname", ["rbf", "Mat52"])
def test_live_oracle_ragged_shapes(h, N, D, M, P, kname):
    rng = np.random.default_rng(N * 31 + D)
    X = rng.uniform(0, 1, (N, D))
    Y = rng.standard_normal((N, P))
    Xs = rng.uniform(-0.1, 1.1, (M, D))
    ard = D > 1
    ls = rng.uniform(0.3, 1.2, D) if ard else np.array([0.6])
    kern = O.make_kernel(kname, D, 1.7, ls, ARD=ard)
    gp = O.OracleGP(X, Y, kern, 0.05)
    h.set_data(X, Y)
    h.set_params(0 if kname == "rbf" else 1, ard, 1.7, ls, 0.05)
    lml, logdet, jit = h.fit()
    p = gp.posterior
    assert abs(lml - p["lml"]) <= 1e-8 * max(1.0, abs(p["lml"]))
    assert relmax(h.alpha(), p["alpha"]) < 1e-6
    h.set_candidates(Xs)
    mu, var = h.predict(True)
    m0, v0 = gp.predict(Xs)
    assert relmax(mu, m0) < 1e-6 and np.max(np.abs(var - v0) / np.abs(v0)) < 1e-6
    dv, dl, dn = h.lml_grad(ls.size)
    r = gp.gradients()
    sc = max(1.0, abs(r[0]), np.max(np.abs(r[1])))
    assert abs(dv - r[0]) < 1e-6 * sc and np.max(np.abs(dl - r[1])) < 1e-6 * sc and abs(dn - r[2]) < 1e-6 * max(1, abs(r[2]))
    dm, dvx = h.predict_grad()
    dm0, dv0 = gp.predictive_gradients(Xs)
    assert np.max(np.abs(dm - dm0)) < 1e-6 * max(1.0, np.max(np.abs(dm0)))
    assert np.max(np.abs(dvx - dv0)) < 1e-6 * max(1.0, np.max(np.abs(dv0)))


def test_near_duplicate_rows_matern(h):
    """Matern-5/2 depends on r, not r^2: near-duplicate inputs are where the reference's Gram-trick
    distance is least accurate ('Distance formula')."""
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (200, 3))
    X[100:] = X[:100] + 1e-9 * rng.standard_normal((100, 3))
    Y = np.sin(X.sum(1, keepdims=True))
    kern = O.Matern52(3, 1.0, 0.5)
    gp = O.OracleGP(X, Y, kern, 1e-2)
    h.set_data(X, Y)
    h.set_params(1, 0, 1.0, [0.5], 1e-2)
    lml, _, _ = h.fit()
    assert abs(lml - gp.log_likelihood()) <= 1e-8 * abs(gp.log_likelihood())
    Xs = rng.uniform(0, 1, (50, 3))
    h.set_candidates(Xs)
    mu, var = h.predict(True)
    m0, v0 = gp.predict(Xs)
    assert relmax(mu, m0) < 1e-6 and np.max(np.abs(var - v0) / v0) < 1e-6


def test_argbest_tie_breaks_to_lowest_index(h):
    rng = np.random.default_rng(8)
    X = rng.uniform(0, 1, (50, 2)); Y = rng.standard_normal((50, 1))
    h.set_data(X, Y)
    h.set_params(0, 0, 1.0, [0.4], 0.1)
    h.fit()
    Xs = rng.uniform(0, 1, (3000, 2))
    Xs[2500] = Xs[7]; Xs[1200] = Xs[7]
    h.set_candidates(Xs)
    a = h.acq(_lib.GP_ACQ_LCB, 2.0, 0.0)[:, 0]
    assert a[7] == a[1200] == a[2500]
    # force the tie to be the winner by scoring only duplicates of row 7 plus worse points
    worst = np.argsort(a)[-500:]
    Xt = np.vstack([Xs[worst], Xs[[7]], Xs[worst[:5]], Xs[[7]]])
    h.set_candidates(Xt)
    at = h.acq(_lib.GP_ACQ_LCB, 2.0, 0.0)[:, 0]
    for sense, fn in ((-1, np.argmin), (+1, np.argmax)):
        idx, val = h.acq_argbest(_lib.GP_ACQ_LCB, 2.0, 0.0, sense)
        assert idx == int(fn(at)) and val == at[idx]


def test_refit_is_bitwise_reproducible(h):
    X, Y, Xs = O.synthetic_problem(900, 4, 100, seed=9)
    h.set_data(X, Y)
    h.set_params(0, 0, 1.0, [0.5], 1e-2)
    h.set_candidates(Xs)
    r1 = (h.fit(), h.alpha(), h.predict(True))
    r2 = (h.fit(), h.alpha(), h.predict(True))
    assert r1[0] == r2[0] and np.array_equal(r1[1], r2[1])
    assert np.array_equal(r1[2][0], r2[2][0]) and np.array_equal(r1[2][1], r2[2][1])
    # panel width and look-ahead are schedule choices: the per-tile arithmetic only changes its summation grouping
    for pt, la in ((1, 1), (2, 1), (3, 0), (4, 1), (8, 0)):
        h.set_option("panel_tiles", pt)
        h.set_option("lookahead", la)
        lml = h.fit()[0]
        assert abs(lml - r1[0][0]) <= 1e-11 * abs(r1[0][0])
    h.set_option("panel_tiles", 8)
    h.set_option("lookahead", 1)


def test_state_errors(h):
    hd = _lib.Handle(0)
    with pytest.raises(RuntimeError):
        hd.fit()
    hd.set_data(np.zeros((3, 2)), np.zeros((3, 1)))
    with pytest.raises(RuntimeError):
        hd.fit()
    with pytest.raises(ValueError):
        hd.set_params(7, 0, 1.0, [1.0], 0.1)
    with pytest.raises(ValueError):
        hd.set_data(np.zeros((3, 65)), np.zeros((3, 1)))
    hd.close()


@pytest.mark.parametrize("stages,start", [(0, 40), (1, 0), (2, 60), (1 << 20, 0), (1 << 20, 70)])
@pytest.mark.parametrize("N,M,pt", [(1500, 700, 2), (2048, 300, 4), (1100, 129, 3), (600, 50, 8)])
def test_fit_predict_pipelined_equals_separate_calls(h, N, M, pt, stages, start):
    """gp_fit_predict runs the first `pipe_stages` candidate stages behind the factorisation (released at
    `pipe_start_pct` % of the panels) and the rest after it; same arithmetic, bitwise the same results."""
    X, Y, Xs = O.synthetic_problem(N, 5, M, seed=N)
    h.set_option("panel_tiles", pt)
    h.set_option("pipe_stages", stages)
    h.set_option("pipe_start_pct", start)
    h.set_data(X, Y)
    h.set_params(1, 0, 1.2, [0.7], 1e-2)
    h.set_candidates(Xs)
    f0 = h.fit()
    m0, v0 = h.predict(True)
    a0 = h.alpha()
    f1, m1, v1 = h.fit_predict(True)
    assert f1 == f0
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    assert np.array_equal(a0, h.alpha())
    # follow-up calls see a fitted, predicted state
    e0 = h.acq(_lib.GP_ACQ_LCB, 2.0, 0.0)
    h.fit(); h.predict(True)
    assert np.array_equal(e0, h.acq(_lib.GP_ACQ_LCB, 2.0, 0.0))
    gp = O.OracleGP(X, Y, O.Matern52(5, 1.2, 0.7), 1e-2)
    mo, vo = gp.predict(Xs)
    assert relmax(m1, mo) < 1e-6 and np.max(np.abs(v1 - vo) / vo) < 1e-6
    h.set_option("panel_tiles", 6)
    h.set_option("pipe_stages", 0)
    h.set_option("pipe_start_pct", -1)


def test_fit_predict_jitter_and_failure(golden, h):
    h.set_data(golden["jit3/X"], golden["jit3/Y"])
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jit3/noise"]))
    h.set_candidates(golden["jit3/X"][:10])
    (lml, logdet, jit), m, v = h.fit_predict(True)
    assert jit == pytest.approx(float(golden["jit3/jitter"]), rel=1e-12)
    h.set_params(_lib.GP_KERNEL_RBF, 0, 1.0, [0.5], float(golden["jitfail/noise"]))
    with pytest.raises(np.linalg.LinAlgError):
        h.fit_predict(True)


@pytest.mark.parametrize("noise", [1e-2, 1e-6])
def test_fmin_identity_equals_direct_product(h, noise):
    """gp_fmin takes min(y - d alpha) (normal equations, O(N)); the reference's get_fmin (gpmodel.py:138-142) takes
    min(K(X,X) alpha).  Both are checked against the oracle and against each other."""
    rng = np.random.default_rng(5)
    N, D = 900, 3
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(6 * X.sum(1, keepdims=True)) + 0.05 * rng.standard_normal((N, 1))
    kern = O.make_kernel("Mat52", D, 1.3, np.array([0.4]), ARD=False)
    gp = O.OracleGP(X, Y, kern, noise)
    f0 = O.OracleGPModel(gp).get_fmin()
    h.set_data(X, Y)
    h.set_params(1, 0, 1.3, [0.4], noise)
    h.fit()
    f_id = h.fmin()
    h.set_option("fmin_direct", 1)
    try:
        f_dir = h.fmin()
    finally:
        h.set_option("fmin_direct", 0)
    assert abs(f_id - f_dir) <= 1e-9 * max(1.0, abs(f_dir))
    assert abs(f_id - f0) <= 1e-6 * max(1.0, abs(f0))


@pytest.mark.parametrize("stages,start", [(0, 40), (1, 0), (3, 70), (1 << 20, 0)])
@pytest.mark.parametrize("N,D,pt,ard", [(1500, 3, 2, True), (2048, 5, 4, False), (1100, 2, 3, True), (300, 4, 6, True)])
def test_fit_grad_equals_separate_calls(h, N, D, pt, ard, stages, start):
    """gp_fit_grad = gp_fit + gp_lml_grad with the first stages of the solve for L^-T behind the factorisation;
    bitwise the same LML and gradients, and the gradients match the oracle (N = 300: the unpipelined fallback)."""
    rng = np.random.default_rng(N + D)
    X = rng.uniform(0, 1, (N, D))
    Y = np.sin(3 * X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((N, 1))
    ls = rng.uniform(0.4, 0.9, D) if ard else np.array([0.6])
    h.set_option("panel_tiles", pt)
    h.set_option("pipe_stages_grad", stages)
    h.set_option("pipe_start_pct_grad", start)
    try:
        h.set_data(X, Y)
        h.set_params(1, ard, 1.3, ls, 0.02)
        f0 = h.fit()
        g0 = h.lml_grad(ls.size)
        Wi0 = h.woodbury_inv()
        f1, g1 = h.fit_grad(ls.size)
        assert f1 == f0
        assert g1[0] == g0[0] and np.array_equal(g1[1], g0[1]) and g1[2] == g0[2]
        assert np.array_equal(Wi0, h.woodbury_inv())
        gp = O.OracleGP(X, Y, O.make_kernel("Mat52", D, 1.3, ls, ARD=ard), 0.02)
        r = gp.gradients()
        sc = max(1.0, abs(r[0]), np.max(np.abs(r[1])), abs(r[2]))
        assert abs(g1[0] - r[0]) < 1e-6 * sc and np.max(np.abs(g1[1] - r[1])) < 1e-6 * sc and abs(g1[2] - r[2]) < 1e-6 * sc
        # a following prediction sees the fitted state
        h.set_candidates(X[:7])
        mu, var = h.predict(True)
        m0, v0 = gp.predict(X[:7])
        assert relmax(mu, m0) < 1e-6 and np.max(np.abs(var - v0) / v0) < 1e-6
    finally:
        h.set_option("panel_tiles", 6)
        h.set_option("pipe_stages_grad", 0)
        h.set_option("pipe_start_pct_grad", 40)

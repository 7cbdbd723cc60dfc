// Candidates: posterior mean / variance, acquisitions, arg-best / top-k, local penalisation, full covariance and
// posterior samples.  Reference: PosteriorExact._raw_predict (posterior.py:273-302), GPyOpt acquisitions/{EI,LCB,MPI,LP}.py.
#include "api_internal.h"

// ---- candidates / predict -----------------------------------------------------------------------
extern "C" int gp_set_candidates(gp_t *g, const double *Xs, int64_t M) {
    if (!g || !Xs) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->have_data) return fail(GP_ERR_STATE, "gp_set_data first");
    if (M < 1) return fail(GP_ERR_ARG, "M < 1");
    HIPCHK(hipSetDevice(g->device));
    GP_SYNC(g->s);
    int rc;
    if ((rc = dev_realloc(&g->dXs, &g->capM, (long)M * g->D))) return rc;
    HIPCHK(hipMemcpy(g->dXs, Xs, sizeof(double) * M * g->D, hipMemcpyHostToDevice));
    g->M = M;
    g->predicted = false;
    return 0;
}

int run_predict(gp_ctx *g, int include_noise, bool tiles_only) {
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, N = g->N, Npad = g->Npad;
    const int P = g->P;
    int rc;
    g->nphases = 0;
    const long mc_max = std::min(g->mc_max, round_up(M, GP_TILE));
    if ((rc = ensure_panel_inv(g))) return rc;
    if ((rc = dev_realloc(&g->dT, &g->capT, mc_max * Npad))) return rc;
    if ((rc = dev_realloc(&g->dT2, &g->capT2, mc_max * Npad))) return rc;
    g->w_in_t2 = false;   // dT2 takes the solved candidate rows
    if (M <= g->small_m && !tiles_only) {
        // A handful of rows (the acquisition optimiser's one-row calls): the solve as matrix-vector work bound by ONE read of
        // L (smallm.hip) instead of ~45 dependent tile launches; always true fp64.
        int ph = phase_begin(g, "cross_k", 0.0, 8.0 * (double)(N + M) * g->D + 8.0 * (double)N * M);
        launch_cross_k_rows(g->s, g->dT, Npad, g->dXs, (int)M, g->dX, g->N, Npad, g->kp);
        phase_end(g, ph);
        ph = phase_begin(g, "cand_solve_rows", (double)N * N * M, 8.0 * (double)N * N / 2);
        launch_small_forward_solve(g->s, g->dA, Npad, g->dInvP, g->invp_W, Npad, g->dT, g->dT2, Npad, (int)M);
        phase_end(g, ph);
        ph = phase_begin(g, "reduce", 0.0, 8.0 * (double)N * M);
        launch_predict_reduce(g->s, g->dT2, Npad, M, N, g->dA + Npad * Npad, Npad, P, g->kp.variance,
                              include_noise ? g->noise : 0.0, g->dMean, g->dVar);
        phase_end(g, ph);
        g->predicted = true;
        g->predicted_noise = include_noise ? 1 : 0;
        return 0;
    }
    for (long m0 = 0; m0 < M; m0 += mc_max) {
        const long mc = std::min(mc_max, M - m0);
        const long mcpad = round_up(mc, GP_TILE);
        int ph = phase_begin(g, "cross_k", 0.0, 8.0 * (double)(N + mc) * g->D + 8.0 * (double)N * mc);
        launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, g->N, Npad, g->kp);
        phase_end(g, ph);
        ph = phase_begin(g, g->emulate_fp64 ? "cand_solve_emulated" : "cand_solve", (double)N * N * mc, 0.0);
        if (g->emulate_fp64) {
            rc = solve_rows_rns(g, g->dT, g->dT2, (int)(mcpad / GP_TILE));
            if (rc == GP_ERR_RANGE) {   // non-finite candidates / factor: this chunk again in true fp64 (NaNs propagate as in the reference)
                ++g->emu_fallbacks;
                launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, g->N, Npad, g->kp);
                solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), 0);
            } else if (rc) {
                return rc;
            }
        } else {
            solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), 0);
        }
        phase_end(g, ph);
        ph = phase_begin(g, "reduce", 0.0, 8.0 * (double)N * mc);
        launch_predict_reduce(g->s, g->dT2, Npad, mc, N, g->dA + Npad * Npad, Npad, P, g->kp.variance,
                              include_noise ? g->noise : 0.0, g->dMean + m0 * P, g->dVar + m0);
        phase_end(g, ph);
    }
    g->predicted = true;
    g->predicted_noise = include_noise ? 1 : 0;
    return 0;
}

int ensure_out(gp_ctx *g) {
    const long need = g->M * (long)std::max(1, g->P);
    if (g->dMean && g->dVar && g->dAcq && g->capOut >= need) return 0;
    for (double **b : {&g->dMean, &g->dVar, &g->dAcq}) {
        if (*b) hipFree(*b);
        *b = nullptr;
    }
    HIPCHK(hipMalloc((void **)&g->dMean, sizeof(double) * need));
    HIPCHK(hipMalloc((void **)&g->dVar, sizeof(double) * need));
    HIPCHK(hipMalloc((void **)&g->dAcq, sizeof(double) * need));
    g->capOut = need;
    return 0;
}

extern "C" int gp_predict(gp_t *g, int include_noise, double *mean, double *var) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise))) return rc;
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * g->M * g->P, hipMemcpyDeviceToHost, g->s));
    if (var) HIPCHK(hipMemcpyAsync(var, g->dVar, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

extern "C" int gp_fmin(gp_t *g, double *fmin) {
    if (!g || !fmin) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->P != 1) return fail(GP_ERR_ARG, "gp_fmin needs P == 1");
    HIPCHK(hipSetDevice(g->device));
    if (!g->fmin_valid) {
        if (g->fmin_direct)
            launch_train_mean(g->s, g->dX, g->N, g->kp, g->dAlpha, g->dMu);
        else
            launch_train_mean_identity(g->s, g->dY, g->dAlpha, g->noise + 1e-8 + g->jitter, g->N, g->dMu);
        launch_argbest(g->s, g->dMu, g->N, -1, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
        double v = 0.0;
        HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
        GP_SYNC(g->s);
        g->fmin = v;
        g->fmin_valid = true;
    }
    *fmin = g->fmin;
    return 0;
}

int run_acq(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std) {
    if (g->P != 1) return fail(GP_ERR_ARG, "acquisitions need P == 1");
    if (type < GP_ACQ_EI || type > GP_ACQ_MPI) return fail(GP_ERR_ARG, "unknown acquisition %d", type);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if (!g->predicted || g->predicted_noise != 1)
        if ((rc = run_predict(g, 1))) return rc;  // GPModel.predict: with_noise=True (gpmodel.py:102)
    launch_acq(g->s, type, par, fmin, y_mean, y_std, g->dMean, g->dVar, g->M, g->dAcq);
    return 0;
}

extern "C" int gp_acq(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, double *out) {
    if (!g || !out) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

extern "C" int gp_acq_argbest(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int sense, int64_t *idx,
                   double *val) {
    if (!g || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
    double v = 0.0;
    long long i = 0;
    HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(&i, g->dRedI + 256, sizeof(long long), hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    *val = v;
    *idx = (int64_t)i;
    return 0;
}

// ---- local penalisation (batch acquisition of run.py:1238-1257; GPyOpt/GPyOpt/acquisitions/LP.py) -----------
// the batch centres, radii and scales of the penaliser (<= 256 rows) in a small device buffer of their own
int upload_lp_batch(gp_ctx *g, const double *Xb, int nb, const double *r0, const double *s0, LpBatch *b) {
    if (nb < 0 || nb > 256) return fail(GP_ERR_ARG, "batch size out of range (0..256)");
    g->lp_cache_nb = -1;   // (api_rows.hip keeps the last batch it uploaded; this upload replaces it)
    int rc;
    if ((rc = dev_realloc(&g->dLp, &g->capLp, (long)256 * (GP_MAX_D + 2)))) return rc;
    b->X = g->dLp;
    b->r = g->dLp + 256 * GP_MAX_D;
    b->s = b->r + 256;
    if (nb > 0) {
        HIPCHK(hipMemcpyAsync(b->X, Xb, sizeof(double) * nb * g->D, hipMemcpyHostToDevice, g->s));
        HIPCHK(hipMemcpyAsync(b->r, r0, sizeof(double) * nb, hipMemcpyHostToDevice, g->s));
        HIPCHK(hipMemcpyAsync(b->s, s0, sizeof(double) * nb, hipMemcpyHostToDevice, g->s));
    }
    return 0;
}

int run_acq_lp(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std, int transform, const double *Xb, int nb, const double *r0, const double *s0) {
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    LpBatch b;
    if ((rc = upload_lp_batch(g, Xb, nb, r0, s0, &b))) return rc;
    launch_lp(g->s, g->dAcq, g->dXs, g->M, g->D, b.X, nb, b.r, b.s, transform, g->dAcq);
    return 0;
}

extern "C" int gp_acq_lp(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
              const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out) {
    if (!g || !out || (nb > 0 && (!Xb || !r_x0 || !s_x0))) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_lp(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0))) return rc;
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

extern "C" int gp_acq_lp_argbest(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
                      const double *Xb, int nb, const double *r_x0, const double *s_x0, int sense,
                      const int64_t *exclude, int nex, int64_t *idx, double *val) {
    if (!g || !idx || !val || (nb > 0 && (!Xb || !r_x0 || !s_x0)) || (nex > 0 && !exclude))
        return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (nex < 0 || nex > 256) return fail(GP_ERR_ARG, "too many excluded rows (<= 256)");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_lp(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0))) return rc;
    if (nex > 0) {  // rows already taken never win (run.py:1249-1252 masks them)
        for (int i = 0; i < nex; ++i)
            if (exclude[i] < 0 || exclude[i] >= g->M) return fail(GP_ERR_ARG, "excluded row out of range");
        HIPCHK(hipMemcpyAsync(g->dRedI + 300, exclude, sizeof(long long) * nex, hipMemcpyHostToDevice, g->s));
        launch_mask(g->s, g->dAcq, g->dRedI + 300, nex, sense > 0 ? -INFINITY : INFINITY);
    }
    launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
    double v = 0.0;
    long long i = 0;
    HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(&i, g->dRedI + 256, sizeof(long long), hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    *val = v;
    *idx = (int64_t)i;
    return 0;
}

// full_cov = True branch of PosteriorExact._raw_predict (posterior.py:280-284)
extern "C" int gp_predict_full_cov(gp_t *g, int include_noise, double *mean, double *cov) {
    if (!g || !cov) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, Npad = g->Npad, Mpad = round_up(M, GP_TILE);
    if (Mpad > g->mc_max) return fail(GP_ERR_ARG, "full covariance needs M <= mc_max (%ld)", g->mc_max);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise, true))) return rc;  // leaves S = K(Xs,X) L^-T in dT2 (single chunk, zero padding rows: tile path)
    if ((rc = dev_realloc(&g->dCov, &g->capCov, std::max(g->capCov, Mpad * Mpad)))) return rc;
    const int mt = (int)(Mpad / GP_TILE);
    launch_kbuild(g->s, g->dCov, Mpad, g->dXs, M, Mpad, g->kp, 0.0, 1);  // K(Xs, Xs)
    gemm(g, g->s, 1, g->dCov, Mpad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, mt, 0, mt, 0});
    if (include_noise) launch_add_diag(g->s, g->dCov, Mpad, M, g->noise);  // gaussian.py:104-105
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * M * g->P, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    HIPCHK(hipMemcpy2D(cov, sizeof(double) * M, g->dCov, sizeof(double) * Mpad, sizeof(double) * M, M,
                       hipMemcpyDeviceToHost));
    return 0;
}

// ---- top-k of the acquisition scores (anchor_points_generator.py:61: argsort(scores)[:num_anchor]) ---------------
// k rounds of the deterministic arg-best reduction, each followed by masking the winner on the device: ties resolve
// to the lowest index in every round, i.e. the order of a stable sort by (score, index).
extern "C" int gp_acq_topk(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int sense, int k,
                int64_t *idx, double *val) {
    if (!g || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (k < 1 || k > GP_TOPK_MAX) return fail(GP_ERR_ARG, "k out of range (1..%d)", GP_TOPK_MAX);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    if ((rc = dev_realloc(&g->dComm, &g->capComm, 2L * GP_TOPK_MAX * (1 + 128)))) return rc;
    double *dv = g->dComm;
    long long *di = (long long *)(g->dComm + GP_TOPK_MAX);
    const int kk = (int)std::min<long>(k, g->M);
    for (int j = 0; j < kk; ++j) {
        launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
        HIPCHK(hipMemcpyAsync(dv + j, g->dRedV + 256, 8, hipMemcpyDeviceToDevice, g->s));
        HIPCHK(hipMemcpyAsync(di + j, g->dRedI + 256, 8, hipMemcpyDeviceToDevice, g->s));
        launch_mask(g->s, g->dAcq, g->dRedI + 256, 1, sense > 0 ? -INFINITY : INFINITY);
    }
    std::vector<long long> hi(kk);
    HIPCHK(hipMemcpyAsync(val, dv, sizeof(double) * kk, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(hi.data(), di, sizeof(long long) * kk, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    for (int j = 0; j < kk; ++j) idx[j] = (int64_t)hi[j];
    for (int j = kk; j < k; ++j) {  // fewer candidates than k: the tail is marked empty
        idx[j] = -1;
        val[j] = sense > 0 ? -INFINITY : INFINITY;
    }
    return 0;
}

// ---- posterior samples of the latent function (GP.posterior_samples_f, gp.py:581-609) ------------------------
// dev[s, :] = C z_s with C C^T = cov(Xs) (+ noise I) the full posterior covariance (posterior.py:280-284) of the resident
// candidates and z_s the caller's standard normals: the M x M Cholesky runs on the device with the same tile kernels as
// the fit, under GPy's jitter ladder (jitchol, linalg.py:56-81).  mean[M,P] is returned beside the deviations; a sample
// of output d is mean[:, d] + dev[s, :].  (The reference draws through numpy's multivariate_normal, whose SVD factor
// differs from C by an orthogonal matrix: same distribution, different draws for the same generator state.)
extern "C" int gp_posterior_samples(gp_t *g, int include_noise, const double *Z, int S, int maxtries, double *mean, double *dev,
                         double *jitter_used) {
    if (!g || !Z || !dev) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (S < 1) return fail(GP_ERR_ARG, "S < 1");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, Npad = g->Npad, Mpad = round_up(M, GP_TILE), Spad = round_up(S, GP_TILE);
    if (Mpad > g->mc_max) return fail(GP_ERR_ARG, "posterior samples need M <= mc_max (%ld)", g->mc_max);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise, true))) return rc;  // S_c = K(Xs,X) L^-T in dT2 (single chunk, zero padding rows: tile path), mean in dMean
    // dCov: [cov Mpad x Mpad][Z^T Spad x Mpad][dev Spad x Mpad]; the inverted diagonal tiles go to dT (free now)
    if ((rc = dev_realloc(&g->dCov, &g->capCov, std::max(g->capCov, Mpad * Mpad + 2 * Spad * Mpad)))) return rc;
    double *C = g->dCov, *Zd = g->dCov + Mpad * Mpad, *Dv = Zd + Spad * Mpad;
    double *invL = g->dT;
    const int mt = (int)(Mpad / GP_TILE), st = (int)(Spad / GP_TILE);
    HIPCHK(hipMemsetAsync(Zd, 0, sizeof(double) * Spad * Mpad, g->s));
    // (the diagonal-tile kernel writes the lower block triangle of an inverted tile only; dT held candidate rows before)
    HIPCHK(hipMemsetAsync(invL, 0, sizeof(double) * Mpad * GP_TILE, g->s));
    HIPCHK(hipMemcpy2DAsync(Zd, sizeof(double) * Mpad, Z, sizeof(double) * M, sizeof(double) * M, S,
                            hipMemcpyHostToDevice, g->s));
    // jitchol scales its ladder by the mean of the diagonal of the matrix it factors (linalg.py:62-66: diagA.mean() * 1e-6):
    // here the POSTERIOR covariance, whose diagonal near training points is orders of magnitude below the prior variance
    double diag_stat[2] = {0.0, 0.0};   // trace and smallest entry of the diagonal of the matrix to factor
    double diag_mean = 0.0;
    double jitter = 0.0;
    int tries = 0, info = 0;
    for (;;) {
        launch_kbuild(g->s, C, Mpad, g->dXs, M, Mpad, g->kp, 0.0, 1);  // K(Xs, Xs), identity on the padding rows
        gemm(g, g->s, 1, C, Mpad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, mt, 0, mt, 0});
        if (include_noise) launch_add_diag(g->s, C, Mpad, M, g->noise);
        if (tries == 0) {
            launch_trace(g->s, C, Mpad, M, g->dScal + 420);
            HIPCHK(hipMemcpyAsync(diag_stat, g->dScal + 420, 2 * sizeof(double), hipMemcpyDeviceToHost, g->s));
        }
        if (jitter != 0.0) launch_add_diag(g->s, C, Mpad, M, jitter);
        HIPCHK(hipMemsetAsync(g->dInfo, 0, sizeof(int) * 4, g->s));
        factor_buf(g, C, Mpad, mt, mt, invL, g->dInfo);
        HIPCHK(hipMemcpyAsync(&info, g->dInfo, sizeof(int), hipMemcpyDeviceToHost, g->s));
        GP_SYNC(g->s);
        if (g->emulate_fp64 && info == 0) {
            int bad = 0;
            HIPCHK(hipMemcpy(&bad, g->dInfo + 2, sizeof(int), hipMemcpyDeviceToHost));
            if (bad) return fail(GP_ERR_STATE, "emulate_fp64: an entry of L left the fixed-point range");
        }
        if (info == 0) break;
        // jitchol: mean(diag) * 1e-6 * 10^k (linalg.py:62-75)
        if (tries == 0) diag_mean = diag_stat[0] / (double)M;
        // np.any(diagA <= 0.) raises before any jitter is tried (linalg.py:61-62): ANY entry, not the mean -- a posterior
        // variance that came out slightly negative at a training point is such an entry
        if (diag_stat[1] <= 0.0 || !(diag_mean > 0.0)) return fail(GP_ERR_NOT_PD_DIAG, "not pd: non-positive diagonal elements");
        jitter = tries == 0 ? diag_mean * 1e-6 : jitter * 10.0;
        if (++tries > maxtries || !std::isfinite(jitter)) {
            g_err = "not positive definite, even with jitter.";
            return info > 0 ? info : 1;
        }
    }
    launch_zero_upper_diag(g->s, C, Mpad, mt);
    // dev[s, m] = sum_{k <= m} z[s, k] C[m, k]: B = the factor's rows, contraction ends at the diagonal tile
    GemmOpt o;
    o.k_end_tri = 1;
    gemm(g, g->s, 0, Dv, Mpad, Zd, Mpad, C, Mpad, 1, (int)Mpad, TileSet{0, st, 0, mt, 0}, o);
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * M * g->P, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpy2DAsync(dev, sizeof(double) * M, Dv, sizeof(double) * Mpad, sizeof(double) * M, S,
                            hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    if (jitter_used) *jitter_used = jitter;
    g->predicted = false;  // dT was used as workspace
    return 0;
}

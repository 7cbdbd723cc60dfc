// Hyper-parameter gradients of the LML, predictive gradients, acquisition gradients, dL_dK.
// Reference: Stationary.update_gradients_full (stationary.py:218-238), GP.predictive_gradients (gp.py:407-454).
#include "api_internal.h"

int lml_grad_impl(gp_ctx *g, double *dvariance, double *dlengthscale, double *dnoise, bool reset_phases) {
    if (!g || !dvariance || !dlengthscale || !dnoise) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->P > 16) return fail(GP_ERR_ARG, "gp_lml_grad supports P <= 16");
    // (a Gower model gets the fork's values: K through the Gower branch in the variance gradient, Euclidean dK/dr on the kernel's own
    // lengthscale in the lengthscale gradients, stationary.py:218-238 -- not derivatives of its LML, which the host layer knows)
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if (reset_phases) g->nphases = 0;
    if ((rc = ensure_wi(g))) return rc;
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE), D = g->D;
    const long ntile = (long)nt * (nt + 1) / 2;
    // per-tile partials live in dT (free after ensure_wi): ntile * NACC doubles << Npad^2
    double *partial = g->dT;
    int ph = phase_begin(g, "lml_grad", 0.0, 8.0 * (double)g->N * g->N / 2);
    std::vector<double> host((size_t)GP_GRAD_NACC * ((D + GP_GRAD_CH - 1) / GP_GRAD_CH));
    int pass = 0;
    for (int d0 = 0; d0 < D; d0 += GP_GRAD_CH, ++pass) {
        launch_lml_grad(g->s, g->dX, g->N, Npad, g->kp, g->ard, d0, g->dAlpha, g->P, g->dWi, Npad, partial,
                        g->dScal + 64 + pass * GP_GRAD_NACC);
        if (!g->ard) break;
    }
    phase_end(g, ph);
    const int npass = g->ard ? (D + GP_GRAD_CH - 1) / GP_GRAD_CH : 1;
    HIPCHK(hipMemcpyAsync(host.data(), g->dScal + 64, sizeof(double) * GP_GRAD_NACC * npass, hipMemcpyDeviceToHost,
                          g->s));
    GP_SYNC(g->s);
    (void)ntile;
    *dvariance = host[0] / g->kp.variance;  // stationary.py:224
    *dnoise = host[1];                      // gaussian.py:78-79
    if (g->ard) {
        for (int d = 0; d < D; ++d)         // -sum tmp (dx_q)^2 / l_q^3, stationary.py:230-235,260-261
            dlengthscale[d] = -host[(d / GP_GRAD_CH) * GP_GRAD_NACC + 2 + (d % GP_GRAD_CH)] / g->kp.ls[d];
    } else {
        dlengthscale[0] = -host[2] / g->kp.ls[0];  // -sum(dL_dr * r) / l, stationary.py:237-238
    }
    return 0;
}

extern "C" int gp_lml_grad(gp_t *g, double *dvariance, double *dlengthscale, double *dnoise) {
    return lml_grad_impl(g, dvariance, dlengthscale, dnoise, true);
}

// gp_fit + gp_lml_grad as ONE call (what every L-BFGS evaluation of the hyper-parameter loop asks for:
// Model.objective_function + objective_function_gradients, core/model.py:96-127).  The first stages of the solve for
// L^-T ride behind the factorisation's latency-bound tail, like the candidate stages of gp_fit_predict.
extern "C" int gp_fit_grad(gp_t *g, int maxtries, double *lml, double *logdet, double *jitter_used, double *dvariance,
                double *dlengthscale, double *dnoise) {
    if (!g || !dvariance || !dlengthscale || !dnoise) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit_grad");
    if (g->P > 16) return fail(GP_ERR_ARG, "gp_lml_grad supports P <= 16");
    HIPCHK(hipSetDevice(g->device));
    const int nt = (int)(g->Npad / GP_TILE);
    // emulated: Ky^-1 in residue form after the factorisation (wi_rns) instead of fp64 stages pipelined behind it
    const bool emu_wi = g->emulate_fp64 && g->emulate_fit && g->panel_tiles % 2 == 0;
    const bool can_pipe = g->lookahead && nt > g->panel_tiles && !emu_wi;
    int rc;
    if ((rc = fit_impl(g, maxtries, can_pipe ? 2 : 0, 0))) return rc;
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    return lml_grad_impl(g, dvariance, dlengthscale, dnoise, false);
}

// ---- second candidate-sized buffer (beta = K(Xs,X) Ky^-1, or the full covariance) -----------------
int ensure_grad_buffers(gp_ctx *g, long elemsBeta, long M) {
    int rc;
    if ((rc = dev_realloc(&g->dCov, &g->capCov, elemsBeta))) return rc;
    const long need = M * (long)g->D * std::max(1, g->P);
    if (!g->dDm || g->capD < need) {
        for (double **b : {&g->dDm, &g->dDv, &g->dDacq}) {
            if (*b) hipFree(*b);
            *b = nullptr;
        }
        HIPCHK(hipMalloc((void **)&g->dDm, sizeof(double) * need));
        HIPCHK(hipMalloc((void **)&g->dDv, sizeof(double) * need));
        HIPCHK(hipMalloc((void **)&g->dDacq, sizeof(double) * need));
        g->capD = need;
    }
    return 0;
}

// predictive gradients of all resident candidates into dDm [M, D, P] and dDv [M, D]
// A Gower model (gp_set_gower) gets what the fork computes for it (gp.py:407-454 over stationary.py:336-364): K(Xs, X) -- and
// with it beta -- takes the Gower branch (stationary.py:116-135), while gradients_X stays the Euclidean formula on the
// kernel's own lengthscale parameter (_inv_dist :251-258, dK_dr_via_X :142-148).  kp.ls holds that parameter, kp.gdiv the
// ranges: cross_k reads the latter, predict_grad_kernel the former.  Inconsistent as a derivative, but it is the function
// the reference's L-BFGS (run.py:1206-1225) and estimate_L (run.py:1244) see.
int run_predict_grad(gp_ctx *g) {
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    int rc;
    if (g->M <= g->small_m && !g->wi_valid) {
        // A handful of locations right after a fit: beta = Ky^-1 k* by TWO substitutions against L (dpotrs, the reference's own route
        // for alpha, exact_gaussian_inference.py:60) instead of building Ky^-1 first -- the potri-equivalent costs 2 N^3 / 3 (50 ms at
        // N = 16384), ~90 short launches cost 1.4 ms.  Posterior mean / variance of the same rows fall out on the way.
        const long M = g->M, N = g->N, Npad = g->Npad;
        if ((rc = ensure_panel_inv(g))) return rc;
        if ((rc = ensure_out(g))) return rc;
        if ((rc = dev_realloc(&g->dT, &g->capT, std::max(g->capT, (long)GP_TILE * Npad)))) return rc;
        if ((rc = dev_realloc(&g->dT2, &g->capT2, std::max(g->capT2, (long)GP_TILE * Npad)))) return rc;
        if ((rc = ensure_grad_buffers(g, std::max(g->capCov, (long)GP_TILE * Npad), M))) return rc;
        g->w_in_t2 = false;
        launch_cross_k_rows(g->s, g->dT, Npad, g->dXs, (int)M, g->dX, N, Npad, g->kp);
        launch_small_forward_solve(g->s, g->dA, Npad, g->dInvP, g->invp_W, Npad, g->dT, g->dT2, Npad, (int)M);   // dT2 = w rows
        launch_predict_reduce(g->s, g->dT2, Npad, M, N, g->dA + Npad * Npad, Npad, g->P, g->kp.variance, g->noise, g->dMean,
                              g->dVar);
        g->predicted = true;       // GPModel.predict: with_noise=True (gpmodel.py:102)
        g->predicted_noise = 1;
        launch_trsv_backward(g->s, g->dA, Npad, g->dInvP, g->invp_W, Npad, g->dT2, Npad, (int)M, g->dCov, g->dT);   // beta = L^-T w
        launch_predict_grad(g->s, g->dXs, M, g->dX, N, g->kp, g->dAlpha, Npad, g->P, g->dCov, Npad, g->dDm, g->dDv);
        return 0;
    }
    if ((rc = ensure_wi(g))) return rc;
    const long M = g->M, N = g->N, Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const long mc_max = std::min(g->mc_max, round_up(M, GP_TILE));
    if ((rc = dev_realloc(&g->dT, &g->capT, std::max(g->capT, mc_max * Npad)))) return rc;
    if ((rc = ensure_grad_buffers(g, mc_max * Npad, M))) return rc;
    for (long m0 = 0; m0 < M; m0 += mc_max) {
        const long mc = std::min(mc_max, M - m0);
        const long mcpad = round_up(mc, GP_TILE);
        const int mt = (int)(mcpad / GP_TILE);
        if (M <= g->small_m)
            launch_cross_k_rows(g->s, g->dT, Npad, g->dXs, (int)M, g->dX, N, Npad, g->kp);
        else
            launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, N, Npad, g->kp);
        // beta = K(Xs, X) Ky^-1   (gp.py:451-452; Ky^-1 symmetric => rows of Wi serve as the B operand)
        if (M <= g->small_m)   // a handful of rows: row dots with Ky^-1, one read of it (smallm.hip)
            launch_small_wi_product(g->s, g->dWi, Npad, Npad, g->dT, Npad, (int)mc, g->dCov, Npad);
        else
            gemm(g, g->s, 0, g->dCov, Npad, g->dT, Npad, g->dWi, Npad, 1, (int)Npad, TileSet{0, mt, 0, nt, 0});
        launch_predict_grad(g->s, g->dXs + m0 * g->D, mc, g->dX, N, g->kp, g->dAlpha, Npad, g->P, g->dCov, Npad,
                            g->dDm + m0 * g->D * g->P, g->dDv + m0 * g->D);
    }
    g->predicted = false;  // dT no longer holds the solved candidates
    return 0;
}

extern "C" int gp_predict_grad(gp_t *g, double *dmdx, double *dvdx) {
    if (!g || !dmdx) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if (!dvdx) {
        // the mean's gradients alone (what estimate_L maximises, batch_local_penalization.py:55-58): gradients_X(alpha^T, X*, X) needs
        // neither Ky^-1 nor K(X*, X) Ky^-1 -- one pass of O(M N D) instead of 2 N^3 / 3 + N^2 M flops
        if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
        if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
        if ((rc = ensure_grad_buffers(g, std::max<long>(g->capCov, 1), g->M))) return rc;
        launch_predict_grad(g->s, g->dXs, g->M, g->dX, g->N, g->kp, g->dAlpha, g->Npad, g->P, nullptr, 0, g->dDm, g->dDv);
        HIPCHK(hipMemcpyAsync(dmdx, g->dDm, sizeof(double) * g->M * g->D * g->P, hipMemcpyDeviceToHost, g->s));
        GP_SYNC(g->s);
        return 0;
    }
    if ((rc = run_predict_grad(g))) return rc;
    HIPCHK(hipMemcpyAsync(dmdx, g->dDm, sizeof(double) * g->M * g->D * g->P, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(dvdx, g->dDv, sizeof(double) * g->M * g->D, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

static int run_acq_grad(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std);

extern "C" int gp_acq_grad(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, double *out, double *dout) {
    if (!g || !out || !dout) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_grad(g, type, par, fmin, y_mean, y_std))) return rc;
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(dout, g->dDacq, sizeof(double) * g->M * g->D, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

// the base acquisition's negated value and gradient of every resident candidate into dAcq [M] and dDacq [M, D]
static int run_acq_grad(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std) {
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (g->P != 1) return fail(GP_ERR_ARG, "acquisitions need P == 1");
    if (type < GP_ACQ_EI || type > GP_ACQ_MPI) return fail(GP_ERR_ARG, "unknown acquisition %d", type);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict_grad(g))) return rc;
    if (!g->predicted || g->predicted_noise != 1)      // (the substitution route of a handful of rows leaves mean / variance behind)
        if ((rc = run_predict(g, 1))) return rc;
    launch_acq_grad(g->s, type, par, fmin, y_mean, y_std, g->dMean, g->dVar, g->dDm, g->dDv, g->M, g->D, g->dAcq,
                    g->dDacq);
    return 0;
}

// AcquisitionLP.acquisition_function_withGradients (GPyOpt/GPyOpt/acquisitions/LP.py:112-140): the base acquisition's value
// and gradient, then the log transform and the penaliser as an epilogue over the same buffers.
extern "C" int gp_acq_lp_grad(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
                              const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out, double *dout) {
    if (!g || !out || !dout || (nb > 0 && (!Xb || !r_x0 || !s_x0))) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (transform != 0 && transform != 1) return fail(GP_ERR_ARG, "transform must be 0 (none) or 1 (softplus)");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_grad(g, type, par, fmin, y_mean, y_std))) return rc;
    LpBatch b;
    if ((rc = upload_lp_batch(g, Xb, nb, r_x0, s_x0, &b))) return rc;
    launch_lp_grad(g->s, g->dAcq, g->dDacq, g->dXs, g->M, g->D, b.X, nb, b.r, b.s, transform);
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(dout, g->dDacq, sizeof(double) * g->M * g->D, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

// ---- dL_dK = 0.5 (alpha alpha^T - P Ky^-1)  (exact_gaussian_inference.py:70) -------------------------------
// What grad_dict['dL_dK'] carries into kern.update_gradients_full (gp.py:269) when the reference's own kernel classes
// consume it on the host.  gp_lml_grad forms the same matrix implicitly inside its fused reduction.
extern "C" int gp_get_dl_dk(gp_t *g, double *dL_dK) {
    if (!g || !dL_dK) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_wi(g))) return rc;  // leaves dT free (Npad x Npad)
    const long N = g->N, Npad = g->Npad;
    launch_dldk(g->s, g->dT, Npad, g->dAlpha, Npad, g->P, g->dWi, Npad, N);
    GP_SYNC(g->s);
    HIPCHK(hipMemcpy2D(dL_dK, sizeof(double) * N, g->dT, sizeof(double) * Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    g->predicted = false;  // dT was reused
    return 0;
}

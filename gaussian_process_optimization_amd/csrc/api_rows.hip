// A handful of locations per call, answered in one entry point: what scipy's L-BFGS-B asks of the acquisition between two fits
// (GPyOpt/GPyOpt/optimization/optimizer.py:36-61 -> acquisitions/base.py:33-50, LP.py:105-140 -> models/gpmodel.py:95-142 ->
// GPy/GPy/core/gp.py:297-354,407-454).  The fused path (onerow.hip) takes the locations BY VALUE and returns through a pinned
// result block; everything it cannot take goes through gp_set_candidates + the batched entry points, same results.
#include "api_internal.h"

// fused path: up to 8 locations of a single-output model whose coordinates fit the kernel arguments
static bool rows_fused_ok(const gp_ctx *g, int64_t M) {
    return g->small_m > 0 && M >= 1 && M <= g->small_m && g->P == 1 && (long)std::min<int64_t>(M, ROWS_MAX_M) * g->D <= ROWS_MAX_XS;
}

// When to build the inverse factor (N^3 / 3 flops once per fit: 0.7 ms at N = 4096, 4.4 ms at N = 8192, 28 ms at N = 16384).  A gradient
// call through the substitution route (~90 short launches against L, no precomputation) costs 1.84 ms at N = 16384 against 0.44 fused
// (0.55 against 0.16 at N = 8192), so the factor pays for itself after build / (substitution - fused) = ~20 calls at N = 16384, ~11 at
// N = 8192: about nt / 6 at either size.  Not knowing how many calls will follow a fit, the rule is the ski-rental one -- rent until
// the rent paid equals the price: the first nt / 6 calls after a fit take the substitution route, the next one builds the factor
// (never worse than twice the best choice in hindsight).  Small matrices (N <= 4096) build at the first call; a factor that exists is
// always used.
static bool rows_use_factor(gp_ctx *g) {
    if (g->li_valid || g->N <= 4096 || g->rows_build == 1) return true;
    if (g->rows_build == 0) return false;
    return ++g->rows_calls_since_fit > g->Npad / GP_TILE / 6;
}

static int rows_scratch(gp_ctx *g, RowsWork *w) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const long nch = nt > 0 ? (nt - 1) / 8 + 1 : 1;
    const long nrb = (long)nt * (GP_TILE / rows_block_height(nt));   // row blocks of the backward pass
    const long n_w = nch * ROWS_MAX_M * Npad, n_b = nrb * ROWS_MAX_M * Npad, n_m = nch * ROWS_MAX_M, n_v = nrb * ROWS_MAX_M;
    const long n_g = (long)rows_gpart_elems(g->N);
    int rc;
    if ((rc = dev_realloc(&g->dRows, &g->capRows, n_w + n_b + n_m + n_v + n_g))) return rc;
    w->wpart = g->dRows;
    w->bpart = w->wpart + n_w;
    w->meanpart = w->bpart + n_b;
    w->vpart = w->meanpart + n_m;
    w->gpart = w->vpart + n_v;
    if (!g->dRowsCounter) {
        HIPCHK(hipMalloc((void **)&g->dRowsCounter, sizeof(unsigned int)));
        HIPCHK(hipMemsetAsync(g->dRowsCounter, 0, sizeof(unsigned int), g->s));
    }
    w->counter = g->dRowsCounter;
    if (!g->hRowsOut) {   // 3 MV (1 + D) doubles + the ticket; coherent host memory the finish kernel writes and the host reads after the sync
        HIPCHK(hipHostMalloc((void **)&g->hRowsOut, sizeof(double) * (ROWS_OUT_DOUBLES + 1), hipHostMallocDefault));
        g->hRowsOut[ROWS_OUT_DOUBLES] = 0.0;
    }
    w->ticket = (g->rows_ticket += 1.0);
    w->counter_base = g->rows_counter_base;
    return 0;
}

// One pass is on the stream: account for its arrivals, wait, and make sure it finished.  The finishing workgroup writes the pass's
// ticket behind the results; if the stream reports an error or the ticket is not this pass's, a launch did not complete and the
// arrival counter may hold anything: counter and base are cleared so that the next call starts clean, and this one fails loudly
// instead of handing back the previous call's numbers.
static int rows_wait(gp_ctx *g, RowsWork &w, unsigned finish_grid) {
    g->rows_counter_base += finish_grid;
    hipError_t e = hipStreamSynchronize(g->s);
    int pending = gp_pending_error();
    if (!pending && e == hipSuccess && g->hRowsOut[ROWS_OUT_DOUBLES] == w.ticket) return 0;
    const std::string noted = pending ? gp_last_error() : std::string();
    hipStreamSynchronize(g->s);
    hipMemset(g->dRowsCounter, 0, sizeof(unsigned int));
    g->rows_counter_base = 0;
    if (pending) return fail(GP_ERR_HIP, "%s", noted.c_str());
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hipStreamSynchronize -> %s (one-location pass)", hipGetErrorString(e));
    return fail(GP_ERR_HIP, "the one-location kernels did not complete (ticket %.0f, expected %.0f)", g->hRowsOut[ROWS_OUT_DOUBLES],
                w.ticket);
}

// the penaliser's batch on the device; re-uploaded only when it changed (an L-BFGS run keeps one batch for hundreds of calls)
static int rows_lp_batch(gp_ctx *g, const double *Xb, int nb, const double *r0, const double *s0, LpBatch *b) {
    const size_t nx = (size_t)nb * g->D;
    bool same = g->lp_cache_nb == nb && g->lp_cache.size() == nx + 2 * (size_t)nb && g->dLp;
    if (same && nb > 0)
        same = !memcmp(g->lp_cache.data(), Xb, sizeof(double) * nx) && !memcmp(g->lp_cache.data() + nx, r0, sizeof(double) * nb) &&
               !memcmp(g->lp_cache.data() + nx + nb, s0, sizeof(double) * nb);
    if (same) {
        b->X = g->dLp;
        b->r = g->dLp + 256 * GP_MAX_D;
        b->s = b->r + 256;
        return 0;
    }
    int rc;
    if ((rc = upload_lp_batch(g, Xb, nb, r0, s0, b))) return rc;
    g->lp_cache.resize(nx + 2 * (size_t)nb);
    if (nb > 0) {
        memcpy(g->lp_cache.data(), Xb, sizeof(double) * nx);
        memcpy(g->lp_cache.data() + nx, r0, sizeof(double) * nb);
        memcpy(g->lp_cache.data() + nx + nb, s0, sizeof(double) * nb);
    }
    g->lp_cache_nb = nb;
    return 0;
}

// One pass of the fused path per ROWS_MAX_M locations.  want_grad selects forward + backward + finish; otherwise forward + finish.
// Results: mean / var / acq [M], dmdx / dvdx / dacq [M, D] (any may be null).
static int rows_fused(gp_ctx *g, const double *Xs, int M, int include_noise, int want_grad, const RowsAcq &aq, double *mean,
                      double *var, double *acq, double *dmdx, double *dvdx, double *dacq) {
    int rc;
    if ((rc = ensure_linv(g))) return rc;
    RowsWork w;
    if ((rc = rows_scratch(g, &w))) return rc;
    const int D = g->D;
    const bool timed = g->profiling;
    if (timed) g->nphases = 0;
    for (int m0 = 0; m0 < M; m0 += ROWS_MAX_M) {
        const int mc = std::min(ROWS_MAX_M, M - m0);
        const int MV = mc == 1 ? 1 : ROWS_MAX_M;
        RowsX rx;
        rx.M = mc;
        memcpy(rx.xs, Xs + (long)m0 * D, sizeof(double) * mc * D);
        int ph = timed ? phase_begin(g, want_grad ? "rows_fused_grad" : "rows_fused", (want_grad ? 2.0 : 1.0) * (double)g->N * g->N * mc,
                                     (want_grad ? 2.0 : 1.0) * 8.0 * (double)g->N * g->N / 2)
                       : -1;
        if (m0 > 0) {
            w.ticket = (g->rows_ticket += 1.0);
            w.counter_base = g->rows_counter_base;
        }
        launch_rows(g->s, g->dLi, g->Npad, rx, g->kp, g->dX, g->N, g->dAlpha, want_grad, g->kp.variance,
                    include_noise ? g->noise : 0.0, aq, w, g->hRowsOut, g->rows_nt < 0 ? (g->Npad > 8192 ? 1 : 0) : g->rows_nt);
        if (timed) phase_end(g, ph);
        if ((rc = rows_wait(g, w, rows_finish_grid(g->N)))) return rc;
        const double *o = g->hRowsOut;
        for (int m = 0; m < mc; ++m) {
            if (mean) mean[m0 + m] = o[m];
            if (var) var[m0 + m] = o[MV + m];
            if (acq) acq[m0 + m] = o[2 * MV + m];
            const double *gm = o + 3 * MV + (long)m * D, *gv = gm + (long)MV * D, *ga = gv + (long)MV * D;
            if (dmdx) memcpy(dmdx + (long)(m0 + m) * D, gm, sizeof(double) * D);
            if (dvdx) memcpy(dvdx + (long)(m0 + m) * D, gv, sizeof(double) * D);
            if (dacq) memcpy(dacq + (long)(m0 + m) * D, ga, sizeof(double) * D);
        }
    }
    ++g->rows_fused_calls;
    return 0;
}

// The posterior of a handful of locations: gp_set_candidates + gp_predict (+ gp_predict_grad when dmdx / dvdx are given) as ONE
// call (PosteriorExact._raw_predict, posterior.py:273-302; GP.predictive_gradients, gp.py:407-454).
extern "C" int gp_predict_rows(gp_t *g, const double *Xs, int64_t M, int include_noise, double *mean, double *var, double *dmdx,
                               double *dvdx) {
    if (!g || !Xs) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (M < 1) return fail(GP_ERR_ARG, "M < 1");
    if (dvdx && !dmdx) return fail(GP_ERR_ARG, "dvdx needs dmdx");
    if (dmdx && !dvdx && (mean || var)) return fail(GP_ERR_ARG, "the mean's gradient alone (dvdx NULL) comes without mean / var");
    HIPCHK(hipSetDevice(g->device));
    if (dmdx && !dvdx) {
        // d mean / dx alone: one pass over the training points, no inverse factor, no substitutions (estimate_L's inner call)
        if (rows_fused_ok(g, M)) {
            int rc;
            RowsWork w;
            if ((rc = rows_scratch(g, &w))) return rc;
            for (int m0 = 0; m0 < (int)M; m0 += ROWS_MAX_M) {
                const int mc = std::min(ROWS_MAX_M, (int)M - m0);
                const int MV = mc == 1 ? 1 : ROWS_MAX_M;
                RowsX rx;
                rx.M = mc;
                memcpy(rx.xs, Xs + (long)m0 * g->D, sizeof(double) * mc * g->D);
                if (m0 > 0) {
                    w.ticket = (g->rows_ticket += 1.0);
                    w.counter_base = g->rows_counter_base;
                }
                launch_rows_mean_grad(g->s, rx, g->kp, g->dX, g->N, g->dAlpha, w, g->hRowsOut);
                if ((rc = rows_wait(g, w, rows_mean_grad_grid(g->N)))) return rc;
                memcpy(dmdx + (long)m0 * g->D, g->hRowsOut + 3 * MV, sizeof(double) * mc * g->D);
            }
            ++g->rows_fused_calls;
            return 0;
        }
        ++g->rows_fallback_calls;
        int rc;
        if ((rc = gp_set_candidates(g, Xs, M))) return rc;
        return gp_predict_grad(g, dmdx, nullptr);
    }
    const int want_grad = dmdx != nullptr;
    if (rows_fused_ok(g, M) && rows_use_factor(g)) {
        RowsAcq aq{};
        return rows_fused(g, Xs, (int)M, include_noise, want_grad, aq, mean, var, nullptr, dmdx, dvdx, nullptr);
    }
    ++g->rows_fallback_calls;
    int rc;
    if ((rc = gp_set_candidates(g, Xs, M))) return rc;
    if (mean || var)
        if ((rc = gp_predict(g, include_noise, mean, var))) return rc;
    if (want_grad)
        if ((rc = gp_predict_grad(g, dmdx, dvdx))) return rc;
    return 0;
}

// The (negated) acquisition at a handful of locations and, when dout is given, its x-gradient: gp_set_candidates + gp_acq /
// gp_acq_grad (lp = 0; acquisitions/base.py:33-50) or gp_acq_lp / gp_acq_lp_grad (lp = 1; LP.py:105-140) as ONE call.
extern "C" int gp_acq_rows(gp_t *g, const double *Xs, int64_t M, int type, double par, double fmin, double y_mean, double y_std,
                           int lp, int transform, const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out,
                           double *dout) {
    if (!g || !Xs || !out) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (M < 1) return fail(GP_ERR_ARG, "M < 1");
    if (g->P != 1) return fail(GP_ERR_ARG, "acquisitions need P == 1");
    if (type < GP_ACQ_EI || type > GP_ACQ_MPI) return fail(GP_ERR_ARG, "unknown acquisition %d", type);
    if (lp && transform != 0 && transform != 1) return fail(GP_ERR_ARG, "transform must be 0 (none) or 1 (softplus)");
    if (lp && nb > 0 && (!Xb || !r_x0 || !s_x0)) return fail(GP_ERR_ARG, "null argument");
    if (lp && (nb < 0 || nb > 256)) return fail(GP_ERR_ARG, "batch size out of range (0..256)");
    HIPCHK(hipSetDevice(g->device));
    const int want_grad = dout != nullptr;
    int rc;
    if (rows_fused_ok(g, M) && rows_use_factor(g)) {
        RowsAcq aq{};
        aq.on = 1;
        aq.type = type;
        aq.par = par;
        aq.fmin = fmin;
        aq.y_mean = y_mean;
        aq.y_std = y_std;
        aq.lp = lp ? 1 : 0;
        aq.transform = transform;
        aq.nb = lp ? nb : 0;
        if (lp) {
            LpBatch b;
            if ((rc = rows_lp_batch(g, Xb, nb, r_x0, s_x0, &b))) return rc;
            aq.Xb = b.X;
            aq.r0 = b.r;
            aq.s0 = b.s;
        }
        return rows_fused(g, Xs, (int)M, 1, want_grad, aq, nullptr, nullptr, out, nullptr, nullptr, dout);   // with_noise=True, gpmodel.py:102
    }
    ++g->rows_fallback_calls;
    if ((rc = gp_set_candidates(g, Xs, M))) return rc;
    if (lp)
        return want_grad ? gp_acq_lp_grad(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0, out, dout)
                         : gp_acq_lp(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0, out);
    return want_grad ? gp_acq_grad(g, type, par, fmin, y_mean, y_std, out, dout) : gp_acq(g, type, par, fmin, y_mean, y_std, out);
}

// how many *_rows calls took the fused path / the batched entry points since the context was created (route checks in tests)
extern "C" int gp_rows_stats(gp_t *g, int64_t *fused, int64_t *fallback) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (fused) *fused = g->rows_fused_calls;
    if (fallback) *fallback = g->rows_fallback_calls;
    return 0;
}

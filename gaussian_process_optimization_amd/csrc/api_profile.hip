// Measurement: per-phase HIP-event timings and the per-launch accounting of the dominant kernels (bench.py).
#include "api_internal.h"

// ---- measurement ----------------------------------------------------------------------------------
extern "C" int gp_last_phases(gp_t *g, int cap, const char **names, double *ms, double *flops, double *bytes) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s);
    int n = std::min(cap, g->nphases);
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        hipEventElapsedTime(&t, g->phases[i].e0, g->phases[i].e1);
        if (names) names[i] = g->phases[i].name;
        if (ms) ms[i] = t;
        if (flops) flops[i] = g->phases[i].flops;
        if (bytes) bytes[i] = g->phases[i].bytes;
    }
    return n;
}

extern "C" int gp_profile(gp_t *g, int on) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    g->profiling = on != 0;
    g->profile_class = on == 2 ? 1 : 0;
    g->gemm_ev_used = 0;
    g->gemm_tiles.clear();
    g->gemm_K.clear();
    g->gemm_launches = 0;
    g->gemm_flops = 0.0;
    g->gemm_flops_all = 0.0;
    g->rns_ev_used = 0;
    g->rns_ops = 0.0;
    return 0;
}

extern "C" int gp_gemm_stats(gp_t *g, int64_t *launches, double *ms, double *flops) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g->gemm_ev_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g->gemm_events[i], g->gemm_events[i + 1]) == hipSuccess) tot += t;
    }
    if (launches) *launches = g->gemm_launches;
    if (ms) *ms = tot;
    if (flops) *flops = g->gemm_flops;
    return 0;
}

// The same for the residue GEMM (rns_gemm256_kernel; option "emulate_fp64"): launches, summed durations and int8
// operations (2 per multiply-add) since gp_profile(1).
extern "C" int gp_rns_stats(gp_t *g, int64_t *launches, double *ms, double *ops) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    hipSetDevice(g->device);
    for (hipStream_t st : {g->s_panel, g->s_bulk, g->s_inv, g->s_pred, g->s})
        if (st) hipStreamSynchronize(st);
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g->rns_ev_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g->rns_events[i], g->rns_events[i + 1]) == hipSuccess) tot += t;
    }
    if (launches) *launches = (int64_t)(g->rns_ev_used / 2);
    if (ms) *ms = tot;
    if (ops) *ops = g->rns_ops;
    return 0;
}

// Wall time during which at least one of the profiled launches was running (the union of their [start, end] intervals,
// measured against the first profiled launch's start event).  With overlapping launches (gp_fit_predict) the SUM of the
// durations counts shared time twice; flops / busy is the kernel's throughput while it runs.
extern "C" int gp_gemm_busy(gp_t *g, double *busy_ms) {
    if (!g || !busy_ms) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    std::vector<std::pair<double, double>> iv;
    for (size_t i = 0; i + 1 < g->gemm_ev_used; i += 2) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, g->gemm_events[0], g->gemm_events[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&b, g->gemm_events[0], g->gemm_events[i + 1]) != hipSuccess) continue;
        iv.emplace_back((double)a, (double)b);
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0.0, cur_a = 0.0, cur_b = -1.0;
    for (auto &p : iv) {
        if (cur_b < cur_a || p.first > cur_b) {
            if (cur_b >= cur_a) busy += cur_b - cur_a;
            cur_a = p.first;
            cur_b = p.second;
        } else if (p.second > cur_b) {
            cur_b = p.second;
        }
    }
    if (cur_b >= cur_a) busy += cur_b - cur_a;
    *busy_ms = busy;
    return 0;
}

extern "C" int gp_gemm_trace(gp_t *g, int cap, int64_t *tiles, int *K, double *ms) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    int n = (int)std::min<size_t>((size_t)cap, g->gemm_tiles.size());
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        hipEventElapsedTime(&t, g->gemm_events[2 * i], g->gemm_events[2 * i + 1]);
        if (tiles) tiles[i] = g->gemm_tiles[i];
        if (K) K[i] = g->gemm_K[i];
        if (ms) ms[i] = t;
    }
    return n;
}

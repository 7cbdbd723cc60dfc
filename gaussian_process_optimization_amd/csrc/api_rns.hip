// Host side of the fp64-equivalent contractions on the int8 matrix cores (option "emulate_fp64", kernels in rns.hip):
// fixed-point geometry, residue planes of L, the emulated row solve and the emulated Ky^-1.
#include "api_internal.h"

// residue GEMM launch with the same accounting (gp_profile): events on the launch's own stream, int8 operations of the
// blocks the launch really computes (a triangular launch skips the blocks above the diagonal)
void rns_gemm(gp_ctx *g, hipStream_t s, const signed char *A, long lda, long a_plane, const signed char *B, long ldb, long b_plane, signed char *R, int mt_all, int nt_all, int mt, int c0, int c1, int K, int first, int tri) {
    if (mt <= 0 || c1 <= c0 || K <= 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g->profiling) {
        if (g->rns_ev_used + 2 > g->rns_events.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            g->rns_events.push_back(a);
            g->rns_events.push_back(b);
        }
        e0 = g->rns_events[g->rns_ev_used++];
        e1 = g->rns_events[g->rns_ev_used++];
        long blocks = 0;
        for (int c = c0; c < c1; ++c) blocks += tri ? std::max(0, mt - c) : mt;
        g->rns_ops += 2.0 * 65536.0 * (double)K * (double)blocks * GP_RNS_T;
        GP_NOTE(hipEventRecord(e0, s));
    }
    launch_rns_gemm256(s, A, lda, a_plane, B, ldb, b_plane, R, mt_all, nt_all, mt, c0, c1, K, first, tri);
    if (e1) GP_NOTE(hipEventRecord(e1, s));
}

int rns_prepare(gp_ctx *g, double jitter, RnsGeom *r) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = std::min(g->panel_tiles, nt);
    if ((long)W * GP_TILE > GP_RNS_KMAX) return fail(GP_ERR_ARG, "emulate_fp64: panel_tiles too wide for one residue contraction");
    if (g->N > (1L << 20)) return fail(GP_ERR_ARG, "emulate_fp64 needs N <= 2^20");
    if (rns_init_constants(g->device)) return fail(GP_ERR_HIP, "rns constants");
    r->Lrows = round_up(Npad, 256);
    // (a row pitch off the power of two was measured: no channel conflicts, no effect)
    r->Lpitch = Npad;
    r->Lplane = r->Lrows * r->Lpitch;
    r->nt256 = (int)(r->Lrows / 256);
    // common power-of-two scale: |L_ij| <= sqrt(max diag of Ky), |S_ik| <= sqrt(prior variance); one spare bit
    const double diag0 = (g->kp.gower ? std::pow(g->kp.variance, g->D) : g->kp.variance) + g->noise + 1e-8 + jitter;
    int e = 1 + (int)std::ceil(std::log2(std::sqrt(std::max(diag0, 1e-300))));
    if (e < 0) e = 0;
    r->e = e;
    r->scale = std::ldexp(1.0, 52 - e);
    r->back = std::ldexp(1.0, 2 * e);
    const long need = (long)GP_RNS_T * r->Lplane;
    if (need > g->capLr || !g->dLr) {
        int rc = byte_realloc(&g->dLr, &g->capLr, need);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(g->dLr, 0, (size_t)need, g->s));
        GP_SYNC(g->s);
        g->lr_valid = false;
    }
    const int nJ = (nt + W - 1) / W;
    if (!g->lr_valid || g->lr_W != W || g->lr_e != e || (int)g->lr_done.size() != nJ) {
        g->lr_done.assign(nJ, 0);
        g->lr_W = W;
        g->lr_e = e;
        g->lr_valid = true;
    }
    return 0;
}

// residues of L's panel J (rows strictly below its diagonal block), once per factor
void rns_convert_panel(gp_ctx *g, hipStream_t s, const RnsGeom &r, int J, int *flag) {
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE), W = g->lr_W;
    const int J0 = J * W, J1 = std::min(J0 + W, nt);
    if (J1 >= nt || g->lr_done[J]) return;
    launch_rns_convert(s, g->dA + (long)J1 * GP_TILE * lda + (long)J0 * GP_TILE, lda, Npad - (long)J1 * GP_TILE,
                       (long)(J1 - J0) * GP_TILE, g->dLr + (long)J1 * GP_TILE * r.Lpitch + (long)J0 * GP_TILE, r.Lplane, r.Lpitch,
                       r.scale, flag);
    g->lr_done[J] = 1;
}

int solve_rows_rns(gp_ctx *g, double *T, double *S, int mt, const RnsSolveOpt &opt) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    const long PB = (long)W * GP_TILE, Mcpad = (long)mt * GP_TILE;
    int rc;
    RnsGeom r;
    if (W != std::min(g->panel_tiles, nt)) return fail(GP_ERR_STATE, "emulate_fp64: panel width changed since the fit");
    if ((rc = rns_prepare(g, g->jitter, &r))) return rc;
    // 256 x 256 workgroup tiles: rows / columns padded to multiples of 256 (zero residues in the padding)
    const long Mc256 = round_up(Mcpad, 256);
    const int mt256 = (int)(Mc256 / 256), nt256 = r.nt256;
    // Panels are taken in groups of G (option "rns_group"): their S panels sit side by side in the residue buffer, the
    // columns of group panel i take ONE launch that contracts the i panels before it (K = i PB), and ONE launch then
    // contracts the whole group (K = G PB) into every column right of it.  Each accumulator block is therefore read,
    // reduced and written once per group instead of once per panel, and the contraction is G times as long; the
    // products summed are the same integers, so the result does not depend on G.  Panel edges must sit on 256-column
    // accumulator blocks for the launches of one group to touch disjoint blocks: odd panel widths take G = 1.
    int G = std::max(1, std::min(g->rns_group, (int)(GP_RNS_KMAX / PB)));
    if (W % 2) G = 1;
    const bool keep = opt.Wr != nullptr;
    const long KS = keep ? opt.wpitch : G * PB;   // row pitch of the S planes
    auto zalloc = [&](signed char **p, long *cap, long need) -> int {
        if (need <= *cap && *p) return 0;
        int r2 = byte_realloc(p, cap, need);
        if (r2) return r2;
        if (hipMemsetAsync(*p, 0, (size_t)need, g->s) != hipSuccess) return fail(GP_ERR_HIP, "hipMemsetAsync");
        return 0;
    };
    if (PB > GP_RNS_KMAX) return fail(GP_ERR_ARG, "emulate_fp64: panel_tiles too wide for one residue contraction");
    if (!keep && (rc = zalloc(&g->dSr, &g->capSr, (long)GP_RNS_T * Mc256 * KS))) return rc;
    if ((rc = zalloc(&g->dRr, &g->capRr, (long)GP_RNS_T * mt256 * nt256 * 65536))) return rc;
    hipStream_t s = g->s;
    int *flag = g->dInfo + 2;
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), s));
    const int eS = opt.eS >= 0 ? opt.eS : r.e;
    const double scaleS = std::ldexp(1.0, 52 - eS), back = std::ldexp(1.0, eS + r.e);
    const long Lplane = r.Lplane, Splane = keep ? opt.wplane : Mc256 * KS;
    signed char *Sr = keep ? opt.Wr : g->dSr;
    for (int J = 0; J < (int)g->lr_done.size(); ++J) rns_convert_panel(g, s, r, J, flag);
    // trapezoid: a launch's row blocks are those that hold a non-zero row of its S panels; blocks the accumulator has
    // never seen must read as zero, so it starts zeroed and no launch overwrites ("first")
    if (opt.trapezoid) HIPCHK(hipMemsetAsync(g->dRr, 0, (size_t)GP_RNS_T * mt256 * nt256 * 65536, s));
    auto rows_of = [&](int Jend) { return opt.trapezoid ? std::min(mt, Jend) : mt; };   // row tiles (128) of a step
    auto panel_solve = [&](int Ja, int Jb, int Jidx) {   // S[:, Ja..Jb) = T[:, Ja..Jb) invP^T, fp64
        GemmOpt o;
        o.k_end_tri = 1;
        o.b_sub = Ja;
        gemm(g, s, 0, S, Npad, T + (long)Ja * GP_TILE, Npad, g->dInvP + (long)Jidx * PB * PB, PB, 1, (Jb - Ja) * GP_TILE,
             TileSet{0, rows_of(Jb), Ja, Jb, 0}, o);
    };
    bool first = !opt.trapezoid;   // no launch has written the accumulator yet: the first group's launches overwrite their blocks
    for (int J0 = 0, J = 0; J0 < nt;) {
        int done = 0;    // panels of this group solved and converted; they span tiles [J0, Ja)
        bool last = false;
        for (int i = 0; i < G; ++i) {
            const int Ja = J0 + i * W, Jb = std::min(Ja + W, nt);
            if (Ja >= nt) break;
            const int rt = rows_of(Jb), rb = (rt + 1) / 2;                 // rows of this panel's steps: tiles, 256-blocks
            const long rrows = (long)rt * GP_TILE;
            // the group's earlier panels -> this panel's columns (whole 256-column blocks: Ja, Jb are even); their S rows
            // beyond tile Ja are zero, so are the products: the launch stops at the blocks that hold rows < Ja
            if (i > 0)
                rns_gemm(g, s, Sr + (keep ? (long)J0 * GP_TILE : 0), KS, Splane, g->dLr + (long)J0 * GP_TILE, r.Lpitch, Lplane,
                         g->dRr, mt256, nt256, opt.trapezoid ? (std::min(mt, Ja) + 1) / 2 : mt256, Ja / 2, (Jb + 1) / 2,
                         (Ja - J0) * GP_TILE, first ? 1 : 0);
            if (opt.trapezoid ? (J0 > 0 || i > 0) : (!first || i > 0))
                launch_rns_reconstruct256(s, g->dRr, mt256, nt256, opt.trapezoid ? rb : mt256, Ja, Jb,
                                          opt.trapezoid ? std::min(rrows, Mcpad) : Mcpad, T, Npad, back);
            panel_solve(Ja, Jb, J + i);
            done = i + 1;
            if (Jb >= nt) { last = true; break; }
            launch_rns_convert(s, S + (long)Ja * GP_TILE, Npad, opt.trapezoid ? rrows : Mcpad, (Jb - Ja) * GP_TILE,
                               Sr + (keep ? (long)Ja * GP_TILE : (long)i * PB), Splane, KS, scaleS, flag);
        }
        if (last) {
            // the last panel's residues are still wanted by the product W W^T
            if (keep) {
                const int Ja = J0 + (done - 1) * W, Jb = std::min(Ja + W, nt);
                launch_rns_convert(s, S + (long)Ja * GP_TILE, Npad, (long)rows_of(Jb) * GP_TILE, (Jb - Ja) * GP_TILE,
                                   Sr + (long)Ja * GP_TILE, Splane, KS, scaleS, flag);
            }
            break;
        }
        const int Jg = J0 + done * W;   // < nt here
        // the whole group -> every column right of it
        rns_gemm(g, s, Sr + (keep ? (long)J0 * GP_TILE : 0), KS, Splane, g->dLr + (long)J0 * GP_TILE, r.Lpitch, Lplane, g->dRr,
                 mt256, nt256, opt.trapezoid ? (std::min(mt, Jg) + 1) / 2 : mt256, G == 1 ? Jg / 2 : (Jg + 1) / 2, nt256,
                 (Jg - J0) * GP_TILE, first ? 1 : 0);
        if (!opt.trapezoid) first = false;
        J0 = Jg;
        J += done;
    }
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    GP_SYNC(s);
    if (bad) return GP_ERR_RANGE;   // (internal: the caller repeats the solve in true fp64)
    return 0;
}

// Ky^-1 in residue form ("emulate_fp64"): W = L^-T by the emulated solve of the identity (trapezoid; the residues of every
// solved panel stay in planes of N columns), then Ky^-1 = W W^T as residue launches over groups of k panels -- group
// [k0, k1) adds to the blocks (i, c), c <= i, whose rows lie above tile k1 -- into a zeroed accumulator, one CRT
// reconstruction of the lower blocks at the end and the same symmetrisation as the fp64 path.  Bound (rns.hip): the rows
// of W have norm sqrt((Ky^-1)_ii) <= 1 / sqrt(noise + 1e-8 + jitter) = 2^(eS-1) at most, so both contractions stay below
// 2^102 in integer units.  Reference: dtrtri + dpotri, GPy/GPy/util/linalg.py:127-145,193-214.
int wi_rns(gp_ctx *g) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    hipStream_t s = g->s;
    int rc;
    RnsGeom r;
    if ((rc = rns_prepare(g, g->jitter, &r))) return rc;
    const int nt256 = r.nt256;
    const double lam = g->noise + 1e-8 + g->jitter;
    if (!(lam > 0.0)) return GP_ERR_RANGE;
    const int eS = std::max(0, 1 + (int)std::ceil(std::log2(1.0 / std::sqrt(lam))));
    const long wpitch = Npad, wrows = (long)nt256 * 256, wplane = wrows * wpitch;
    if ((rc = byte_realloc(&g->dWr, &g->capWr, (long)GP_RNS_T * wplane))) return rc;
    HIPCHK(hipMemsetAsync(g->dWr, 0, (size_t)GP_RNS_T * wplane, s));
    int ph = phase_begin(g, "potri_solve_emulated", (double)g->N * g->N * g->N / 3.0, 0.0);
    launch_set_identity(s, g->dT, Npad, Npad);
    RnsSolveOpt o;
    o.trapezoid = true;
    o.eS = eS;
    o.Wr = g->dWr;
    o.wpitch = wpitch;
    o.wplane = wplane;
    if ((rc = solve_rows_rns(g, g->dT, g->dT2, nt, o))) return rc;
    phase_end(g, ph);
    ph = phase_begin(g, "potri_lauum_emulated", (double)g->N * g->N * g->N / 3.0, 0.0);
    HIPCHK(hipMemsetAsync(g->dRr, 0, (size_t)GP_RNS_T * nt256 * nt256 * 65536, s));
    const long PB = (long)W * GP_TILE;
    const int G = std::max(1, std::min(g->rns_group, (int)(GP_RNS_KMAX / PB)));
    for (int k0 = 0; k0 < nt; k0 += G * W) {
        const int k1 = std::min(k0 + G * W, nt);
        const int rb = (k1 + 1) / 2;   // row blocks that hold a non-zero row of these columns of W
        rns_gemm(g, s, g->dWr + (long)k0 * GP_TILE, wpitch, wplane, g->dWr + (long)k0 * GP_TILE, wpitch, wplane, g->dRr, nt256, nt256,
                 rb, 0, rb, (k1 - k0) * GP_TILE, 0, 1);
    }
    HIPCHK(hipMemsetAsync(g->dWi, 0, sizeof(double) * Npad * Npad, s));
    launch_rns_reconstruct256(s, g->dRr, nt256, nt256, nt256, 0, nt, Npad, g->dWi, Npad, -std::ldexp(1.0, 2 * eS), 1);
    launch_symmetrize(s, g->dWi, Npad, Npad);
    phase_end(g, ph);
    return 0;
}

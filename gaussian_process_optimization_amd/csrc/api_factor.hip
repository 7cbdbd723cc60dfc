// The factorisation scheduler: blocked right-looking Cholesky with one panel of look-ahead over the per-device streams,
// the pipelined one-call entry points (gp_fit_predict, the fit half of gp_fit_grad) and gp_fit itself.
// Reference: jitchol / pdinv (GPy/GPy/util/linalg.py:56-81,193-214), ExactGaussianInference.inference
// (inference/latent_function_inference/exact_gaussian_inference.py:37-74).
#include "api_internal.h"

// ---- blocked right-looking Cholesky (two-level: 128-column steps inside panel_tiles-wide panels) ----
// A: nt x nt tiles (lower) plus R1 - nt extra row tiles that ride through the panel solves and updates (the RHS rows)
// side_inv (the model's own factor only): the inverted diagonal panel of each panel is built on the side stream as soon as that
// panel's columns are final, beside the trailing update and the next panel -- one event record per panel on this stream.
// One step of the in-panel factorisation starting at tile column j of a panel that ends at J1; returns the next column.
// Option "inner_tiles" 2 (default 1: measured neutral) takes two columns at a time: the 256 x 256 diagonal block in ONE launch
// (potrf_pair_kernel: both diagonal tiles, the tile between them solved and the second one updated inside), ONE launch that
// solves both tile columns of the rows below (trsm2.hip), ONE K = 256 update of the panel's remaining columns -- three
// dependent launches per 256 columns where the 128-column step takes six, and contractions twice as long.
static int chain_step(gp_ctx *g, hipStream_t s, double *A, long lda, int j, int J1, int R1, double *invL, int *info) {
    if (g->inner_tiles >= 2 && j + 1 < J1 && R1 - (j + 2) >= g->inner_min_rows) {
        launch_potrf_pair(s, A, lda, j, invL, info);
        launch_trsm2(s, A, lda, j, invL, j + 2, R1);
        if (j + 2 < J1)
            gemm(g, s, 1, A, lda, A + (long)j * GP_TILE, lda, A + (long)j * GP_TILE, lda, 1, 2 * GP_TILE,
                 TileSet{0, R1, j + 2, J1, 1});
        return j + 2;
    }
    launch_potrf_tile(s, A, lda, j, invL, info);
    // panel solve: A[i, j] <- A[i, j] * inv(L_jj)^T for the row tiles below (and the RHS tile)
    gemm(g, s, 0, A, lda, A + (long)j * GP_TILE, lda, invL + (long)j * GP_TILE * GP_TILE, GP_TILE, 0, GP_TILE,
         TileSet{j + 1, R1, j, j + 1, 0}, inplace_opt());
    // update of the remaining columns of this panel (K = 128)
    if (j + 1 < J1)
        gemm(g, s, 1, A, lda, A + (long)j * GP_TILE, lda, A + (long)j * GP_TILE, lda, 1, GP_TILE, TileSet{0, R1, j + 1, J1, 1});
    return j + 1;
}

void factor_buf(gp_ctx *g, double *A, long lda, int nt, int R1, double *invL, int *info, bool side_inv) {
    const int W = g->panel_tiles;
    hipStream_t s = g->s;
    if (side_inv) {
        hipEvent_t e0 = la_event(g, EV_MISC, 0);
        GP_NOTE(hipEventRecord(e0, s));
        GP_NOTE(hipStreamWaitEvent(g->s_inv, e0, 0));
    }
    for (int J0 = 0; J0 < nt; J0 += W) {
        const int J1 = std::min(J0 + W, nt);
        for (int j = J0; j < J1;) j = chain_step(g, s, A, lda, j, J1, R1, invL, info);
        if (side_inv) {
            hipEvent_t eF = la_event(g, EV_CHAIN, J0 / W);
            GP_NOTE(hipEventRecord(eF, s));
            GP_NOTE(hipStreamWaitEvent(g->s_inv, eF, 0));
            build_panel_inv_one(g, g->s_inv, J0 / W, W, nt);
        }
        // trailing update with the whole panel (K = W * 128): the dense contraction on MFMA
        if (J1 < nt)
            gemm(g, s, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, (J1 - J0) * GP_TILE,
                 TileSet{0, R1, J1, nt, 1});
    }
    if (side_inv) {
        hipEvent_t ei = la_event(g, EV_MISC, 4);
        GP_NOTE(hipEventRecord(ei, g->s_inv));
        GP_NOTE(hipStreamWaitEvent(s, ei, 0));
    }
}

// The single-stream factorisation of the model's Ky (N <= 768, and the sizes where the look-ahead's per-panel events, waits and
// separate look-ahead launch cost more than the overlap gives: measured 1.99 vs 2.31 ms at N = 4096, 0.84 vs 1.05 at N = 2048,
// equal from N = 5120 on; same arithmetic, bitwise the same factor).  Panels wider than the matrix take the batched inverse build.
int factor(gp_ctx *g) {
    const int nt = (int)(g->Npad / GP_TILE);
    const int W = g->panel_tiles;
    const bool side = nt > W && g->s_inv;
    if (side) {
        const long PB = (long)W * GP_TILE;
        const int nJ = (nt + W - 1) / W;
        int rc;
        if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJ * PB * PB))) return rc;
        if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJ * PB * PB))) return rc;
    }
    factor_buf(g, g->dA, g->Npad, nt, nt + 1, g->dInvL, g->dInfo, side);
    if (side) {
        g->invp_W = W;
        g->invp_valid = true;   // (fit_impl drops it again when the attempt turns out not positive definite)
        return la_events_ok(g);
    }
    return 0;
}

void build_panel_inv_one(gp_ctx *g, hipStream_t s, int J, int W, int nt) {
    const long lda = g->Npad;
    const long PB = (long)W * GP_TILE;
    const int J0 = J * W, Wp = std::min(W, nt - J0);
    double *Wb = g->dInvPw + (long)J * PB * PB;
    const double *Lb = g->dA + (long)J * (PB * lda + PB);
    const double *Ib = g->dInvL + (long)J0 * GP_TILE * GP_TILE;
    launch_set_identity_blocks(s, Wb, PB, 1);
    for (int b = 0; b < Wp; ++b) {
        gemm(g, s, 0, Wb, PB, Wb + (long)b * GP_TILE, PB, Ib + (long)b * GP_TILE * GP_TILE, GP_TILE, 0, GP_TILE,
             TileSet{0, b + 1, b, b + 1, 0}, inplace_opt());
        if (b + 1 < Wp)
            gemm(g, s, 1, Wb, PB, Wb + (long)b * GP_TILE, PB, Lb + (long)b * GP_TILE, lda, 1, GP_TILE,
                 TileSet{0, b + 1, b + 1, Wp, 0});
    }
    launch_transpose_blocks(s, g->dInvP + (long)J * PB * PB, Wb, PB, 1);
}

static inline long Npad_rows(gp_ctx *g) { return g->Npad; }

int factor_lookahead(gp_ctx *g, const PredPipe &pp) {
    int rc;
    if ((rc = ensure_bulk_stream(g))) return rc;
    const long lda = g->Npad;
    const int nt = (int)(g->Npad / GP_TILE);
    const int R1 = nt + 1;
    const int W = g->panel_tiles;
    double *A = g->dA;
    hipStream_t sp = g->s_panel, sb = g->s_bulk;
    // fork
    hipEvent_t e0 = la_event(g, EV_MISC, 0);
    GP_NOTE(hipEventRecord(e0, g->s));
    GP_NOTE(hipStreamWaitEvent(sp, e0, 0));
    GP_NOTE(hipStreamWaitEvent(sb, e0, 0));
    const long PB = (long)W * GP_TILE;
    const int nJu = (nt + W - 1) / W;
    // Every inverted diagonal panel (alpha, the candidate solve and Ky^-1 all need them) is built on the side stream as soon
    // as its panel of L is final, beside the rest of the factorisation: after the join nothing is left to build (as a pass of
    // its own, 2W - 1 short launches in series, it held the main stream for 0.3 ms between the factor and its first consumer).
    if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJu * PB * PB))) return rc;
    if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJu * PB * PB))) return rc;
    GP_NOTE(hipStreamWaitEvent(g->s_inv, e0, 0));
    if (pp.on) {
        GP_NOTE(hipStreamWaitEvent(g->s_pred, e0, 0));
        if (pp.init) pp.init(g->s_pred);
    }
    int next_pred = 0;
    int own_prev = nt;   // columns owned by the chain stream at the previous panel (the range never grows)
    // Only the first `pipe_stages` candidate stages ride behind the factorisation (on the CU-masked stream, released
    // at panel pred_start); the caller runs the rest on the main stream, on every CU, once the factor is complete.
    const int pstages = pp.on ? std::max(1, std::min(nJu, pp.stages)) : 0;
    const int pred_start = std::max(0, std::min(nJu - 1, nJu * pp.start_pct / 100));
    // panel boundaries (uniform panels of W tiles; two sentinels)
    std::vector<int> pb;
    for (int j = 0; j < nt; j += W) pb.push_back(j);
    pb.push_back(nt);
    pb.push_back(nt);
    const int nJ = (int)pb.size() - 2;
    // "emulate_fp64": the trailing update (the launches of the bulk stream) in residue form on the int8 matrix cores
    // (rns.hip).  The Schur complement right of the look-ahead panel lives as Ky (untouched, in dA) minus an exact integer
    // accumulator dRm; a panel's columns are rebuilt in fp64 once, right before they become the look-ahead target.  The
    // chain (diagonal tiles, panel solves, in-panel and look-ahead updates) and the right-hand-side tile row stay fp64.
    const bool emu = emu_fit_applies(g);
    RnsGeom rg;
    int *rflag = g->dInfo + 2;
    if (emu) {
        if ((rc = rns_prepare(g, g->jitter_try, &rg))) return rc;
        const long need = (long)GP_RNS_T * rg.nt256 * rg.nt256 * 65536;
        if ((rc = byte_realloc(&g->dRm, &g->capRm, need))) return rc;
    }
    // emulated: panels per residue launch (the far launches ride on the otherwise idle candidate stream)
    const int Gf = (emu && !pp.on) ? std::max(1, std::min(g->rns_group_fit, (int)(GP_RNS_KMAX / PB))) : 1;
    hipStream_t sfar = g->s_pred;
    std::vector<char> far_issued(nJ / std::max(1, Gf) + 2, 0);
    int mid_Jg = -1, mid_end = 0, mid_first = 0, mid_base = 0, mid_next = 0;
    if (Gf > 1) GP_NOTE(hipStreamWaitEvent(sfar, e0, 0));
    for (int J = 0; J < nJ; ++J) {
        const int J0 = pb[J], J1 = pb[J + 1], J2 = pb[J + 2];
        bool bulk_recorded = false;
        for (int j = J0; j < J1;) j = chain_step(g, sp, A, lda, j, J1, R1, g->dInvL, g->dInfo);
        hipEvent_t eF = la_event(g, EV_CHAIN, J);
        GP_NOTE(hipEventRecord(eF, sp));
        const int K = (J1 - J0) * GP_TILE;
        GP_NOTE(hipStreamWaitEvent(g->s_inv, eF, 0));
        build_panel_inv_one(g, g->s_inv, J, W, nt);
        if (pp.on) {
            if (J < pstages) GP_NOTE(hipEventRecord(la_event(g, EV_INVP, J), g->s_inv));
            // Two concurrent MFMA-bound launches run slower than one after the other (measured 51 vs 63 TFLOP/s), and
            // the candidate stream is CU-masked like the trailing update (the diagonal-tile workgroup needs an empty
            // CU), which costs it 1/8 of the chip.  So only the first `pipe_stages` stages ride here, released once
            // the factorisation turns latency-bound (panel >= pred_start): they fill the CUs the chain leaves idle in
            // the tail.  The rest run after the join on the main stream, on every CU (fit_impl).  Measured at C3:
            // 73.4 ms against 77.0 for gp_fit + gp_predict; every stage pipelined: 78.1.
            if (J >= pred_start) {
                for (; next_pred <= J && next_pred < pstages; ++next_pred) {
                    const int Q = next_pred, Q0 = Q * W, Q1 = std::min(Q0 + W, nt);
                    const int KQ = (Q1 - Q0) * GP_TILE;
                    const int prow = pp.trapezoid ? std::min(pp.mt, Q1) : pp.mt;
                    GP_NOTE(hipStreamWaitEvent(g->s_pred, la_event(g, EV_CHAIN, J), 0));
                    GP_NOTE(hipStreamWaitEvent(g->s_pred, la_event(g, EV_INVP, Q), 0));
                    GemmOpt o;
                    o.k_end_tri = 1;
                    o.b_sub = Q0;
                    gemm(g, g->s_pred, 0, pp.S, g->Npad, pp.T + (long)Q0 * GP_TILE, g->Npad, g->dInvP + (long)Q * PB * PB,
                         PB, 1, KQ, TileSet{0, prow, Q0, Q1, 0}, o);
                    if (Q1 < nt)
                        gemm(g, g->s_pred, 1, pp.T, g->Npad, pp.S + (long)Q0 * GP_TILE, g->Npad, A + (long)Q0 * GP_TILE, lda,
                             1, KQ, TileSet{0, prow, Q1, nt, 0});
                }
            }
        }
        if (J1 >= nt) break;
        // Columns owned by the chain stream (options own_keep_*): in the head of the factorisation the chain finishes panel J+1 long
        // before bulk(J) has drained and would wait for it, with the CUs kept free of the bulk stream idle.  bulk(J) therefore keeps
        // only what lasts as long as the chain is busy with the next panel (a count of tiles linear in the rows below it); the
        // rest -- the last tile columns oc .. nt -- takes panel J's update on THIS stream, after chain(J), on every CU.  The owned
        // range only shrinks with J, so own(J) never meets a tile bulk(J-1) writes, and a column handed back to the bulk stream
        // had its last update here before chain(J+1), which bulk(J+1) waits for.  Same contraction per tile in the same order:
        // the same bits as without.
        int oc = nt;
        if (!emu && g->own_keep_per_row > 0 && J2 < nt) {
            const long n = nt - J2;
            // tiles right of J2 (with the rhs row) minus the kept ones (the chain's time per panel grows with the panel width: per_row is per 6 tiles)
            long keep = g->own_keep_base + (long)g->own_keep_per_row * n * W / 6;
            // (one-call entry points: once the candidate stages share the bulk stream's CUs the trailing update lasts longer and the
            // chain waits again; the bulk stream then keeps own_keep_pipe_pct % of the rule's share)
            // (never at the FIRST panel that owns columns: there own_prev is still its initial nt, and keep = 0 would hand the
            // whole trailing update to the chain stream -- pipe_start_pct = 0, or so few panels that pred_start rounds to 0)
            if (pp.on && J >= pred_start && own_prev < nt) keep = keep * g->own_keep_pipe_pct / 100;
            long t_own = n * (n + 1) / 2 + n - keep;
            int c = 0;
            while (c < own_prev && (long)(c + 1) * (c + 2) / 2 + (c + 1) <= t_own) ++c;
            own_prev = c;
            oc = nt - c;
        }
        if (oc < nt && J >= 1)
            gemm(g, sp, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K, TileSet{0, R1, oc, nt, 1});
        // the look-ahead update is on the critical path: enqueue it before the trailing update so that its
        // workgroups reach the dispatcher first once bulk(J-1) has drained
        if (J >= 1) GP_NOTE(hipStreamWaitEvent(sp, la_event(g, EV_BULK, J - 1), 0));
        // (emulated: the look-ahead panel's columns took everything the residue accumulator holds for them -- panels
        // 0 .. J-1 -- on the bulk stream, before bulk(J-1) was recorded)
        gemm(g, sp, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
             TileSet{0, R1, J1, J2, 1});
        if (oc < nt && J == 0)   // (the first panel has no bulk launch to wait for: the critical update goes first)
            gemm(g, sp, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K, TileSet{0, R1, oc, nt, 1});
        if (J2 < nt) {
            GP_NOTE(hipStreamWaitEvent(sb, eF, 0));
            if (emu) {
                rns_convert_panel(g, sb, rg, J, rflag);
                // the right-hand-side tile row rides in fp64
                gemm(g, sb, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
                     TileSet{nt, R1, J2, nt, 0});
                auto rlaunch = [&](hipStream_t st, int Jfirst, int t0, int t1, int first, int Tend = -1) {   // panels Jfirst..J (tiles up to Tend) -> tiles [t0, t1)
                    t1 = std::min(t1, nt);
                    if (t0 >= t1) return;
                    const int T0 = pb[Jfirst];
                    rns_gemm(g, st, g->dLr + (long)T0 * GP_TILE, rg.Lpitch, rg.Lplane, g->dLr + (long)T0 * GP_TILE, rg.Lpitch,
                                       rg.Lplane, g->dRm, rg.nt256, rg.nt256, rg.nt256, t0 / 2, (t1 + 1) / 2,
                                       ((Tend < 0 ? J1 : Tend) - T0) * GP_TILE, first, 1);
                };
                auto pbi = [&](int k) { return pb[std::min(k, nJ + 1)]; };
                // rebuild the columns of panel J+2 in fp64 (Ky minus everything accumulated for them: panels 0 .. J) as soon as
                // the last residue launch into them is enqueued -- on this stream, off the chain
                auto rebuild_next = [&]() {
                    if (pbi(J + 2) < nt)
                        launch_rns_reconstruct256(sb, g->dRm, rg.nt256, rg.nt256, rg.nt256, pbi(J + 2), std::min(pbi(J + 3), nt),
                                                  Npad_rows(g), A, lda, rg.back, 1);
                };
                if (Gf == 1) {
                    rlaunch(sb, J, J2, nt, J == 0 ? 1 : 0);
                    rebuild_next();
                } else {
                    // Panels in groups of Gf (all panel edges sit on 256-column accumulator blocks).  Pair (panel j, column
                    // panel c >= j+2; c = j+1 is the fp64 look-ahead) is served exactly once, by
                    //   near(J)  on the bulk stream, every iteration: the group's panels so far -> the columns of panel J+2,
                    //   mid(g)   on the bulk stream, from the group's last panel on, ONE column panel per iteration: the whole group
                    //            -> the next Gf column panels, each slice an iteration before its columns are rebuilt (as one
                    //            launch of 3.7 ms at N = 16384 it sat in front of near(J+1) and the chain stalled 2.6 ms behind it),
                    //   far(g)   on a stream of its own: the whole group -> everything right of that,
                    // so the accumulator makes one round trip per group for the far columns and the long launch (K = Gf PB)
                    // overlaps the next group's chain.  Ordering: near(J) and mid(g) accumulate into blocks far(g-1) / far(g-2)
                    // wrote (mid waits for far(g-1); near follows mid(g-1) in stream order); far(g) follows far(g-1) in stream
                    // order; the chain's reconstruction of panel J+1's columns waits for bulk(J-1) = near(J-1), recorded
                    // BEFORE mid so that the chain does not wait for it.  The integers summed are those of Gf = 1.
                    const int gi = J / Gf, Jg = gi * Gf;
                    const int first = gi == 0 ? 1 : 0;
                    GP_NOTE(hipEventRecord(la_event(g, EV_CONV, J), sb));
                    rlaunch(sb, Jg, pbi(J + 2), pbi(J + 3), first);
                    rebuild_next();
                    GP_NOTE(hipEventRecord(la_event(g, EV_BULK, J), sb));
                    bulk_recorded = true;
                    if (J % Gf == Gf - 1) {
                        if (gi >= 1 && far_issued[gi - 1]) GP_NOTE(hipStreamWaitEvent(sb, la_event(g, EV_FAR, gi - 1), 0));
                        mid_Jg = Jg;            // the slices of mid(g): column panel mid_base + k at iteration J + k
                        mid_end = J1;
                        mid_first = first;
                        mid_base = J + 3;
                        mid_next = 0;
                        if (pbi(J + 3 + Gf) < nt) {
                            GP_NOTE(hipStreamWaitEvent(sfar, la_event(g, EV_CONV, J), 0));
                            rlaunch(sfar, Jg, pbi(J + 3 + Gf), nt, first);
                            GP_NOTE(hipEventRecord(la_event(g, EV_FAR, gi), sfar));
                            far_issued[gi] = true;
                        }
                    }
                    if (mid_Jg >= 0 && mid_next < Gf) {
                        rlaunch(sb, mid_Jg, pbi(mid_base + mid_next), pbi(mid_base + mid_next + 1), mid_first, mid_end);
                        ++mid_next;
                    }
                }
            } else {
                if (J2 < oc)
                    gemm(g, sb, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
                         TileSet{0, R1, J2, oc, 1});
            }
            if (!bulk_recorded) GP_NOTE(hipEventRecord(la_event(g, EV_BULK, J), sb));
        }
    }
    // join
    hipEvent_t ep = la_event(g, EV_MISC, 1), eb = la_event(g, EV_MISC, 2);
    GP_NOTE(hipEventRecord(ep, sp));
    GP_NOTE(hipEventRecord(eb, sb));
    GP_NOTE(hipStreamWaitEvent(g->s, ep, 0));
    GP_NOTE(hipStreamWaitEvent(g->s, eb, 0));
    if (Gf > 1) {   // (every far launch ends before the factor is complete: mid of the next group waits for it; join anyway)
        hipEvent_t ef = la_event(g, EV_MISC, 7);
        GP_NOTE(hipEventRecord(ef, sfar));
        GP_NOTE(hipStreamWaitEvent(g->s, ef, 0));
    }
    g->pipe_done = pstages;
    if (pp.on) {
        hipEvent_t eq = la_event(g, EV_MISC, 3);
        GP_NOTE(hipEventRecord(eq, g->s_pred));
        GP_NOTE(hipStreamWaitEvent(g->s, eq, 0));
    }
    hipEvent_t ei = la_event(g, EV_MISC, 4);
    GP_NOTE(hipEventRecord(ei, g->s_inv));
    GP_NOTE(hipStreamWaitEvent(g->s, ei, 0));
    g->invp_W = W;
    g->invp_valid = true;   // (fit_impl drops it again when the attempt turns out not positive definite)
    return la_events_ok(g);   // (an error return makes fit_impl quiesce every stream before it reports)
}

__global__ void dot_ay_kernel(const double *alpha, long lda_, const double *Y, long N, int P, double *out) {
    __shared__ double sh[16];
    const int p = blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < N; i += 1024) s = fma(alpha[p * lda_ + i], Y[i * P + p], s);
    // block reduce
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < 16; ++i) r += sh[i];
        out[p] = r;
    }
}

// Shared body of gp_fit and gp_fit_predict.  pipe != 0: the candidate solve of the resident candidates is
// pipelined behind the factorisation (PredPipe above) and the posterior reductions are appended.
// Any error return of fit_impl after work was forked onto the side streams must leave them joined: the guard waits for
// every stream of the context unless the normal exit (where the joins are stream-ordered) dismissed it.
struct QuiesceOnError {
    gp_ctx *g;
    bool armed = true;
    ~QuiesceOnError() {
        if (!armed) return;
        for (hipStream_t st : {g->s_panel, g->s_bulk, g->s_inv, g->s_pred, g->s})
            if (st) hipStreamSynchronize(st);
    }
};

int fit_impl(gp_ctx *g, int maxtries, int pipe, int include_noise) {
    HIPCHK(hipSetDevice(g->device));
    QuiesceOnError guard{g};
    const long N = g->N, Npad = g->Npad, lda = g->Npad;
    const int P = g->P;
    const int nt_ = (int)(Npad / GP_TILE);
    // pipe 1: candidate solve of the resident candidates; pipe 2: the solve of the identity (L^-T, for Ky^-1)
    const long mcpad = pipe == 1 ? round_up(g->M, GP_TILE) : (pipe == 2 ? Npad : 0);
    PredPipe pp;
    if (pipe) {
        int rc;
        const int W = std::min(g->panel_tiles, nt_);
        const long PB = (long)W * GP_TILE;
        const int nJ = (nt_ + W - 1) / W;
        if (pipe == 1 && (rc = ensure_out(g))) return rc;
        if ((rc = dev_realloc(&g->dT, &g->capT, mcpad * Npad))) return rc;
        if ((rc = dev_realloc(&g->dT2, &g->capT2, mcpad * Npad))) return rc;
        if (pipe == 2 && (rc = dev_realloc(&g->dWi, &g->capWi, Npad * Npad))) return rc;
        if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJ * PB * PB))) return rc;
        if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJ * PB * PB))) return rc;
        pp.on = true;
        pp.mt = (int)(mcpad / GP_TILE);
        pp.T = g->dT;
        pp.S = g->dT2;
        pp.trapezoid = (pipe == 2);
        // 0 = automatic: the share of the panels that was best at N = 16384 (3 of 22 candidate stages, 8 of 22 L^-T stages)
        pp.stages = pipe == 2 ? (g->pipe_stages_grad > 0 ? g->pipe_stages_grad : std::max(1, (nJ * 36 + 50) / 100))
                              : (g->pipe_stages > 0 ? g->pipe_stages : std::max(1, (nJ * 14 + 50) / 100) + (nJ <= 12 ? 1 : 0));  // small N: the chain is everything
        pp.start_pct = pipe == 2 ? g->pipe_start_pct_grad : (g->pipe_start_pct >= 0 ? g->pipe_start_pct : (nJ <= 24 ? 32 : 40));
    }
    const double diag_add = g->noise + 1e-8;  // exact_gaussian_inference.py:56
    const double diag0 = (g->kp.gower ? std::pow(g->kp.variance, g->D) : g->kp.variance) + diag_add;
    g->nphases = 0;
    g->emu_off_call = false;
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;

    double jitter = 0.0;
    int tries = 0;  // number of jittered attempts so far
    int info = 0;
    for (;;) {
        g->lr_valid = false;     // residue planes of L belong to one factorisation attempt
        g->jitter_try = jitter;
        int ph = phase_begin(g, "kbuild", 0.0, 8.0 * N * g->D + 8.0 * (double)N * N / 2);
        launch_kbuild(g->s, g->dA, lda, g->dX, N, Npad, g->kp, diag_add, 0);
        // jitchol retries factor (Ky + jitter I): the jitter lands on the assembled diagonal (linalg.py:69)
        if (jitter != 0.0) launch_add_diag(g->s, g->dA, lda, N, jitter);
        launch_set_rhs(g->s, g->dA, lda, g->dY, N, Npad, P);
        phase_end(g, ph);
        HIPCHK(hipMemsetAsync(g->dInfo, 0, sizeof(int) * 4, g->s));
        if (pipe == 1) {
            pp.init = [g, mcpad, N, Npad](hipStream_t st) {
                launch_cross_k(st, g->dT, Npad, g->dXs, g->M, mcpad, g->dX, N, Npad, g->kp);
            };
            ph = phase_begin(g, "cholesky+cand_solve", (double)N * N * N / 3.0 + (double)N * N * g->M, 0.0);
            int rcf = factor_lookahead(g, pp);
            if (rcf) return rcf;
        } else if (pipe == 2) {
            ph = phase_begin(g, "cholesky+potri_stages", (double)N * N * N / 3.0, 0.0);
            pp.init = [g, Npad](hipStream_t st) { launch_set_identity(st, g->dT, Npad, Npad); };
            int rcf = factor_lookahead(g, pp);
            if (rcf) return rcf;
        } else {
            ph = phase_begin(g, "cholesky", (double)N * N * N / 3.0, 0.0);
            // (the emulated trailing update lives in the look-ahead scheduler)
            if (g->lookahead && nt_ > g->panel_tiles && (nt_ > g->lookahead_min_tiles || emu_fit_applies(g))) {
                int rcf = factor_lookahead(g);
                if (rcf) return rcf;
            } else {
                int rcf = factor(g);
                if (rcf) return rcf;
            }
        }
        phase_end(g, ph);
        HIPCHK(hipMemcpyAsync(&info, g->dInfo, sizeof(int), hipMemcpyDeviceToHost, g->s));
        GP_SYNC(g->s);
        if (g->emulate_fp64 && !g->emu_off_call && info == 0) {
            int bad = 0;
            HIPCHK(hipMemcpy(&bad, g->dInfo + 2, sizeof(int), hipMemcpyDeviceToHost));
            if (bad) {
                // an entry of L outside the fixed-point range (non-finite data): the same attempt again in true fp64, whose
                // result is what the reference would return for such data
                g->emu_off_call = true;
                ++g->emu_fallbacks;
                g->nphases = 0;
                g->invp_valid = false;
                continue;
            }
        }
        if (info == 0) break;
        g->invp_valid = false;
        // jitter ladder, GPy/GPy/util/linalg.py:62-75
        if (!(diag0 > 0.0)) return fail(GP_ERR_NOT_PD_DIAG, "not pd: non-positive diagonal elements");
        if (tries == 0)
            jitter = diag0 * 1e-6;
        else
            jitter *= 10.0;
        ++tries;
        if (tries > maxtries || !std::isfinite(jitter)) {
            g_err = "not positive definite, even with jitter.";
            return info > 0 ? info : 1;
        }
        g->nphases = 0;
    }
    g->jitter = jitter;
    // alpha = L^-T z, log det and alpha'y: 45 short dependent launches (latency-bound, 1 ms).  When candidate / L^-T
    // stages are still to run on the main stream they go on the side stream instead, beside those long launches.
    bool side_alpha = false;
    auto alpha_lml = [&](hipStream_t st) {
        launch_logdet(st, g->dA, lda, N, g->dScal);
        launch_trsv_backward(st, g->dA, lda, g->dInvP, g->invp_W, Npad, g->dA + Npad * lda, lda, P, g->dAlpha, g->dW);
        GP_LAUNCH(dot_ay_kernel, dim3(P), dim3(1024), 0, st, g->dAlpha, Npad, g->dY, N, P, g->dScal + 8);
    };
    if (pipe) {
        const int W = std::min(g->panel_tiles, nt_);
        const int nJ = (nt_ + W - 1) / W;
        if (g->pipe_done < nJ) {  // the remaining stages on the main stream, every CU
            int phr = phase_begin(g, pipe == 2 ? "potri_solve_rest" : "cand_solve_rest", 0.0, 0.0);
            int rci = ensure_panel_inv(g);
            if (rci) return rci;
            hipEvent_t eI = la_event(g, EV_MISC, 5);
            if (g->s_inv && g->side_alpha) GP_NOTE(hipEventRecord(eI, g->s));
            // the long launches first: the 45 short launches of alpha / log det take the host 0.7 ms to enqueue, during which
            // the main stream sat empty when they went first
            solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), pipe == 2 ? 1 : 0, g->pipe_done);
            if (g->s_inv && g->side_alpha) {
                GP_NOTE(hipStreamWaitEvent(g->s_inv, eI, 0));
                alpha_lml(g->s_inv);
                GP_NOTE(hipEventRecord(la_event(g, EV_MISC, 6), g->s_inv));
                side_alpha = true;
            }
            phase_end(g, phr);
        }
        if (pipe == 2) {
            int rcl = wi_lauum(g);
            if (rcl) return rcl;
            g->wi_valid = true;
            g->w_in_t2 = true;   // dT2 = L^-T of this factor: ensure_linv transposes it instead of solving again
        }
    }

    int ph = phase_begin(g, "alpha_lml", 2.0 * (double)N * N * P, 8.0 * (double)N * N / 2);
    if (side_alpha) {
        GP_NOTE(hipStreamWaitEvent(g->s, la_event(g, EV_MISC, 6), 0));
    } else {
        int rci = ensure_panel_inv(g);
        if (rci) return rci;
        alpha_lml(g->s);
    }
    phase_end(g, ph);
    if (pipe == 1) {
        ph = phase_begin(g, "reduce", 0.0, 8.0 * (double)N * g->M);
        launch_predict_reduce(g->s, g->dT2, Npad, g->M, N, g->dA + Npad * Npad, Npad, P, g->kp.variance,
                              include_noise ? g->noise : 0.0, g->dMean, g->dVar);
        phase_end(g, ph);
    }
    std::vector<double> sc(8 + P);
    HIPCHK(hipMemcpyAsync(sc.data(), g->dScal, sizeof(double) * (8 + P), hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    double fit = 0.0;
    for (int p = 0; p < P; ++p) fit += sc[8 + p];
    g->logdet = sc[0];
    const double log_2_pi = std::log(2.0 * M_PI);
    g->lml = 0.5 * (-(double)N * P * log_2_pi - P * g->logdet - fit);  // exact_gaussian_inference.py:62
    g->fitted = true;
    if (pipe == 1) {
        g->predicted = true;
        g->predicted_noise = include_noise ? 1 : 0;
    }
    guard.armed = false;
    return 0;
}

extern "C" int gp_fit(gp_t *g, int maxtries, double *lml, double *logdet, double *jitter_used) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit");
    int rc = fit_impl(g, maxtries, 0, 0);
    if (rc) return rc;
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    return 0;
}

// gp_fit followed by gp_predict on the resident candidates, as ONE pipelined pass (the BO loop always runs
// them back to back: GPyOpt/GPyOpt/core/bo.py:236-254 then acquisitions/base.py:33-39).  Results are those of
// the two separate calls; the candidate solve merely overlaps the factorisation's latency-bound phases.
extern "C" int gp_fit_predict(gp_t *g, int maxtries, int include_noise, double *lml, double *logdet, double *jitter_used,
                   double *mean, double *var) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit_predict");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const int nt = (int)(g->Npad / GP_TILE);
    // the emulated candidate solve runs after the factorisation (its residue planes of L need the complete factor)
    // (a handful of candidates take the matrix-vector solve of gp_predict after the factorisation: both entry points then
    // return the same bits)
    const bool can_pipe = g->lookahead && nt > g->panel_tiles && round_up(g->M, GP_TILE) <= g->mc_max && !g->emulate_fp64 &&
                          g->M > g->small_m;
    int rc;
    if (can_pipe) {
        if ((rc = fit_impl(g, maxtries, 1, include_noise))) return rc;
    } else {
        if ((rc = fit_impl(g, maxtries, 0, 0))) return rc;
        if ((rc = ensure_out(g))) return rc;
        if ((rc = run_predict(g, include_noise))) return rc;
    }
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * g->M * g->P, hipMemcpyDeviceToHost, g->s));
    if (var) HIPCHK(hipMemcpyAsync(var, g->dVar, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    return 0;
}

// Triangular solves as products with inverted diagonal panels, and Ky^-1 (dtrtri + dpotri re-expressed on the NT GEMM).
// Reference: dtrtrs (posterior.py:294), dpotrs (exact_gaussian_inference.py:60), dpotri (linalg.py:127-145).
#include "api_internal.h"

// ---- inverted diagonal panels ---------------------------------------------------------------------
// invP_J = L_JJ^-1 for every panel J of W tiles (PB = W*128 rows), so that every triangular solve
// against L -- candidates (dtrtrs, posterior.py:294), alpha (dpotrs, exact_gaussian_inference.py:60),
// Ky^-1 (dpotri, linalg.py:127-145) -- is ONE product per panel on the MFMA GEMM instead of a chain of
// W dependent 128-column steps.  Built batched over all panels at once: the solve of the identity
// against L_JJ (2W-1 small launches, each covering every panel) gives L_JJ^-T, then one transpose.
int ensure_panel_inv(gp_ctx *g) {
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = std::min(g->panel_tiles, nt);
    // (the width actually built is min(panel_tiles, nt): compared with panel_tiles itself, a matrix of fewer tiles than one
    // panel -- N <= 640 by default, the size of most BO loops -- rebuilt its inverted panel on EVERY predict call: round 4 finding)
    if (g->invp_valid && g->invp_W == W) return 0;
    const long PB = (long)W * GP_TILE;
    const int nJ = (nt + W - 1) / W, nF = nt / W, Wl = nt % W;
    int rc;
    if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJ * PB * PB))) return rc;
    if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJ * PB * PB))) return rc;
    double *Wk = g->dInvPw;
    hipStream_t s = g->s;
    launch_set_identity_blocks(s, Wk, PB, nJ);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: the nF full panels as one batch; pass 1: the ragged last panel (Wl tiles)
        const int batch = pass == 0 ? nF : (Wl ? 1 : 0), Wp = pass == 0 ? W : Wl;
        if (batch == 0) continue;
        const long z0 = pass == 0 ? 0 : nF;
        double *Wb = Wk + z0 * PB * PB;
        const double *Lb = g->dA + z0 * (PB * lda + PB);
        const double *Ib = g->dInvL + z0 * (long)W * GP_TILE * GP_TILE;
        for (int b = 0; b < Wp; ++b) {
            GemmOpt o;
            o.batch = batch;
            o.inplace = 1;
            o.sC = o.sA = PB * PB;
            o.sB = (long)W * GP_TILE * GP_TILE;
            gemm(g, s, 0, Wb, PB, Wb + (long)b * GP_TILE, PB, Ib + (long)b * GP_TILE * GP_TILE, GP_TILE, 0, GP_TILE,
                 TileSet{0, b + 1, b, b + 1, 0}, o);
            if (b + 1 < Wp) {
                o.inplace = 0;
                o.sB = PB * lda + PB;
                gemm(g, s, 1, Wb, PB, Wb + (long)b * GP_TILE, PB, Lb + (long)b * GP_TILE, lda, 1, GP_TILE,
                     TileSet{0, b + 1, b + 1, Wp, 0}, o);
            }
        }
    }
    launch_transpose_blocks(s, g->dInvP, Wk, PB, nJ);
    g->invp_W = W;
    g->invp_valid = true;
    return 0;
}

// Row solve  S = T L^-T  for `mt` row tiles of T (row-major, ld = Npad); T is consumed as the running
// right-hand side.  trapezoid = 1: T is block upper-triangular (row tile r is zero left of column tile r:
// the identity, for L^-T), so panel J only touches the row tiles above its end.
void solve_rows(gp_ctx *g, double *T, double *S, int mt, int trapezoid, int J_from) {
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    const long PB = (long)W * GP_TILE;
    const double *L = g->dA;
    hipStream_t s = g->s;
    auto panel_solve = [&](int J, int J0, int J1, int rows) {
        GemmOpt o;
        o.k_end_tri = 1;
        o.b_sub = J0;
        // S[:, J] = T[:, J] invP_J^T   (invP_J lower triangular: column tile c contracts k <= c)
        gemm(g, s, 0, S, Npad, T + (long)J0 * GP_TILE, Npad, g->dInvP + (long)J * PB * PB, PB, 1, (J1 - J0) * GP_TILE,
             TileSet{0, rows, J0, J1, 0}, o);
    };
    for (int J0 = J_from * W, J = J_from; J0 < nt;) {
        const int J1 = std::min(J0 + W, nt), J2 = std::min(J1 + W, nt);
        const int Kp = (J1 - J0) * GP_TILE;
        const int rows = trapezoid ? std::min(mt, J1) : mt;
        panel_solve(J, J0, J1, rows);
        if (J1 >= nt) break;
        // Two panels per update (full row sets only): panel J+1's columns take panel J's update as a small launch of
        // their own, then ONE launch contracts both panels (K = 2 PB) into everything right of them -- half the round
        // trips of the running right-hand side through HBM and a contraction twice as long.  The accumulator sees the
        // same products in the same order as with one launch per panel: bitwise the same result.
        const bool two = g->pair_panels && !trapezoid && J2 > J1 && J2 < nt;
        if (!two) {
            // T[:, > J] -= S[:, J] L[> J, J]^T
            gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, Kp,
                 TileSet{0, rows, J1, nt, 0});
            J0 = J1;
            ++J;
            continue;
        }
        gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, Kp, TileSet{0, rows, J1, J2, 0});
        panel_solve(J + 1, J1, J2, rows);
        gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, (J2 - J0) * GP_TILE,
             TileSet{0, rows, J2, nt, 0});
        J0 = J2;
        J += 2;
    }
}

// ---- Ky^-1 (potri-equivalent): dtrtri + dlauum re-expressed on the NT GEMM -------------------------
// W = L^-T is the candidate solve applied to the identity (row c of W = (L^-1 e_c)^T); rows above
// the current panel are still zero, so the tile sets are trapezoids and the cost is N^3/3.
// Ky^-1 = W W^T with the contraction of tile row a starting at column a*128: another N^3/3.
// Reference: pdinv / dpotri (GPy/GPy/util/linalg.py:127-145,193-214), Posterior.woodbury_inv
// (posterior.py:176-196).
// Ky^-1 from dT2 = L^-T (block upper triangular) into dWi
int wi_lauum(gp_ctx *g) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    hipStream_t s = g->s;
    int ph;
    ph = phase_begin(g, "potri_lauum", (double)g->N * g->N * g->N / 3.0, 0.0);
    if (g->lauum_panels) {
        // Ky^-1 = (L^-T)(L^-T)^T accumulated k-panel by k-panel: panel p (W tiles of k) adds to the tiles (i, c), c <= i,
        // with i below the panel's end.  Every tile of a launch then walks the SAME k range, so the workgroups of an
        // XCD share their operand panels in L2 like the trailing updates do; as one launch over k = i*128 .. N each
        // tile streams its own up-to-33 MB row panels at its own offset and the product runs at the fabric's pace
        // (47 TFLOP/s at N = 32768).  The accumulator holds -Ky^-1 (C -= A B^T is the kernel's update form).
        HIPCHK(hipMemsetAsync(g->dWi, 0, sizeof(double) * Npad * Npad, s));
        const int W = g->panel_tiles;
        for (int k0 = 0; k0 < nt; k0 += W) {
            const int k1 = std::min(k0 + W, nt);
            GemmOpt o;
            o.k_tri = 1;
            o.k_sub = k0;
            gemm(g, s, 1, g->dWi, Npad, g->dT2 + (long)k0 * GP_TILE, Npad, g->dT2 + (long)k0 * GP_TILE, Npad, 1,
                 (k1 - k0) * GP_TILE, TileSet{0, k1, 0, k1, 1}, o);
        }
        launch_symmetrize_scale(s, g->dWi, Npad, Npad, -1.0);
    } else {
        GemmOpt o;
        o.k_tri = 1;
        gemm(g, s, 0, g->dWi, Npad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, nt, 0, nt, 1}, o);
        launch_symmetrize(s, g->dWi, Npad, Npad);
    }
    phase_end(g, ph);
    return 0;
}

int ensure_wi(gp_ctx *g) {
    if (g->wi_valid) return 0;
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    int rc;
    if ((rc = ensure_panel_inv(g))) return rc;
    if ((rc = dev_realloc(&g->dT, &g->capT, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dT2, &g->capT2, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dWi, &g->capWi, Npad * Npad))) return rc;
    double *T = g->dT;
    hipStream_t s = g->s;
    if (g->emulate_fp64 && g->emulate_fit && g->invp_W % 2 == 0) {
        rc = wi_rns(g);
        if (rc == 0) {
            g->wi_valid = true;
            g->predicted = false;
            return 0;
        }
        if (rc != GP_ERR_RANGE) return rc;
        ++g->emu_fallbacks;   // non-finite factor: the true-fp64 path below returns what the reference would
        g->nphases = 0;
    }
    if (g->li_valid && !g->w_in_t2) {
        // the inverse factor of this fit is at hand (ensure_linv, the fused one-row path): L^-T is its transpose, N^2 traffic
        // instead of the N^3 / 3 solve
        int ph = phase_begin(g, "potri_transpose", 0.0, 16.0 * (double)g->N * g->N / 2);
        launch_transpose_tri(s, g->dT2, g->dLi, Npad, 1);
        phase_end(g, ph);
    } else if (!g->w_in_t2) {
        int ph = phase_begin(g, "potri_solve", (double)g->N * g->N * g->N / 3.0, 0.0);
        launch_set_identity(s, T, Npad, Npad);
        solve_rows(g, T, g->dT2, nt, 1);  // dT2 = L^-T (block upper triangular)
        phase_end(g, ph);
    }
    g->w_in_t2 = true;
    if ((rc = wi_lauum(g))) return rc;
    g->wi_valid = true;
    g->predicted = false;  // dT was reused
    return 0;
}

// ---- the explicit inverse factor Li = L^-1 (dtrtri, linalg.py:217-227) for the fused one-row path (onerow.hip) -------------------
// The same solve of the identity as Ky^-1 starts with, kept: lower triangular, row-major, exact zeros above the diagonal.  Half of
// the potri-equivalent's work (no W W^T product) -- all the acquisition optimiser's gradient calls need between two fits.
int ensure_linv(gp_ctx *g) {
    if (g->li_valid) return 0;
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    int rc;
    if ((rc = ensure_panel_inv(g))) return rc;
    if ((rc = dev_realloc(&g->dT, &g->capT, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dT2, &g->capT2, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dLi, &g->capLi, Npad * Npad))) return rc;
    if (!g->w_in_t2) {
        int ph = phase_begin(g, "potri_solve", (double)g->N * g->N * g->N / 3.0, 0.0);
        launch_set_identity(g->s, g->dT, Npad, Npad);
        solve_rows(g, g->dT, g->dT2, nt, 1);  // dT2 = L^-T (true fp64 in either arithmetic mode)
        phase_end(g, ph);
        g->w_in_t2 = true;
        g->predicted = false;  // dT was reused
    }
    launch_transpose_tri(g->s, g->dLi, g->dT2, Npad, 0);
    g->li_valid = true;
    return 0;
}

extern "C" int gp_get_woodbury_inv(gp_t *g, double *Wi) {
    if (!g || !Wi) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_wi(g))) return rc;
    GP_SYNC(g->s);
    HIPCHK(hipMemcpy2D(Wi, sizeof(double) * g->N, g->dWi, sizeof(double) * g->Npad, sizeof(double) * g->N, g->N,
                       hipMemcpyDeviceToHost));
    return 0;
}

// A handful of candidates at a time BY SUBSTITUTION against L: the route of gp_set_candidates + gp_predict / gp_predict_grad /
// gp_acq_grad with M <= "small_m" resident rows, and of the one-call entry points (gp_*_rows, api_rows.hip) while the inverse factor
// of the fused path (onerow.hip) does not exist yet -- the first N / 768 calls after a fit above N = 4096.
//
// scipy's L-BFGS-B asks for ONE row per call (GPyOpt/GPyOpt/optimization/optimizer.py:28-61 -> models/gpmodel.py:131-142 ->
// GPy/GPy/core/gp.py:407-454, posterior.py:273-302), hundreds of times per BO iteration.  Through the tile path such a call pads
// the row to a 128-row tile and walks ~45 dependent GEMM launches whose workgroups each contract K = 768 ... 1536 for a single tile:
// 3.2 ms at N = 16384 for 34 MFLOP of useful work, all of it launch and pipeline latency.  With M <= GP_SMALL_M rows the solve is
// a matrix-VECTOR problem bound by reading L once (N^2 / 2 doubles: 1.07 GB = 0.27 ms at N = 16384), and it runs as such:
//
//   forward substitution  w = L^-1 k*  by panels, with the inverted diagonal panels invP_J the fit already built:
//       w_J = invP_J t_J ;   t_r -= L[r, J] . w_J   for every row r below the panel
//   (dtrtrs, posterior.py:294) and, for the gradients, beta = Ky^-1 k* -- as row dots with the symmetric Ky^-1 when it exists
//   (gp.py:451-452), else by the backward substitution of the alpha solve (api_grad.hip, run_predict_grad).
//
// Both are ROW DOTS of a row-major matrix with up to four vectors at once -- one kernel, rowdot_kernel: a wave takes one matrix
// row at a time, lane l the elements 2 l, 2 l + 1 (+ 128 q) of the chunk (1 KB per wave instruction, fully coalesced), the
// vectors' matching elements sit in registers for the whole launch, and a fixed-order wave reduction finishes each
// (row, vector) sum: no atomics, bitwise reproducible.  M > 4 runs in passes of four vectors.
#include "gphip_internal.h"

#define RD_CHUNK 768              // k elements per chunk = 64 lanes x 2 x 6
#define RD_ROWS_PER_WAVE 4
#define RD_WAVES 4
#define RD_ROWS (RD_ROWS_PER_WAVE * RD_WAVES)   // rows per workgroup

__device__ __forceinline__ double rd_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// out[m][row0 + r] (MODE 0: =, MODE 1: -=) sum_{k < K} Mat[(row0 + r) ldm + k] v[m][k]   for r < nrows, m < M (<= 4)
// v[m] = V + m ldv (K elements each); out[m] = O + m ldo.  TRI: row r only contracts k < min(K, round_up(r + 1, 128)) -- the
// matrix is lower triangular with zeros above the diagonal (an inverted panel), so the rest of the row is skipped.
template <int MODE, bool TRI>
__global__ __launch_bounds__(64 * RD_WAVES) void rowdot_kernel(const double *Mat, long ldm, long row0, long nrows, int K,
                                                               const double *V, long ldv, int M, double *O, long ldo) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long rbase = (long)blockIdx.x * RD_ROWS + wave * RD_ROWS_PER_WAVE;
    double acc[RD_ROWS_PER_WAVE][4];
#pragma unroll
    for (int r = 0; r < RD_ROWS_PER_WAVE; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[r][m] = 0.0;
    for (int kc = 0; kc < K; kc += RD_CHUNK) {
        const int kn = min(RD_CHUNK, K - kc);                  // a multiple of 128 (panels and Npad are)
        // this lane's elements of the vectors for the chunk
        double2_t vv[4][6];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int k = 2 * lane + 128 * q;
                vv[m][q] = (m < M && k < kn) ? *(const double2_t *)(V + (long)m * ldv + kc + k) : (double2_t){0.0, 0.0};
            }
#pragma unroll
        for (int r = 0; r < RD_ROWS_PER_WAVE; ++r) {
            const long row = rbase + r;
            if (row >= nrows) break;
            int klim = kn;
            if (TRI) klim = min(kn, (int)((row / GP_TILE + 1) * GP_TILE) - kc);
            const double *mp = Mat + (row0 + row) * ldm + kc + 2 * lane;
            double2_t x[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) x[q] = (128 * q < klim) ? *(const double2_t *)(mp + 128 * q) : (double2_t){0.0, 0.0};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    s0 = fma(x[q][0], vv[m][q][0], s0);
                    s1 = fma(x[q][1], vv[m][q][1], s1);
                }
                acc[r][m] += s0 + s1;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RD_ROWS_PER_WAVE; ++r) {
        const long row = rbase + r;
        if (row >= nrows) break;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const double s = rd_wave_sum(acc[r][m]);
            if (lane == 0 && m < M) {
                double *o = O + (long)m * ldo + row0 + row;
                *o = MODE == 1 ? *o - s : s;
            }
        }
    }
}

template <int MODE, bool TRI>
static void launch_rowdot(hipStream_t s, const double *Mat, long ldm, long row0, long nrows, int K, const double *V, long ldv,
                          int M, double *O, long ldo) {
    if (nrows <= 0 || K <= 0 || M <= 0) return;
    GP_LAUNCH((rowdot_kernel<MODE, TRI>), dim3((unsigned)((nrows + RD_ROWS - 1) / RD_ROWS)), dim3(64 * RD_WAVES), 0, s, Mat, ldm,
              row0, nrows, K, V, ldv, M, O, ldo);
}

// S[m, :] = T[m, :] L^-T  for m < M: T (M x ldt) holds K(Xs, X) and is consumed as the running right-hand side.
void launch_small_forward_solve(hipStream_t s, const double *L, long lda, const double *invP, int W, long Npad, double *T,
                                double *S, long ldt, int M) {
    const int nt = (int)(Npad / GP_TILE);
    const long PB = (long)W * GP_TILE;
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mc = std::min(4, M - m0);
        double *Tm = T + (long)m0 * ldt, *Sm = S + (long)m0 * ldt;
        for (int J0 = 0, J = 0; J0 < nt; J0 += W, ++J) {
            const int J1 = std::min(J0 + W, nt);
            const int Kp = (J1 - J0) * GP_TILE;
            const long off = (long)J0 * GP_TILE;
            // w_J = invP_J t_J   (invP_J: Kp x Kp lower triangular, row-major with pitch PB)
            launch_rowdot<0, true>(s, invP + (long)J * PB * PB, PB, 0, Kp, Kp, Tm + off, ldt, mc, Sm + off, ldt);
            // t_r -= L[r, J] . w_J for the rows below the panel
            if (J1 < nt)
                launch_rowdot<1, false>(s, L + off, lda, (long)J1 * GP_TILE, Npad - (long)J1 * GP_TILE, Kp, Sm + off, ldt, mc,
                                        Tm, ldt);
        }
    }
}

// beta[m, :] = k*[m, :] Ky^-1 for m < M (Ky^-1 symmetric: row j of Wi dotted with k*)
void launch_small_wi_product(hipStream_t s, const double *Wi, long ldw, long Npad, const double *Kx, long ldk, int M,
                             double *beta, long ldb) {
    for (int m0 = 0; m0 < M; m0 += 4) {
        const int mc = std::min(4, M - m0);
        launch_rowdot<0, false>(s, Wi, ldw, 0, Npad, (int)Npad, Kx + (long)m0 * ldk, ldk, mc, beta + (long)m0 * ldb, ldb);
    }
}

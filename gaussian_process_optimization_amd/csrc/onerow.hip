// The acquisition optimiser's one-row calls as THREE launches over the explicit inverse factor.
//
// scipy's L-BFGS-B evaluates acquisition_function_withGradients one location at a time, hundreds of times between two fits
// (GPyOpt/GPyOpt/optimization/optimizer.py:36-61 -> acquisitions/base.py:42-50 -> models/gpmodel.py:131-142 ->
// GPy/GPy/core/gp.py:407-454 + posterior.py:273-302).  Per location x the reference needs
//     k* = K(X, x),  mean = k*^T alpha,  w = L^-1 k* (dtrtrs),  var = kss - |w|^2,  beta = Ky^-1 k*,
//     d mean / dx = gradients_X(alpha^T, x, X),  d var / dx = gradients_X(-2 beta^T, x, X)       (stationary.py:336-364)
// and the EI / LCB / MPI (+ local penalisation) chain rule on top.  With the inverse factor Li = L^-1 kept from the potri-
// equivalent (api_solve.hip, ensure_linv; lower triangular, row-major) this is matrix-vector work bound by reading the lower
// triangle of Li twice (2 x 8 N^2 / 2 bytes: 2.15 GB at N = 16384):
//
//   rows_forward_kernel    w = Li k*           row dots; k* generated on the fly per 1024-column chunk (never stored)
//   rows_backward_kernel   beta = Li^T w       column sums over the same tiles; |w|^2 per row block on the way
//   rows_finish_kernel     beta -> the two gradients_X sums over the training points, then -- in the last workgroup to arrive --
//                          mean, variance, acquisition, penaliser, written straight into the caller's pinned result block
//
// x travels in the kernel arguments and the results land in host-visible memory: no copy commands either side of the launches.
// The smallm.hip route walked ~22 panels x 2 dependent launches for the same substitution (1.36 ms per gradient call at
// N = 16384, 0.18 ms at N = 512).  A posterior-only call (no gradient) is forward + finish: two launches, one read of Li.
//
// Tiling: row blocks of 128 rows x chunks of 1024 columns of the lower triangle (row block R has R / 8 + 1 chunks, the last
// one cut at the diagonal tile's end).  Every partial sum has ONE writer and is reduced in a fixed order: bitwise repeatable.
#include "gphip_internal.h"
#include "acq_math.h"

#define RW_CW 1024   // columns per tile
#define RW_Q 8       // 128-column groups per tile: a lane holds the column pair 2 lane + 128 q

__host__ __device__ static inline int rw_nch(int R) { return R / 8 + 1; }
long rows_tiles(int nt) {
    long t = 0;
    for (int R = 0; R < nt; ++R) t += rw_nch(R);
    return t;
}
// tile index -> (row block, chunk): groups of 8 row blocks with g + 1 chunks each; 4 g (g + 1) tiles lie before group g
__device__ __forceinline__ void rw_decode(int idx, int &R, int &C) {
    int g = (int)(0.5f * sqrtf((float)idx));
    while (4 * (g + 1) * (g + 2) <= idx) ++g;
    while (g > 0 && 4 * g * (g + 1) > idx) --g;
    const int rem = idx - 4 * g * (g + 1);
    R = 8 * g + rem / (g + 1);
    C = rem % (g + 1);
}
__device__ __forceinline__ double rw_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
// sum over the workgroup's 256 threads in a fixed order; every thread gets the result.  sh: 4 doubles.
__device__ __forceinline__ double rw_block_sum(double v, double *sh) {
    v = rw_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// one 16-byte load of the inverse factor; NT: marked non-temporal (the factor is read once per pass and is far larger than the caches)
template <bool NT>
__device__ __forceinline__ double2_t rw_load2(const double *p) {
    if (NT) return __builtin_nontemporal_load((const double2_t *)p);
    return *(const double2_t *)p;
}

// sum of p[i stride] for i = first, first + step, ... < end, added in that order; the loads go out eight at a time (one load
// after the other, each waiting for its predecessor's add, cost ~1 us apiece: 60 us for 64 partials)
__device__ __forceinline__ double rw_strided_sum(const double *p, long stride, int first, int step, int end) {
    double s = 0.0;
    for (int i = first; i < end; i += 8 * step) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = i + u * step;
            v[u] = k < end ? p[(long)k * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    return s;
}

// ---- forward: wpart[C][m][r] = sum_{k in chunk C} Li[r, k] k*_m[k];  meanpart[C][m] = sum_{k in chunk C} k*_m[k] alpha[k] -----------
// RB rows of the tile per workgroup (128, or 32 for matrices of a few tiles: four times the workgroups, a quarter of the
// dependent load -> reduce steps in each -- the launch is latency-bound there, not bandwidth-bound)
template <int MV, int RB, bool NT>
__global__ __launch_bounds__(256) void rows_forward_kernel(const double *Li, long ld, RowsX rx, KernParams kp, const double *X, long N,
                                                           const double *alpha, double *wpart, long Npad, int nt,
                                                           double *meanpart) {
    __shared__ __attribute__((aligned(16))) double ks[MV][RW_CW];
    __shared__ double xs_s[ROWS_MAX_XS];
    __shared__ double red[4];
    constexpr int SUB = GP_TILE / RB;
    int R, C;
    rw_decode((int)blockIdx.x / SUB, R, C);
    const int sub = (int)blockIdx.x % SUB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = kp.D, M = rx.M;
    const long c0 = (long)C * RW_CW;
    const int klim = min(RW_CW, (R + 1) * GP_TILE - (int)c0);   // a multiple of 128
    for (int i = tid; i < M * D; i += 256) xs_s[i] = rx.xs[i] / kp_div(kp, i % D);
    __syncthreads();
    // k*_m[c0 + j]: the arithmetic of cross_k_rows_kernel (inputs divided first, squares summed in dimension order): the same bits
    for (int j = tid; j < RW_CW; j += 256) {
        const long i = c0 + j;
        double acc[MV];
#pragma unroll
        for (int m = 0; m < MV; ++m) acc[m] = kp.gower ? 1.0 : 0.0;
        const bool live = j < klim && i < N;
        if (live) {
            for (int d = 0; d < D; ++d) {
                const double b = X[i * D + d] / kp_div(kp, d);
#pragma unroll
                for (int m = 0; m < MV; ++m) {
                    if (m < M) {
                        const double df = xs_s[m * D + d] - b;
                        if (kp.gower) {
                            const double r = kp.gdisc[d] ? (df != 0.0 ? 1.0 : 0.0) : fabs(df);
                            acc[m] *= gp_k_of_r2(kp.kernel, kp.variance, r * r);
                        } else {
                            acc[m] = fma(df, df, acc[m]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MV; ++m)
            ks[m][j] = (live && m < M) ? (kp.gower ? acc[m] : gp_k_of_r2(kp.kernel, kp.variance, acc[m])) : 0.0;
    }
    __syncthreads();
    if (R == nt - 1 && sub == 0) {   // the last row block meets every chunk: it carries the mean's partial sums
#pragma unroll
        for (int m = 0; m < MV; ++m) {
            double s = 0.0;
            for (int j = tid; j < klim; j += 256)
                if (c0 + j < N) s = fma(ks[m][j], alpha[c0 + j], s);
            s = rw_block_sum(s, red);
            if (tid == 0 && m < M) meanpart[C * MV + m] = s;
        }
    }
    double2_t vv[MV][RW_Q];
#pragma unroll
    for (int m = 0; m < MV; ++m)
#pragma unroll
        for (int q = 0; q < RW_Q; ++q) vv[m][q] = *(const double2_t *)&ks[m][2 * lane + 128 * q];
    const long rbase = (long)R * GP_TILE + sub * RB + wave * (RB / 4);
    const double2_t zero2 = {0.0, 0.0};
    for (int r = 0; r < RB / 4; r += 2) {
        const double *p0 = Li + (rbase + r) * ld + c0 + 2 * lane;
        const double *p1 = p0 + ld;
        double2_t x0[RW_Q], x1[RW_Q];
#pragma unroll
        for (int q = 0; q < RW_Q; ++q) {
            x0[q] = (128 * q < klim) ? rw_load2<NT>(p0 + 128 * q) : zero2;
            x1[q] = (128 * q < klim) ? rw_load2<NT>(p1 + 128 * q) : zero2;
        }
#pragma unroll
        for (int m = 0; m < MV; ++m) {
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int q = 0; q < RW_Q; ++q) {
                a0 = fma(x0[q][0], vv[m][q][0], a0);
                a1 = fma(x0[q][1], vv[m][q][1], a1);
                b0 = fma(x1[q][0], vv[m][q][0], b0);
                b1 = fma(x1[q][1], vv[m][q][1], b1);
            }
            const double sa = rw_wave_sum(a0 + a1), sb = rw_wave_sum(b0 + b1);
            if (lane == 0 && m < M) {
                double *o = wpart + ((long)C * MV + m) * Npad + rbase + r;
                o[0] = sa;
                o[1] = sb;
            }
        }
    }
}

// ---- backward: bpart[R][m][k] = sum_{r in block R} Li[r, k] w_m[r];  vpart[R][m] = sum_{r in block R} w_m[r]^2 -------------------------
template <int MV, int RB, bool NT>
__global__ __launch_bounds__(256) void rows_backward_kernel(const double *Li, long ld, const double *wpart, long Npad, int M,
                                                            double *bpart, double *vpart) {
    __shared__ double wv[MV][RB];
    __shared__ double red[4];
    constexpr int SUB = GP_TILE / RB;
    int R, C;
    rw_decode((int)blockIdx.x / SUB, R, C);
    const int sub = (int)blockIdx.x % SUB;
    const long row0 = (long)R * GP_TILE + sub * RB;      // first of this workgroup's RB rows
    const int rb = R * SUB + sub;                         // its index among the row blocks of height RB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long c0 = (long)C * RW_CW;
    const int klim = min(RW_CW, (R + 1) * GP_TILE - (int)c0);
    const int nch = rw_nch(R);
    for (int i = tid; i < MV * RB; i += 256) {
        const int m = i / RB, r = i % RB;
        wv[m][r] = m < M ? rw_strided_sum(wpart + (long)m * Npad + row0 + r, (long)MV * Npad, 0, 1, nch) : 0.0;
    }
    __syncthreads();
    if (C == 0) {
#pragma unroll
        for (int m = 0; m < MV; ++m) {
            const double w = tid < RB ? wv[m][tid] : 0.0;
            const double s = rw_block_sum(w * w, red);
            if (tid == 0 && m < M) vpart[rb * MV + m] = s;
        }
    }
    // wave w owns the column pairs 2 lane + 128 (2 w + j), j = 0, 1, over all RB rows of the block
    const int q0 = 2 * wave;
    if (128 * q0 >= klim) return;
    const bool two = 128 * (q0 + 1) < klim;
    double2_t acc[MV][2];
#pragma unroll
    for (int m = 0; m < MV; ++m) acc[m][0] = acc[m][1] = (double2_t){0.0, 0.0};
    const double *base = Li + row0 * ld + c0 + 2 * lane + 128 * q0;
    const double2_t zero2 = {0.0, 0.0};
    for (int r = 0; r < RB; r += 8) {
        double2_t x[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *p = base + (long)(r + u) * ld;
            x[u][0] = rw_load2<NT>(p);
            x[u][1] = two ? rw_load2<NT>(p + 128) : zero2;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int m = 0; m < MV; ++m) {
                const double w = wv[m][r + u];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[m][j][0] = fma(x[u][j][0], w, acc[m][j][0]);
                    acc[m][j][1] = fma(x[u][j][1], w, acc[m][j][1]);
                }
            }
    }
#pragma unroll
    for (int m = 0; m < MV; ++m) {
        if (m >= M) break;
        double *o = bpart + ((long)rb * MV + m) * Npad + c0 + 2 * lane + 128 * q0;
        *(double2_t *)o = acc[m][0];
        if (two) *(double2_t *)(o + 128) = acc[m][1];
    }
}

// ---- finish --------------------------------------------------------------------------------------------------------------------------
// grid = ceil(N / 64) workgroups of 256 threads; a workgroup carries 64 training points n.  Its four waves first split the
// row-block partials of those points between them -- beta_m[n] = sum_{R >= n / rbh} bpart[R][m][n] (gradient call), or
// w_m[n] = sum_C wpart[C][m][n] for |w|^2 (value call) -- wave v takes every fourth partial, the four sums are added in wave
// order.  Wave 0 then takes point n through the two gradients_X sums (the geometry of predict_grad_kernel in grad.hip: Euclidean
// scaled differences on the kernel's own lengthscale -- under Gower too, as the fork does, stationary.py:336-364).  Per-workgroup
// sums go to gpart, and the LAST workgroup to arrive (device-scope counter, counted from this pass's base: it is never reset, so a
// pass cannot inherit a stale count) reduces them in workgroup order -- eight interleaved slices, added in slice order -- and
// writes the results, then the pass's ticket behind them.
// gpart row (per workgroup): [2 M D gradient sums | M sums of w^2], RW_GROW doubles apart
// out (host-visible): [mean MV][var MV][acq MV][dmdx MV D][dvdx MV D][dacq MV D]
#define RW_GROW (2 * ROWS_MAX_XS + ROWS_MAX_M)
template <int MV>
__global__ __launch_bounds__(256) void rows_finish_kernel(RowsX rx, KernParams kp, const double *X, long N, const double *alpha,
                                                          const double *wpart, const double *bpart, const double *meanpart,
                                                          const double *vpart, long Npad, int nt, int rbh, int want_grad,
                                                          double kss, double noise_add, RowsAcq aq, double *gpart,
                                                          unsigned int *counter, unsigned int counter_base, double *out,
                                                          double ticket) {
    __shared__ double xs_s[ROWS_MAX_XS], xraw_s[ROWS_MAX_XS];
    __shared__ double part_s[4][MV][64];
    __shared__ double fin_s[8][RW_GROW + 2 * ROWS_MAX_M];
    __shared__ double res_s[RW_GROW + 2 * ROWS_MAX_M];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = kp.D, M = rx.M;
    const int nval = want_grad ? 2 * M * D : 0;       // gradient sums per workgroup; the M sums of w^2 follow (value call)
    const int nsum = want_grad ? nval : M;
    const int off = want_grad ? 0 : 2 * ROWS_MAX_XS;
    for (int i = tid; i < M * D; i += 256) {
        xraw_s[i] = rx.xs[i];
        xs_s[i] = rx.xs[i] / kp.ls[i % D];
    }
    const long n = (long)blockIdx.x * 64 + lane;
    const bool live = n < N;
    const int Rn = (int)(n / GP_TILE);
    const int nrb = nt * (GP_TILE / rbh);     // row blocks of the backward pass (height rbh)
    // this wave's share of the partial sums of point n
    for (int m = 0; m < M; ++m) {
        double s = 0.0;
        if (live)
            s = want_grad ? rw_strided_sum(bpart + (long)m * Npad + n, (long)MV * Npad, (int)(n / rbh) + wave, 4, nrb)
                          : rw_strided_sum(wpart + (long)m * Npad + n, (long)MV * Npad, wave, 4, rw_nch(Rn));
        part_s[wave][m][lane] = s;
    }
    __syncthreads();
    if (wave == 0) {
        for (int m = 0; m < M; ++m) {
            const double b = ((part_s[0][m][lane] + part_s[1][m][lane]) + part_s[2][m][lane]) + part_s[3][m][lane];
            if (!want_grad) {
                const double v = rw_wave_sum(b * b);
                if (lane == 0) gpart[(long)blockIdx.x * RW_GROW + off + m] = v;
                continue;
            }
            double s = 0.0, a = 0.0;
            if (live) {
                for (int d = 0; d < D; ++d) {
                    const double df = xs_s[m * D + d] - X[n * D + d] / kp.ls[d];
                    s = fma(df, df, s);
                }
                a = alpha[n];
            }
            double kv, gv;
            gp_k_and_g(kp.kernel, kp.variance, s, kv, gv);
            if (s == 0.0) gv = 0.0;   // invdist = 0 where the distance is exactly 0 (stationary.py:251-258)
            const double tm = live ? gv * a : 0.0, tv = live ? gv * (-2.0 * b) : 0.0;
            for (int d = 0; d < D; ++d) {
                const double dq = live ? xs_s[m * D + d] - X[n * D + d] / kp.ls[d] : 0.0;
                const double vm = rw_wave_sum(tm * dq), vvv = rw_wave_sum(tv * dq);
                if (lane == 0) {
                    gpart[(long)blockIdx.x * RW_GROW + m * D + d] = vm;
                    gpart[(long)blockIdx.x * RW_GROW + M * D + m * D + d] = vvv;
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) last_s = (atomicAdd(counter, 1u) - counter_base == gridDim.x - 1) ? 1 : 0;   // (unsigned: wraps with the base)
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    // Every sum over partials -- the per-workgroup sums, the row blocks' |w|^2, the chunks' mean terms -- in eight interleaved
    // slices: slice j adds the partials j, j + 8, ... in order, then the slices are added in order.  (One thread walking 128
    // partials one load after the other took 60 us here.)
    const int nmean = rw_nch(nt - 1);
    const int ntot = nsum + (want_grad ? M : 0) + M;              // [workgroup sums | w^2 per row block (gradient call) | mean]
    for (int idx = tid; idx < 8 * ntot; idx += 256) {
        const int v = idx % ntot, j = idx / ntot;
        double s;
        if (v < nsum)
            s = rw_strided_sum(gpart + off + v, RW_GROW, j, 8, (int)gridDim.x);
        else if (want_grad && v < nsum + M)
            s = rw_strided_sum(vpart + (v - nsum), MV, j, 8, nrb);
        else
            s = rw_strided_sum(meanpart + (v - (ntot - M)), MV, j, 8, nmean);
        fin_s[j][v] = s;
    }
    __syncthreads();
    for (int v = tid; v < ntot; v += 256) {
        double s = 0.0;
        for (int j = 0; j < 8; ++j) s += fin_s[j][v];
        res_s[v] = (want_grad && v < nval) ? s / kp.ls[v % D] : s;   // (x - x') / l^2 = scaled difference / l
    }
    __syncthreads();
    if (tid < M) {
        const int m = tid;
        const double mean = res_s[ntot - M + m];
        const double ssq = res_s[nsum + (want_grad ? m : m - M)];       // gradient call: after the 2 M D sums; value call: the sums themselves
        const double var = (kss - ssq) + noise_add;
        out[m] = mean;
        out[MV + m] = var;
        double *dm = out + 3 * MV + m * D, *dv = out + 3 * MV + MV * D + m * D, *da = out + 3 * MV + 2 * MV * D + m * D;
        if (want_grad)
            for (int d = 0; d < D; ++d) {
                dm[d] = res_s[m * D + d];
                dv[d] = res_s[M * D + m * D + d];
            }
        if (aq.on) {
            double f, c_m, c_s, ds_scale;
            acq_terms(aq.type, aq.par, aq.fmin, aq.y_mean, aq.y_std, mean, var, f, c_m, c_s, ds_scale);
            double neg = -f;
            if (want_grad)
                for (int d = 0; d < D; ++d) da[d] = -(c_m * (dm[d] * aq.y_std) + c_s * (dv[d] * ds_scale));
            if (aq.lp) {
                const double *x = xraw_s + m * D;
                neg = want_grad ? lp_value_grad(neg, da, x, D, aq.Xb, aq.nb, aq.r0, aq.s0, aq.transform)
                                : lp_value(neg, x, D, aq.Xb, aq.nb, aq.r0, aq.s0, aq.transform);
            }
            out[2 * MV + m] = neg;
        }
    }
    __syncthreads();
    if (tid == 0) out[ROWS_OUT_DOUBLES] = ticket;   // this pass is complete (the host checks the ticket behind the results)
}

// ---- the mean's gradient alone ---------------------------------------------------------------------------------------------------------
// d mean / dx = gradients_X(alpha^T, x, X) (gp.py:433-438 over stationary.py:336-364) needs neither L^-1 nor k* products: one pass
// over the training points.  What estimate_L's L-BFGS-B asks for, one location at a time and D + 1 times per step (it differentiates by
// forward differences, batch_local_penalization.py:55-67).  grid = ceil(N / 256); the last workgroup to arrive adds the workgroups'
// sums in workgroup order and writes dmdx [M, D] at out + 3 MV (the place rows_finish_kernel writes it).
template <int MV>
__global__ __launch_bounds__(256) void rows_mean_grad_kernel(RowsX rx, KernParams kp, const double *X, long N, const double *alpha,
                                                             double *gpart, unsigned int *counter, unsigned int counter_base,
                                                             double *out, double ticket) {
    __shared__ double xs_s[ROWS_MAX_XS];
    __shared__ double sh[4][ROWS_MAX_XS];
    __shared__ double fin_s[8][ROWS_MAX_XS];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = kp.D, M = rx.M, nval = M * D;
    for (int i = tid; i < nval; i += 256) xs_s[i] = rx.xs[i] / kp.ls[i % D];
    __syncthreads();
    const long n = (long)blockIdx.x * 256 + tid;
    const bool live = n < N;
    const double a = live ? alpha[n] : 0.0;
    for (int m = 0; m < M; ++m) {
        double s = 0.0;
        if (live)
            for (int d = 0; d < D; ++d) {
                const double df = xs_s[m * D + d] - X[n * D + d] / kp.ls[d];
                s = fma(df, df, s);
            }
        double kv, gv;
        gp_k_and_g(kp.kernel, kp.variance, s, kv, gv);
        if (s == 0.0) gv = 0.0;   // invdist = 0 where the distance is exactly 0 (stationary.py:251-258)
        const double tm = live ? gv * a : 0.0;
        for (int d = 0; d < D; ++d) {
            const double dq = live ? xs_s[m * D + d] - X[n * D + d] / kp.ls[d] : 0.0;
            const double v = rw_wave_sum(tm * dq);
            if (lane == 0) sh[wave][m * D + d] = v;
        }
    }
    __syncthreads();
    if (tid < nval) gpart[(long)blockIdx.x * RW_GROW + tid] = ((sh[0][tid] + sh[1][tid]) + sh[2][tid]) + sh[3][tid];
    __threadfence();
    __syncthreads();
    if (tid == 0) last_s = (atomicAdd(counter, 1u) - counter_base == gridDim.x - 1) ? 1 : 0;   // (unsigned: wraps with the base)
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    for (int idx = tid; idx < 8 * nval; idx += 256) {
        const int v = idx % nval, j = idx / nval;
        fin_s[j][v] = rw_strided_sum(gpart + v, RW_GROW, j, 8, (int)gridDim.x);
    }
    __syncthreads();
    for (int v = tid; v < nval; v += 256) {
        double s = 0.0;
        for (int j = 0; j < 8; ++j) s += fin_s[j][v];
        out[3 * MV + v] = s / kp.ls[v % D];   // (x - x') / l^2 = scaled difference / l
    }
    __syncthreads();
    if (tid == 0) out[ROWS_OUT_DOUBLES] = ticket;
}

void launch_rows_mean_grad(hipStream_t s, const RowsX &rx, const KernParams &kp, const double *X, long N, const double *alpha,
                           const RowsWork &w, double *out) {
    const unsigned grid = rows_mean_grad_grid(N);
    if (rx.M == 1)
        GP_LAUNCH(rows_mean_grad_kernel<1>, dim3(grid), dim3(256), 0, s, rx, kp, X, N, alpha, w.gpart, w.counter, w.counter_base, out, w.ticket);
    else
        GP_LAUNCH(rows_mean_grad_kernel<ROWS_MAX_M>, dim3(grid), dim3(256), 0, s, rx, kp, X, N, alpha, w.gpart, w.counter, w.counter_base, out, w.ticket);
}

// ---- launchers -------------------------------------------------------------------------------------------------------------------------
size_t rows_gpart_elems(long N) { return (size_t)std::max(rows_finish_grid(N), rows_mean_grad_grid(N)) * RW_GROW; }

int rows_block_height(int nt) { return nt <= 16 ? 32 : GP_TILE; }

template <int MV, int RB, bool NT>
static void launch_rows_t(hipStream_t s, const double *Li, long Npad, const RowsX &rx, const KernParams &kp, const double *X, long N,
                          const double *alpha, int want_grad, double kss, double noise_add, const RowsAcq &aq, const RowsWork &w,
                          double *out) {
    const int nt = (int)(Npad / GP_TILE);
    const unsigned tiles = (unsigned)rows_tiles(nt) * (GP_TILE / RB);
    const unsigned fin = rows_finish_grid(N);
    GP_LAUNCH((rows_forward_kernel<MV, RB, NT>), dim3(tiles), dim3(256), 0, s, Li, Npad, rx, kp, X, N, alpha, w.wpart, Npad, nt, w.meanpart);
    if (want_grad)
        GP_LAUNCH((rows_backward_kernel<MV, RB, NT>), dim3(tiles), dim3(256), 0, s, Li, Npad, w.wpart, Npad, rx.M, w.bpart, w.vpart);
    GP_LAUNCH(rows_finish_kernel<MV>, dim3(fin), dim3(256), 0, s, rx, kp, X, N, alpha, w.wpart, w.bpart, w.meanpart, w.vpart, Npad, nt,
              RB, want_grad, kss, noise_add, aq, w.gpart, w.counter, w.counter_base, out, w.ticket);
}

void launch_rows(hipStream_t s, const double *Li, long Npad, const RowsX &rx, const KernParams &kp, const double *X, long N,
                 const double *alpha, int want_grad, double kss, double noise_add, const RowsAcq &aq, const RowsWork &w,
                 double *out, int nt_loads) {
    const bool low = rows_block_height((int)(Npad / GP_TILE)) == 32;
#define RW_GO(MV, RB, NT) launch_rows_t<MV, RB, NT>(s, Li, Npad, rx, kp, X, N, alpha, want_grad, kss, noise_add, aq, w, out)
    if (rx.M == 1) {
        if (low) RW_GO(1, 32, false);
        else if (nt_loads) RW_GO(1, GP_TILE, true);
        else RW_GO(1, GP_TILE, false);
    } else {
        if (low) RW_GO(ROWS_MAX_M, 32, false);
        else if (nt_loads) RW_GO(ROWS_MAX_M, GP_TILE, true);
        else RW_GO(ROWS_MAX_M, GP_TILE, false);
    }
#undef RW_GO
}

// ---- Li = (L^-T)^T: lower triangular, explicit zeros above the diagonal; and back -------------------------------------------------------
// mode 0: dst lower <- transpose of src upper;  mode 1: dst upper <- transpose of src lower.  32 x 32 LDS tiles.
__global__ __launch_bounds__(256) void transpose_tri_kernel(double *dst, const double *src, long n, int mode) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;   // destination tile (by, bx)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool skip = mode == 0 ? bx > by : bx < by;   // wholly on the zero side
    if (!skip)
        for (int r = ty; r < 32; r += 8) t[r][tx] = src[((long)bx * 32 + r) * n + (long)by * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long i = (long)by * 32 + r, j = (long)bx * 32 + tx;
        const bool keep = !skip && (mode == 0 ? j <= i : j >= i);
        dst[i * n + j] = keep ? t[tx][r] : 0.0;
    }
}
void launch_transpose_tri(hipStream_t s, double *dst, const double *src, long n, int mode) {
    dim3 grid((unsigned)(n / 32), (unsigned)(n / 32));
    GP_LAUNCH(transpose_tri_kernel, grid, dim3(256), 0, s, dst, src, n, mode);
}

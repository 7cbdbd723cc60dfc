// Gradient-side kernels: hyper-parameter gradients of the LML, predictive gradients,
// acquisition gradients, and the small helpers of the potri-equivalent (identity, symmetrise).
//
// Reference: Stationary.update_gradients_full (GPy/GPy/kern/src/stationary.py:218-238),
// _inv_dist (:251-258), _lengthscale_grads_pure (:260-261) and its native twin
// stationary_utils.c:34-48 (_lengthscale_grads), RBF.dK_dr (rbf.py:53-54), Matern52.dK_dr
// (stationary.py:578-579), Gaussian.exact_inference_gradients (gaussian.py:78-79),
// dL_dK = 0.5 (alpha alpha^T - P Ky^-1) (exact_gaussian_inference.py:70);
// Stationary.gradients_X (stationary.py:336-364) / stationary_utils.c:1-14 (_grad_X) as used by
// GP.predictive_gradients (gp.py:407-454); acquisition gradients EI.py:42-51, LCB.py:39-46,
// MPI.py:42-51 and GPModel.predict_withGradients (gpmodel.py:131-142).
//
// The reference makes D+3 full N x N passes with N x N temporaries (K, dL_dr, tmp, one per ARD
// dimension).  Here ONE pass over the lower tiles regenerates r and K from X (staged in LDS),
// reads Ky^-1 once, and produces all D+2 sums; per-tile partials are reduced in fixed order.
#include "gphip_internal.h"
#include "../../include/gphip.h"
#include "acq_math.h"

#define GCH 16  // ARD dimensions handled per pass (accumulators stay in registers)

__device__ __forceinline__ double wave_sum_g(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

#define k_and_g gp_k_and_g   // gphip_internal.h

// ---- identity / symmetrise ----------------------------------------------------------------------
__global__ void set_identity_kernel(double *T, long ld, long n) {
    const long j2 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const long i = blockIdx.y;
    if (j2 >= n) return;
    double2_t v;
    v[0] = (i == j2) ? 1.0 : 0.0;
    v[1] = (i == j2 + 1) ? 1.0 : 0.0;
    *(double2_t *)(T + i * ld + j2) = v;
}
void launch_set_identity(hipStream_t s, double *T, long ld, long n) {
    dim3 grid((unsigned)((n / 2 + 255) / 256), (unsigned)n);
    GP_LAUNCH(set_identity_kernel, grid, dim3(256), 0, s, T, ld, n);
}

// upper <- lower, 32x32 LDS transpose tiles
__global__ void symmetrize_kernel(double *A, long ld, long n) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;  // by >= bx: source tile (by, bx) in the lower part
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long gi = (long)by * 32 + r, gj = (long)bx * 32 + tx;
        t[r][tx] = (gi < n && gj < n) ? A[gi * ld + gj] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long gi = (long)bx * 32 + r, gj = (long)by * 32 + tx;  // destination (bx-tile row, by-tile col)
        if (gi < n && gj < n && gj > gi) A[gi * ld + gj] = t[tx][r];
    }
}
__global__ void symmetrize_scale_kernel(double *A, long ld, long n, double scale) {
    __shared__ double t[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;  // by >= bx: source tile (by, bx) in the lower part
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long gi = (long)by * 32 + r, gj = (long)bx * 32 + tx;
        t[r][tx] = (gi < n && gj < n) ? A[gi * ld + gj] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long gi = (long)by * 32 + r, gj = (long)bx * 32 + tx;
        if (gi < n && gj < n && gj <= gi) A[gi * ld + gj] = scale * t[r][tx];
    }
    for (int r = ty; r < 32; r += 8) {
        const long gi = (long)bx * 32 + r, gj = (long)by * 32 + tx;
        if (gi < n && gj < n && gj > gi) A[gi * ld + gj] = scale * t[tx][r];
    }
}
void launch_symmetrize_scale(hipStream_t s, double *A, long ld, long n, double scale) {
    const unsigned nb = (unsigned)((n + 31) / 32);
    GP_LAUNCH(symmetrize_scale_kernel, dim3(nb, nb), dim3(256), 0, s, A, ld, n, scale);
}
void launch_symmetrize(hipStream_t s, double *A, long ld, long n) {
    const unsigned nb = (unsigned)((n + 31) / 32);
    GP_LAUNCH(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, s, A, ld, n);
}

// ---- LML hyper-gradients: one pass over the lower tiles ------------------------------------------
#define NACC (GCH + 2)
// partial[tile][NACC]: [0] sum K dL_dK (w), [1] sum diag dL_dK, [2+q] sum w g dL_dK dq^2 (ARD) or [2] sum w g dL_dK r^2 (iso)
__global__ __launch_bounds__(256) void lml_grad_tile_kernel(const double *X, long N, KernParams kp, int ard, int d0,
                                                            const double *alpha, long lda_, int P, const double *Wi,
                                                            long ldw, double *partial) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = kp.D;
    double *xi = sm;                        // [D][128]
    double *xj = xi + (long)D * GP_TILE;    // [D][128]
    double *ai = xj + (long)D * GP_TILE;    // [P][128]
    double *aj = ai + (long)P * GP_TILE;    // [P][128]
    // Gower model: the fork's update_gradients_full takes K -- the variance gradient's weight -- through the Gower branch
    // (stationary.py:224 over :116-135) and everything with dK_dr through the Euclidean distance on the kernel's own lengthscale
    // (:227-238): a second staging of the inputs, divided by the variables' ranges
    double *ui = aj + (long)P * GP_TILE;    // [D][128]  (gower only)
    double *uj = ui + (long)D * GP_TILE;    // [D][128]
    __shared__ double red[4][NACC];
    const int tid = threadIdx.x;
    const long t = blockIdx.x;
    int tm = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)tm * (tm + 1) / 2 > t) --tm;
    while ((long)(tm + 1) * (tm + 2) / 2 <= t) ++tm;
    const int tn = (int)(t - (long)tm * (tm + 1) / 2);
    for (int idx = tid; idx < GP_TILE * D; idx += 256) {
        const int r = idx / D, d = idx - r * D;
        const long gi = (long)tm * GP_TILE + r, gj = (long)tn * GP_TILE + r;
        xi[d * GP_TILE + r] = (gi < N) ? X[gi * D + d] / kp.ls[d] : 0.0;
        xj[d * GP_TILE + r] = (gj < N) ? X[gj * D + d] / kp.ls[d] : 0.0;
        if (kp.gower) {
            ui[d * GP_TILE + r] = (gi < N) ? X[gi * D + d] / kp.gdiv[d] : 0.0;
            uj[d * GP_TILE + r] = (gj < N) ? X[gj * D + d] / kp.gdiv[d] : 0.0;
        }
    }
    for (int idx = tid; idx < GP_TILE * P; idx += 256) {
        const int p = idx / GP_TILE, r = idx - p * GP_TILE;
        const long gi = (long)tm * GP_TILE + r, gj = (long)tn * GP_TILE + r;
        ai[idx] = (gi < N) ? alpha[p * lda_ + gi] : 0.0;
        aj[idx] = (gj < N) ? alpha[p * lda_ + gj] : 0.0;
    }
    __syncthreads();

    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    const int cx = (tid & 63) * 2, ry = tid >> 6;
    const long gc = (long)tn * GP_TILE + cx;
    // gridDim.y workgroups share a tile (few tiles: the launch is latency-bound, a quarter of the rows each): q4 in [q_lo, q_hi)
    const int q_per = 32 / (int)gridDim.y, q_lo = (int)blockIdx.y * q_per, q_hi = q_lo + q_per;
    for (int q4 = q_lo; q4 < q_hi; ++q4) {
        const int r = ry + 4 * q4;
        const long gr = (long)tm * GP_TILE + r;
        if (gr >= N) continue;
        const double2_t w2 = *(const double2_t *)(Wi + gr * ldw + gc);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long gcc = gc + e;
            if (gcc > gr || gcc >= N) continue;
            const int c = cx + e;
            double s = 0.0;
            double dq2[GCH];
#pragma unroll
            for (int q = 0; q < GCH; ++q) dq2[q] = 0.0;
            for (int d = 0; d < D; ++d) {
                const double df = xi[d * GP_TILE + r] - xj[d * GP_TILE + c];
                s = fma(df, df, s);
            }
#pragma unroll
            for (int q = 0; q < GCH; ++q) {
                const int d = d0 + q;
                if (d < D) {
                    const double df = xi[d * GP_TILE + r] - xj[d * GP_TILE + c];
                    dq2[q] = df * df;
                }
            }
            double aa = 0.0;
            for (int p = 0; p < P; ++p) aa = fma(ai[p * GP_TILE + r], aj[p * GP_TILE + c], aa);
            const double dLdK = 0.5 * (aa - (double)P * w2[e]);  // exact_gaussian_inference.py:70
            double kv, gv;
            k_and_g(kp.kernel, kp.variance, s, kv, gv);
            const bool diag = (gcc == gr);
            if (kp.gower) {              // product of the 1-D factors, diagonal included (variance^D there, as the fork's K has it)
                kv = 1.0;
                for (int d = 0; d < D; ++d) {
                    const double df = ui[d * GP_TILE + r] - uj[d * GP_TILE + c];
                    const double rr = kp.gdisc[d] ? (df != 0.0 ? 1.0 : 0.0) : fabs(df);
                    kv *= gp_k_of_r2(kp.kernel, kp.variance, rr * rr);
                }
            } else if (diag) {
                kv = kp.variance;  // Kdiag is exactly the variance (stationary.py:162-166, r = 0)
            }
            const double w = diag ? 1.0 : 2.0;
            acc[0] = fma(w * kv, dLdK, acc[0]);
            if (diag) acc[1] += dLdK;
            const double tq = diag ? 0.0 : w * gv * dLdK;  // _inv_dist is 0 on the diagonal (stationary.py:251-258)
            if (ard) {
#pragma unroll
                for (int q = 0; q < GCH; ++q) acc[2 + q] = fma(tq, dq2[q], acc[2 + q]);
            } else {
                acc[2] = fma(tq, s, acc[2]);
            }
        }
    }
    // block reduce (fixed order)
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        const double v = wave_sum_g(acc[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (tid < NACC) partial[(t * gridDim.y + blockIdx.y) * NACC + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ __launch_bounds__(1024) void sum_partials_kernel(const double *partial, long ntile, int nacc, double *out) {
    __shared__ double sh[16];
    const int q = blockIdx.x;
    double s = 0.0;
    for (long t = threadIdx.x; t < ntile; t += 1024) s += partial[t * nacc + q];
    s = wave_sum_g(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < 16; ++i) r += sh[i];
        out[q] = r;
    }
}

// out (device, NACC doubles): sums for dims [d0, d0+GCH)
void launch_lml_grad(hipStream_t s, const double *X, long N, long Npad, const KernParams &kp, int ard, int d0,
                     const double *alpha, int P, const double *Wi, long ldw, double *partial, double *out) {
    const int nt = (int)(Npad / GP_TILE);
    const long ntile = (long)nt * (nt + 1) / 2;
    const size_t shm = ((size_t)(kp.gower ? 4 : 2) * kp.D * GP_TILE + (size_t)2 * P * GP_TILE) * sizeof(double);
    // few tiles (N <= ~2900): four workgroups per tile, 32 rows each -- 6 workgroups of 128 x 128 covariance evaluations were a
    // quarter of an LML + gradient evaluation at N = 300; the partial sums are added per (tile, quarter) in the same fixed order
    const unsigned split = ntile < 256 ? 4u : 1u;
    GP_LAUNCH(lml_grad_tile_kernel, dim3((unsigned)ntile, split), dim3(256), shm, s, X, N, kp, ard, d0, alpha, Npad,
                       P, Wi, ldw, partial);
    GP_LAUNCH(sum_partials_kernel, dim3(NACC), dim3(1024), 0, s, partial, ntile * split, NACC, out);
}

// ---- predictive gradients (gp.py:407-454) ----------------------------------------------------------
// One workgroup per candidate m:
//   dmdx[m, q, p] = sum_n g(r_mn) alpha_p[n] (xs_mq - x_nq) / l_q^2
//   dvdx[m, q]    = sum_n g(r_mn) (-2 beta[m, n]) (xs_mq - x_nq) / l_q^2,   beta = K(Xs, X) Ky^-1
__global__ __launch_bounds__(256) void predict_grad_kernel(const double *Xs, const double *X, long N, KernParams kp,
                                                           const double *alpha, long lda_, int P, const double *beta,
                                                           long ldb, int d0, double *dmdx, double *dvdx, int first_pass) {
    __shared__ double xs[GP_MAX_D];
    __shared__ double red[4][GCH];
    const int D = kp.D;
    const long m = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < D) xs[tid] = Xs[m * D + tid] / kp.ls[tid];
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    // pass A: variance gradient; passes p: mean gradients (weights differ, geometry identical)
    for (int pass = first_pass; pass <= P; ++pass) {   // first_pass = 1: mean gradients only (beta not read)
        double acc[GCH];
#pragma unroll
        for (int q = 0; q < GCH; ++q) acc[q] = 0.0;
        for (long n = tid; n < N; n += 256) {
            double s = 0.0;
            double dq[GCH];
#pragma unroll
            for (int q = 0; q < GCH; ++q) dq[q] = 0.0;
            for (int d = 0; d < D; ++d) {
                const double df = xs[d] - X[n * D + d] / kp.ls[d];
                s = fma(df, df, s);
            }
#pragma unroll
            for (int q = 0; q < GCH; ++q) {
                const int d = d0 + q;
                if (d < D) dq[q] = xs[d] - X[n * D + d] / kp.ls[d];
            }
            double kv, gv;
            k_and_g(kp.kernel, kp.variance, s, kv, gv);
            if (s == 0.0) gv = 0.0;  // invdist = 0 where the distance is exactly 0 (stationary.py:251-258)
            const double w = (pass == 0) ? -2.0 * beta[m * ldb + n] : alpha[(long)(pass - 1) * lda_ + n];
            const double t = gv * w;
#pragma unroll
            for (int q = 0; q < GCH; ++q) acc[q] = fma(t, dq[q], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < GCH; ++q) {
            const double v = wave_sum_g(acc[q]);
            if (lane == 0) red[wv][q] = v;
        }
        __syncthreads();
        if (tid < GCH && d0 + tid < D) {
            // (x - x') / l^2 = scaled difference / l
            const double v = (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]) / kp.ls[d0 + tid];
            if (pass == 0)
                dvdx[m * D + d0 + tid] = v;
            else
                dmdx[(m * D + d0 + tid) * P + (pass - 1)] = v;
        }
        __syncthreads();
    }
}
void launch_predict_grad(hipStream_t s, const double *Xs, long M, const double *X, long N, const KernParams &kp,
                         const double *alpha, long lda_, int P, const double *beta, long ldb, double *dmdx,
                         double *dvdx) {
    // beta == nullptr: the mean's gradients only (dvdx untouched)
    for (int d0 = 0; d0 < kp.D; d0 += GCH)
        GP_LAUNCH(predict_grad_kernel, dim3((unsigned)M), dim3(256), 0, s, Xs, X, N, kp, alpha, lda_, P, beta,
                           ldb, d0, dmdx, dvdx, beta ? 0 : 1);
}

// ---- acquisition gradients (EI.py:42-51, LCB.py:39-46, MPI.py:42-51; gpmodel.py:131-142) ------------
__global__ void acq_grad_kernel(int type, double par, double fmin, double y_mean, double y_std, const double *mean,
                                const double *var, const double *dmdx, const double *dvdx, long M, int D, double *out,
                                double *dout) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double f, c_m, c_s, ds_scale;                          // df = c_m * dmdx + c_s * dsdx (acq_math.h)
    acq_terms(type, par, fmin, y_mean, y_std, mean[i], var[i], f, c_m, c_s, ds_scale);
    out[i] = -f;
    for (int d = 0; d < D; ++d) {
        const double dm = dmdx[i * D + d] * y_std;
        const double ds = dvdx[i * D + d] * ds_scale;
        dout[i * D + d] = -(c_m * dm + c_s * ds);
    }
}
void launch_acq_grad(hipStream_t s, int type, double par, double fmin, double y_mean, double y_std, const double *mean,
                     const double *var, const double *dmdx, const double *dvdx, long M, int D, double *out,
                     double *dout) {
    GP_LAUNCH(acq_grad_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, type, par, fmin, y_mean,
                       y_std, mean, var, dmdx, dvdx, M, D, out, dout);
}

// diag(C) += v for the first n rows (noise on the full covariance)
__global__ void add_diag2_kernel(double *A, long lda, long n, double v) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * lda + i] += v;
}
void launch_add_diag(hipStream_t s, double *A, long lda, long N, double v) {
    GP_LAUNCH(add_diag2_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, A, lda, N, v);
}
// out[0] = sum_i A[i][i], out[1] = min_i A[i][i]  (jitchol looks at both: linalg.py:61-66)
__global__ __launch_bounds__(1024) void trace_kernel(const double *A, long lda, long N, double *out) {
    __shared__ double sh[16], shm[16];
    double s = 0.0, mn = INFINITY;
    for (long i = threadIdx.x; i < N; i += 1024) {
        const double d = A[i * lda + i];
        s += d;
        mn = fmin(mn, d);
    }
    s = wave_sum_g(s);
    for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, __shfl_down(mn, o));
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = s;
        shm[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, m = INFINITY;
        for (int i = 0; i < 16; ++i) {
            r += sh[i];
            m = fmin(m, shm[i]);
        }
        out[0] = r;
        out[1] = m;
    }
}
void launch_trace(hipStream_t s, const double *A, long lda, long N, double *out) {
    GP_LAUNCH(trace_kernel, dim3(1), dim3(1024), 0, s, A, lda, N, out);
}

__global__ void set_identity_blocks_kernel(double *T, long n) {
    const long j2 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const long row = blockIdx.y;  // stacked row index over all blocks
    if (j2 >= n) return;
    const long i = row % n;
    double2_t v;
    v[0] = (i == j2) ? 1.0 : 0.0;
    v[1] = (i == j2 + 1) ? 1.0 : 0.0;
    *(double2_t *)(T + row * n + j2) = v;
}
void launch_set_identity_blocks(hipStream_t s, double *T, long n, int nb) {
    dim3 grid((unsigned)((n / 2 + 255) / 256), (unsigned)(n * nb));
    GP_LAUNCH(set_identity_blocks_kernel, grid, dim3(256), 0, s, T, n);
}
__global__ void transpose_blocks_kernel(double *dst, const double *src, long n) {
    __shared__ double t[32][33];
    const long b = blockIdx.z;
    const double *S = src + b * n * n;
    double *D = dst + b * n * n;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) t[r][tx] = S[((long)blockIdx.y * 32 + r) * n + (long)blockIdx.x * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) D[((long)blockIdx.x * 32 + r) * n + (long)blockIdx.y * 32 + tx] = t[tx][r];
}
void launch_transpose_blocks(hipStream_t s, double *dst, const double *src, long n, int nb) {
    dim3 grid((unsigned)(n / 32), (unsigned)(n / 32), (unsigned)nb);
    GP_LAUNCH(transpose_blocks_kernel, grid, dim3(256), 0, s, dst, src, n);
}

// ---- local-penalisation epilogue (GPyOpt/GPyOpt/acquisitions/LP.py:40-110) -----------------------------
// in: negacq[M] = -acq(x) (gp_acq output).  out[M] = -log-transformed acq - sum_k logcdf((|x - x0_k| - r_k)/s_k)  (acq_math.h)
__global__ void lp_kernel(const double *negacq, const double *Xs, long M, int D, const double *Xb, int nb,
                          const double *r0, const double *s0, int transform, double *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[i] = lp_value(negacq[i], Xs + i * D, D, Xb, nb, r0, s0, transform);
}
void launch_lp(hipStream_t s, const double *negacq, const double *Xs, long M, int D, const double *Xb, int nb,
               const double *r0, const double *s0, int transform, double *out) {
    GP_LAUNCH(lp_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, negacq, Xs, M, D, Xb, nb, r0, s0,
                       transform, out);
}
// value and gradient of the penalised acquisition (LP.py:112-140).  in: negacq[M] = -acq(x), dneg[M, D] = -d acq / dx (the
// outputs of acq_grad_kernel), overwritten in place by the penalised value and its gradient (acq_math.h).
__global__ void lp_grad_kernel(double *negacq, double *dneg, const double *Xs, long M, int D, const double *Xb, int nb,
                               const double *r0, const double *s0, int transform) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    negacq[i] = lp_value_grad(negacq[i], dneg + i * D, Xs + i * D, D, Xb, nb, r0, s0, transform);
}
void launch_lp_grad(hipStream_t s, double *negacq, double *dneg, const double *Xs, long M, int D, const double *Xb, int nb,
                    const double *r0, const double *s0, int transform) {
    GP_LAUNCH(lp_grad_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, negacq, dneg, Xs, M, D, Xb, nb, r0,
                       s0, transform);
}
__global__ void mask_kernel(double *v, const long long *idx, int n, double fill) {
    const int i = threadIdx.x;
    if (i < n) v[idx[i]] = fill;
}
void launch_mask(hipStream_t s, double *v, const long long *idx, int n, double fill) {
    if (n > 0) GP_LAUNCH(mask_kernel, dim3(1), dim3(256), 0, s, v, idx, n, fill);
}

__global__ __launch_bounds__(256) void dldk_kernel(double *out, long ldo, const double *alpha, long lda_, int P,
                                                   const double *Wi, long ldw, long N) {
    const long j = blockIdx.x * 256L + threadIdx.x, i = blockIdx.y;
    if (j >= N) return;
    double s = 0.0;
    for (int p = 0; p < P; ++p) s = fma(alpha[p * lda_ + i], alpha[p * lda_ + j], s);
    out[i * ldo + j] = 0.5 * (s - (double)P * Wi[i * ldw + j]);
}
void launch_dldk(hipStream_t s, double *out, long ldo, const double *alpha, long lda_, int P, const double *Wi,
                 long ldw, long N) {
    GP_LAUNCH(dldk_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, s, out, ldo, alpha,
                       lda_, P, Wi, ldw, N);
}

__global__ __launch_bounds__(256) void zero_upper_diag_kernel(double *A, long lda) {
    double *At = A + (long)blockIdx.x * GP_TILE * lda + (long)blockIdx.x * GP_TILE;
    for (int id = threadIdx.x; id < GP_TILE * GP_TILE; id += 256) {
        const int r = id >> 7, c = id & 127;
        if (c > r) At[(long)r * lda + c] = 0.0;
    }
}
void launch_zero_upper_diag(hipStream_t s, double *A, long lda, int nt) {
    GP_LAUNCH(zero_upper_diag_kernel, dim3((unsigned)nt), dim3(256), 0, s, A, lda);
}

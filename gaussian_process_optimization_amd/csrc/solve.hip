// Vector-sized kernels around the factorisation: log-determinant, alpha back-substitution,
// posterior reductions, training-set mean (fmin), acquisition values and arg-best.
// All reductions are fixed-order (no float atomics) so results are bitwise reproducible.
#include "gphip_internal.h"
#include "../../include/gphip.h"
#include <algorithm>

// ---- block reduction helpers (256 or 1024 threads) ---------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

template <int NT>
__device__ __forceinline__ double block_sum(double v, double *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < NT / 64; ++i) r += sh[i];
        sh[0] = r;
    }
    __syncthreads();
    r = sh[0];
    __syncthreads();
    return r;
}

// ---- logdet = 2 sum log L_ii  (GPy/GPy/util/linalg.py:208) --------------------------------
__global__ __launch_bounds__(1024) void logdet_kernel(const double *A, long lda, long N, double *out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < N; i += 1024) s += log(A[i * lda + i]);
    s = block_sum<1024>(s, sh);
    if (threadIdx.x == 0) out[0] = 2.0 * s;
}
void launch_logdet(hipStream_t s, const double *A, long lda, long N, double *out) {
    GP_LAUNCH(logdet_kernel, dim3(1), dim3(1024), 0, s, A, lda, N, out);
}

// ---- out[p] = z_p . z_p  (= Y^T Ky^-1 Y, the data-fit term of exact_gaussian_inference.py:62) ----
__global__ __launch_bounds__(1024) void rhs_sumsq_kernel(const double *A, long lda, long N, long Npad, double *out) {
    __shared__ double sh[16];
    const double *z = A + (Npad + blockIdx.x) * lda;
    double s = 0.0;
    for (long i = threadIdx.x; i < N; i += 1024) s = fma(z[i], z[i], s);
    s = block_sum<1024>(s, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
void launch_rhs_sumsq(hipStream_t s, const double *A, long lda, long N, long Npad, int P, double *out) {
    GP_LAUNCH(rhs_sumsq_kernel, dim3(P), dim3(1024), 0, s, A, lda, N, Npad, out);
}

// ---- alpha = L^-T z  (second half of dpotrs, exact_gaussian_inference.py:60) -------------------
// Blocked by panels of PB = W*128 columns, from the last panel to the first, using the inverted
// diagonal panels invP_J = L_JJ^-1 (built once per fit on the MFMA GEMM, see api.hip):
//   alpha_J = invP_J^T (z_J - w_J)                 (panel_solve: one workgroup per 128 outputs)
//   w_j    += L[J rows, j cols]^T alpha_J, j < J   (panel_update: one workgroup per 128 columns)
// 2 launches per panel (32 at N = 16384) instead of one per 128-row block; L is streamed once.
#define PT 1024  // threads of the panel kernels: 128 columns x 8 row groups, 8 independent loads in flight each
__global__ __launch_bounds__(PT) void panel_solve_kernel(const double *invP, long ldp, int Kp, const double *Z,
                                                         long ldz, const double *w, long ldw, int P, long off,
                                                         double *Aout) {
    __shared__ double part[PT];
    extern __shared__ double v[];  // Kp
    const int tid = threadIdx.x;
    const int c = blockIdx.x * GP_TILE + (tid & 127), rg = tid >> 7;
    // invP lower triangular: only rows r >= first column of this chunk can be non-zero
    const int r0 = blockIdx.x * GP_TILE;
    const int per = (Kp - r0) / 8;  // multiple of 16 (Kp - r0 is a multiple of 128)
    for (int p = 0; p < P; ++p) {
        for (int r = tid; r < Kp; r += PT) v[r] = Z[p * ldz + off + r] - w[p * ldw + off + r];
        __syncthreads();
        const double *ip = invP + (long)(r0 + rg * per) * ldp + c;
        const double *vp = v + r0 + rg * per;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int r = 0; r < per; r += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = ip[(long)(r + u) * ldp];
            s0 = fma(x[0], vp[r + 0], s0); s1 = fma(x[1], vp[r + 1], s1);
            s2 = fma(x[2], vp[r + 2], s2); s3 = fma(x[3], vp[r + 3], s3);
            s0 = fma(x[4], vp[r + 4], s0); s1 = fma(x[5], vp[r + 5], s1);
            s2 = fma(x[6], vp[r + 6], s2); s3 = fma(x[7], vp[r + 7], s3);
        }
        part[tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < GP_TILE) {
            double t = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += part[tid + 128 * u];
            Aout[p * ldz + off + c] = t;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(PT) void panel_update_kernel(const double *L, long lda, int Kp, long rowoff,
                                                          const double *alpha, long ldz, int P, double *w, long ldw) {
    __shared__ double part[PT];
    extern __shared__ double al[];  // Kp
    const int tid = threadIdx.x;
    const long c = (long)blockIdx.x * GP_TILE + (tid & 127);
    const int rg = tid >> 7;
    const int per = Kp / 8;  // multiple of 16
    const double *Lb = L + (rowoff + (long)rg * per) * lda + c;
    for (int p = 0; p < P; ++p) {
        for (int r = tid; r < Kp; r += PT) al[r] = alpha[p * ldz + rowoff + r];
        __syncthreads();
        const double *ap = al + rg * per;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int r = 0; r < per; r += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = Lb[(long)(r + u) * lda];
            s0 = fma(x[0], ap[r + 0], s0); s1 = fma(x[1], ap[r + 1], s1);
            s2 = fma(x[2], ap[r + 2], s2); s3 = fma(x[3], ap[r + 3], s3);
            s0 = fma(x[4], ap[r + 4], s0); s1 = fma(x[5], ap[r + 5], s1);
            s2 = fma(x[6], ap[r + 6], s2); s3 = fma(x[7], ap[r + 7], s3);
        }
        part[tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (tid < GP_TILE) {
            double t = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += part[tid + 128 * u];
            w[p * ldw + c] += t;
        }
        __syncthreads();
    }
}
__global__ void zero_kernel(double *p, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}
// invP: panels of W tiles, each stored as a PB x PB row-major block (PB = W*128) at invP + J*PB*PB
void launch_trsv_backward(hipStream_t s, const double *L, long lda, const double *invP, int W, long Npad,
                          const double *Z, long ldz, int P, double *Aout, double *w) {
    const long nw = (long)P * Npad;
    GP_LAUNCH(zero_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w, nw);
    const int nt = (int)(Npad / GP_TILE);
    const long PB = (long)W * GP_TILE;
    const int nJ = (nt + W - 1) / W;
    for (int J = nJ - 1; J >= 0; --J) {
        const int J0 = J * W, J1 = std::min(J0 + W, nt);
        const int Kp = (J1 - J0) * GP_TILE;
        const long off = (long)J0 * GP_TILE;
        GP_LAUNCH(panel_solve_kernel, dim3(J1 - J0), dim3(PT), Kp * sizeof(double), s, invP + (long)J * PB * PB,
                           PB, Kp, Z, ldz, w, Npad, P, off, Aout);
        if (J0 > 0)
            GP_LAUNCH(panel_update_kernel, dim3(J0), dim3(PT), Kp * sizeof(double), s, L, lda, Kp, off, Aout,
                               ldz, P, w, Npad);
    }
}

// ---- posterior reductions (posterior.py:277,292-295; gaussian.py:109) --------------------------
__global__ __launch_bounds__(256) void predict_reduce_kernel(const double *T, long ldt, long N, const double *Z,
                                                             long ldz, int P, double kss, double noise_add,
                                                             double *mean, double *var) {
    __shared__ double sh[4];
    const long c = blockIdx.x;
    const double *t = T + c * ldt;
    const long n2 = N >> 1;
    double s = 0.0, m0 = 0.0;
    // one pass over the row: sum of squares and the first output's mean together (P = 1 is the BO case)
    for (long i = threadIdx.x; i < n2; i += 256) {
        const double2_t x = *(const double2_t *)(t + 2 * i);
        const double2_t y = *(const double2_t *)(Z + 2 * i);
        s = fma(x[0], x[0], s);
        s = fma(x[1], x[1], s);
        m0 = fma(x[0], y[0], m0);
        m0 = fma(x[1], y[1], m0);
    }
    if ((N & 1) && threadIdx.x == 0) {
        s = fma(t[N - 1], t[N - 1], s);
        m0 = fma(t[N - 1], Z[N - 1], m0);
    }
    s = block_sum<256>(s, sh);
    if (threadIdx.x == 0) var[c] = (kss - s) + noise_add;
    m0 = block_sum<256>(m0, sh);
    if (threadIdx.x == 0) mean[c * P] = m0;
    for (int p = 1; p < P; ++p) {
        const double *z = Z + p * ldz;
        double m = 0.0;
        for (long i = threadIdx.x; i < n2; i += 256) {
            const double2_t x = *(const double2_t *)(t + 2 * i);
            const double2_t y = *(const double2_t *)(z + 2 * i);
            m = fma(x[0], y[0], m);
            m = fma(x[1], y[1], m);
        }
        if ((N & 1) && threadIdx.x == 0) m = fma(t[N - 1], z[N - 1], m);
        m = block_sum<256>(m, sh);
        if (threadIdx.x == 0) mean[c * P + p] = m;
    }
}
void launch_predict_reduce(hipStream_t s, const double *T, long ldt, long M, long N, const double *Z, long ldz, int P,
                           double kss, double noise_add, double *mean, double *var) {
    if (M <= 0) return;
    GP_LAUNCH(predict_reduce_kernel, dim3((unsigned)M), dim3(256), 0, s, T, ldt, N, Z, ldz, P, kss, noise_add,
                       mean, var);
}

// ---- posterior mean at the training inputs (GPModel.get_fmin, gpmodel.py:125-129) --------------
#define k_of_r2_s gp_k_of_r2   // gphip_internal.h
#define TM_SPLIT 8
// grid (N/128 row tiles, TM_SPLIT column slices); part[slice][i]
__global__ __launch_bounds__(256) void train_mean_kernel(const double *X, long N, KernParams kp, const double *alpha,
                                                         double *part) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *xi = sm;                         // [D][128]
    double *xj = sm + (long)kp.D * GP_TILE;  // [D][128]
    double *aj = xj + (long)kp.D * GP_TILE;  // [128]
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int r = tid & 127, h = tid >> 7;
    const long row0 = (long)blockIdx.x * GP_TILE;
    const long ntile = (N + GP_TILE - 1) / GP_TILE;
    for (int idx = tid; idx < GP_TILE * kp.D; idx += 256) {
        const int rr = idx / kp.D, d = idx - rr * kp.D;
        const long g = row0 + rr;
        xi[d * GP_TILE + rr] = (g < N) ? X[g * kp.D + d] / kp_div(kp, d) : 0.0;
    }
    double acc = 0.0;
    for (long tj = blockIdx.y; tj < ntile; tj += TM_SPLIT) {
        __syncthreads();
        for (int idx = tid; idx < GP_TILE * kp.D; idx += 256) {
            const int rr = idx / kp.D, d = idx - rr * kp.D;
            const long g = tj * GP_TILE + rr;
            xj[d * GP_TILE + rr] = (g < N) ? X[g * kp.D + d] / kp_div(kp, d) : 0.0;
        }
        if (tid < GP_TILE) {
            const long g = tj * GP_TILE + tid;
            aj[tid] = (g < N) ? alpha[g] : 0.0;
        }
        __syncthreads();
        for (int jj = h * 64; jj < h * 64 + 64; ++jj) {
            double kv;
            if (kp.gower) {
                kv = 1.0;
                for (int d = 0; d < kp.D; ++d) {
                    const double df = xi[d * GP_TILE + r] - xj[d * GP_TILE + jj];
                    const double rr = kp.gdisc[d] ? (df != 0.0 ? 1.0 : 0.0) : fabs(df);
                    kv *= k_of_r2_s(kp.kernel, kp.variance, rr * rr);
                }
            } else {
                double s = 0.0;
                for (int d = 0; d < kp.D; ++d) {
                    const double df = xi[d * GP_TILE + r] - xj[d * GP_TILE + jj];
                    s = fma(df, df, s);
                }
                kv = k_of_r2_s(kp.kernel, kp.variance, s);
            }
            acc = fma(kv, aj[jj], acc);
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < GP_TILE && row0 + tid < N) part[(long)blockIdx.y * N + row0 + tid] = red[tid] + red[tid + 128];
}
__global__ void train_mean_sum_kernel(const double *part, long N, double *mu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
    for (int q = 0; q < TM_SPLIT; ++q) s += part[(long)q * N + i];
    mu[i] = s;
}
// mu must have room for (TM_SPLIT + 1) * N doubles: [mu | partials]
void launch_train_mean(hipStream_t s, const double *X, long N, const KernParams &kp, const double *alpha, double *mu) {
    const size_t shm = ((size_t)2 * kp.D * GP_TILE + GP_TILE) * sizeof(double);
    dim3 grid((unsigned)((N + GP_TILE - 1) / GP_TILE), TM_SPLIT);
    GP_LAUNCH(train_mean_kernel, grid, dim3(256), shm, s, X, N, kp, alpha, mu + N);
    GP_LAUNCH(train_mean_sum_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, mu + N, N, mu);
}

// Posterior mean at the training inputs from the normal equations instead of an N^2 pass:
//   (K + d I) alpha = y  =>  K alpha = y - d alpha,   d = noise + 1e-8 (+ jitter)
// GPModel.get_fmin (gpmodel.py:138-142) takes the minimum of exactly this vector; SURVEY.md 8a A9 marks the
// reference's N^3 recomputation as cacheable.  The two differ by the residual of the solve (backward stable).
__global__ void train_mean_identity_kernel(const double *Y, const double *alpha, double d, long N, double *mu) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) mu[i] = fma(-d, alpha[i], Y[i]);
}
void launch_train_mean_identity(hipStream_t s, const double *Y, const double *alpha, double d, long N, double *mu) {
    GP_LAUNCH(train_mean_identity_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, Y, alpha, d, N, mu);
}

// ---- min / arg-best (NumPy tie rule: lowest index) -------------------------------------------
__device__ __forceinline__ void best_combine(double &v, long long &i, double v2, long long i2, int sense) {
    // sense +1: larger wins; -1: smaller wins; ties -> lower index
    const bool better = (sense > 0) ? (v2 > v) : (v2 < v);
    if (better || (v2 == v && i2 < i)) {
        v = v2;
        i = i2;
    }
}
__global__ __launch_bounds__(256) void argbest_kernel(const double *x, long n, int sense, double *ov, long long *oi,
                                                      const long long *in_idx) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    const int tid = threadIdx.x;
    double v = (sense > 0) ? -INFINITY : INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (long i = (long)blockIdx.x * 256 + tid; i < n; i += (long)gridDim.x * 256)
        best_combine(v, bi, x[i], in_idx ? in_idx[i] : (long long)i, sense);
    sv[tid] = v;
    si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) best_combine(sv[tid], si[tid], sv[tid + o], si[tid + o], sense);
        __syncthreads();
    }
    if (tid == 0) {
        ov[blockIdx.x] = sv[0];
        oi[blockIdx.x] = si[0];
    }
}
// scratch_val / scratch_idx: >= 256 entries
void launch_argbest(hipStream_t s, const double *v, long n, int sense, double *best_val, long long *best_idx,
                    double *scratch_val, long long *scratch_idx) {
    int nb = (int)((n + 255) / 256);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    GP_LAUNCH(argbest_kernel, dim3(nb), dim3(256), 0, s, v, n, sense, scratch_val, scratch_idx,
                       (const long long *)nullptr);
    GP_LAUNCH(argbest_kernel, dim3(1), dim3(256), 0, s, scratch_val, (long)nb, sense, best_val, best_idx,
                       (const long long *)scratch_idx);
}
void launch_min(hipStream_t s, const double *v, long n, double *out) {
    // out: [0] = min, scratch after it: needs 1 + 256 doubles and 257 long longs behind (see api)
    double *sv = out + 1;
    long long *si = (long long *)(out + 1 + 256);
    launch_argbest(s, v, n, -1, out, si + 256, sv, si);
}

// ---- acquisition values (GPyOpt acquisitions/{EI,LCB,MPI}.py, util/general.py:113-129) --------
__global__ void acq_kernel(int type, double par, double fmin, double y_mean, double y_std, const double *mean,
                           const double *var, long M, double *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    // GP.predict un-normalisation (gp.py:344-352) then GPModel._predict clip (gpmodel.py:99)
    const double m = mean[i] * y_std + y_mean;
    double v = var[i] * (y_std * y_std);
    v = (v < 1e-10) ? 1e-10 : v;
    double s = sqrt(v);
    double f;
    if (type == GP_ACQ_LCB) {
        f = -m + par * s;  // LCB.py:36
    } else {
        if (s < 1e-10) s = 1e-10;  // general.py:121-124
        const double u = (fmin - m - par) / s;
        const double phi = exp(-0.5 * u * u) / 2.50662827463100050241576528481105;  // sqrt(2 pi)
        const double Phi = 0.5 * erfc(-u / 1.41421356237309504880168872420970);
        f = (type == GP_ACQ_EI) ? s * (u * Phi + phi) : Phi;  // EI.py:39 / MPI.py:39
    }
    out[i] = -f;  // acquisitions/base.py:39 (cost 1, no constraints)
}
void launch_acq(hipStream_t s, int type, double par, double fmin, double y_mean, double y_std, const double *mean,
                const double *var, long M, double *out) {
    if (M <= 0) return;
    GP_LAUNCH(acq_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, type, par, fmin, y_mean, y_std,
                       mean, var, M, out);
}

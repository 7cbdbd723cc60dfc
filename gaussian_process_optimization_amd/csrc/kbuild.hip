// Covariance-matrix construction kernels (HBM-bound): K(X,X)+diag and K(Xs,X).
//
// Reference: Stationary._scaled_dist / _unscaled_dist (GPy/GPy/kern/src/stationary.py:155-193),
// RBF.K_of_r (rbf.py:50-51), Matern52.K_of_r (stationary.py:575-576), and the diagonal term of
// ExactGaussianInference (exact_gaussian_inference.py:55-56: Ky = K; diag += noise + 1e-8).
//
// The reference forms r^2 by the Gram trick (-2 X X^T + |x|^2 + |x'|^2, diagonal forced to 0,
// clipped at 0).  Here r^2 = sum_d ((x_d - x'_d) / l_d)^2 is accumulated directly: D <= 64, the
// inputs for a 128x128 output tile are 2 x 128 x D doubles staged once in LDS, and the direct
// form has no cancellation (its diagonal is exactly 0 and it is never negative), so it sits
// inside the reference's own rounding error.
//
// Tile = 128 x 128 outputs per 256-thread workgroup; a thread owns 2 adjacent columns (one
// 16-B store, 1 KiB contiguous per wave-row) of 32 rows.  Algorithmic bytes: 8 N D read +
// 8 N^2 / 2 written (lower tiles only).
#include "gphip_internal.h"

#define k_of_r2 gp_k_of_r2   // gphip_internal.h
// one 1-D factor of the Gower product kernel: K_of_r(|dx|) for a continuous dimension (dx already divided by
// the variable's range), K_of_r(dx != 0) for a discrete one (stationary.py:122-129)
__device__ __forceinline__ double gower_factor(int kernel, double variance, double dx, int disc) {
    const double r = disc ? (dx != 0.0 ? 1.0 : 0.0) : fabs(dx);
    return k_of_r2(kernel, variance, r * r);
}

// Stage rows [row0, row0+128) of X (N x D row-major) divided by the lengthscale (stationary.py:188-191) into LDS transposed: dst[d*128 + r].
__device__ __forceinline__ void stage_rows_T(double *dst, const double *X, long row0, long N, int D,
                                             const double *ls, int tid) {
    for (int idx = tid; idx < GP_TILE * D; idx += 256) {
        const int r = idx / D, d = idx - r * D;
        const long g = row0 + r;
        dst[d * GP_TILE + r] = (g < N) ? X[g * D + d] / ls[d] : 0.0;
    }
}

// DU > 0: the dimension loop unrolled to DU (LDS rows D .. DU-1 are staged as zeros), the thread's two columns held in registers
// across its 32 rows; DU = 0: run-time loop (D > 16, and the Gower product kernel).
// grid: lower tiles enumerated row-wise (tm >= tn): t = tm(tm+1)/2 + tn
template <int DU>
__global__ __launch_bounds__(256) void kbuild_kernel(double *A, long lda, const double *X, long N, long Npad,
                                                     KernParams kp, double diag_add, int full, int nt) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int DS = DU > 0 ? DU : kp.D;        // staged dimensions
    double *xi = sm;                          // [DS][128]
    double *xj = sm + (long)DS * GP_TILE;     // [DS][128]
    __shared__ double ils[GP_MAX_D];
    const int tid = threadIdx.x;
    int tm, tn;
    if (full) {
        tm = blockIdx.x / nt;
        tn = blockIdx.x % nt;
    } else {
        const long t = blockIdx.x;
        tm = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long)tm * (tm + 1) / 2 > t) --tm;
        while ((long)(tm + 1) * (tm + 2) / 2 <= t) ++tm;
        tn = (int)(t - (long)tm * (tm + 1) / 2);
    }
    if (tid < kp.D) ils[tid] = kp_div(kp, tid);
    __syncthreads();
    stage_rows_T(xi, X, (long)tm * GP_TILE, N, kp.D, ils, tid);
    stage_rows_T(xj, X, (long)tn * GP_TILE, N, kp.D, ils, tid);
    if (DU > 0)
        for (int idx = kp.D * GP_TILE + tid; idx < DU * GP_TILE; idx += 256) xi[idx] = xj[idx] = 0.0;
    __syncthreads();

    const int cx = (tid & 63) * 2;   // two columns
    const int ry = tid >> 6;         // rows ry + 4q
    const long gc = (long)tn * GP_TILE + cx;
    double2_t bq[DU > 0 ? DU : 1];
    if (DU > 0) {
#pragma unroll
        for (int d = 0; d < DU; ++d) bq[d] = *(const double2_t *)(xj + d * GP_TILE + cx);
    }
    for (int q = 0; q < 32; ++q) {
        const int r = ry + 4 * q;
        const long gr = (long)tm * GP_TILE + r;
        double2_t out;
        if (kp.gower) {
            double p0 = 1.0, p1 = 1.0;
            for (int d = 0; d < kp.D; ++d) {
                const double a = xi[d * GP_TILE + r];
                const double2_t b = *(const double2_t *)(xj + d * GP_TILE + cx);
                p0 *= gower_factor(kp.kernel, kp.variance, a - b[0], kp.gdisc[d]);
                p1 *= gower_factor(kp.kernel, kp.variance, a - b[1], kp.gdisc[d]);
            }
            out[0] = p0;
            out[1] = p1;
        } else {
            double s0 = 0.0, s1 = 0.0;
            if (DU > 0) {
#pragma unroll
                for (int d = 0; d < DU; ++d) {
                    const double a = xi[d * GP_TILE + r];
                    const double d0 = a - bq[d][0], d1 = a - bq[d][1];
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
            } else {
                for (int d = 0; d < kp.D; ++d) {
                    const double a = xi[d * GP_TILE + r];
                    const double2_t b = *(const double2_t *)(xj + d * GP_TILE + cx);
                    const double d0 = a - b[0], d1 = a - b[1];
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
            }
            out[0] = k_of_r2(kp.kernel, kp.variance, s0);
            out[1] = k_of_r2(kp.kernel, kp.variance, s1);
        }
        // diagonal and padding
        if (gr >= N) {
            out[0] = (gr == gc) ? 1.0 : 0.0;
            out[1] = (gr == gc + 1) ? 1.0 : 0.0;
        } else {
            // the Euclidean diagonal is exactly the variance (r forced to 0, stationary.py:164); the Gower product
            // at r = 0 is variance^D and is already in out[]
            if (gc >= N) out[0] = 0.0;
            else if (gr == gc) out[0] = (kp.gower ? out[0] : kp.variance) + diag_add;
            if (gc + 1 >= N) out[1] = 0.0;
            else if (gr == gc + 1) out[1] = (kp.gower ? out[1] : kp.variance) + diag_add;
        }
        *(double2_t *)(A + gr * lda + gc) = out;
    }
}

void launch_kbuild(hipStream_t s, double *A, long lda, const double *X, long N, long Npad,
                   const KernParams &kp, double diag_add, int full) {
    const int nt = (int)(Npad / GP_TILE);
    const long nblk = full ? (long)nt * nt : (long)nt * (nt + 1) / 2;
    const int DU = kp.gower ? 0 : (kp.D <= 8 ? 8 : (kp.D <= 16 ? 16 : 0));
    const size_t shm = (size_t)2 * (DU ? DU : kp.D) * GP_TILE * sizeof(double);
    if (DU == 8)
        GP_LAUNCH(kbuild_kernel<8>, dim3((unsigned)nblk), dim3(256), shm, s, A, lda, X, N, Npad, kp, diag_add, full, nt);
    else if (DU == 16)
        GP_LAUNCH(kbuild_kernel<16>, dim3((unsigned)nblk), dim3(256), shm, s, A, lda, X, N, Npad, kp, diag_add, full, nt);
    else
        GP_LAUNCH(kbuild_kernel<0>, dim3((unsigned)nblk), dim3(256), shm, s, A, lda, X, N, Npad, kp, diag_add, full, nt);
}

__global__ void set_rhs_kernel(double *A, long lda, const double *Y, long N, long Npad, int P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= Npad) return;
    A[(Npad + p) * lda + i] = (p < P && i < N) ? Y[i * P + p] : 0.0;
}

void launch_set_rhs(hipStream_t s, double *A, long lda, const double *Y, long N, long Npad, int P) {
    dim3 grid((unsigned)((Npad + 255) / 256), GP_MAX_RHS);
    GP_LAUNCH(set_rhs_kernel, grid, dim3(256), 0, s, A, lda, Y, N, Npad, P);
}

// T[c][i] = k(xs_c, x_i); tiles (tc over candidates, ti over training points)
template <int DU>
__global__ __launch_bounds__(256) void cross_k_kernel(double *T, long ldt, const double *Xs, long M, const double *X,
                                                      long N, KernParams kp, int nti) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int DS = DU > 0 ? DU : kp.D;
    double *xc = sm;
    double *xt = sm + (long)DS * GP_TILE;
    __shared__ double ils[GP_MAX_D];
    const int tid = threadIdx.x;
    const int tc = blockIdx.x / nti, ti = blockIdx.x % nti;
    if (tid < kp.D) ils[tid] = kp_div(kp, tid);
    __syncthreads();
    stage_rows_T(xc, Xs, (long)tc * GP_TILE, M, kp.D, ils, tid);
    stage_rows_T(xt, X, (long)ti * GP_TILE, N, kp.D, ils, tid);
    if (DU > 0)
        for (int idx = kp.D * GP_TILE + tid; idx < DU * GP_TILE; idx += 256) xc[idx] = xt[idx] = 0.0;
    __syncthreads();
    const int cx = (tid & 63) * 2;
    const int ry = tid >> 6;
    const long gi = (long)ti * GP_TILE + cx;
    double2_t bq[DU > 0 ? DU : 1];
    if (DU > 0) {
#pragma unroll
        for (int d = 0; d < DU; ++d) bq[d] = *(const double2_t *)(xt + d * GP_TILE + cx);
    }
    for (int q = 0; q < 32; ++q) {
        const int r = ry + 4 * q;
        const long gcand = (long)tc * GP_TILE + r;
        double k0, k1;
        if (kp.gower) {
            k0 = 1.0;
            k1 = 1.0;
            for (int d = 0; d < kp.D; ++d) {
                const double a = xc[d * GP_TILE + r];
                const double2_t b = *(const double2_t *)(xt + d * GP_TILE + cx);
                k0 *= gower_factor(kp.kernel, kp.variance, a - b[0], kp.gdisc[d]);
                k1 *= gower_factor(kp.kernel, kp.variance, a - b[1], kp.gdisc[d]);
            }
        } else {
            double s0 = 0.0, s1 = 0.0;
            if (DU > 0) {
#pragma unroll
                for (int d = 0; d < DU; ++d) {
                    const double a = xc[d * GP_TILE + r];
                    const double d0 = a - bq[d][0], d1 = a - bq[d][1];
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
            } else {
                for (int d = 0; d < kp.D; ++d) {
                    const double a = xc[d * GP_TILE + r];
                    const double2_t b = *(const double2_t *)(xt + d * GP_TILE + cx);
                    const double d0 = a - b[0], d1 = a - b[1];
                    s0 = fma(d0, d0, s0);
                    s1 = fma(d1, d1, s1);
                }
            }
            k0 = k_of_r2(kp.kernel, kp.variance, s0);
            k1 = k_of_r2(kp.kernel, kp.variance, s1);
        }
        double2_t out;
        out[0] = (gcand < M && gi < N) ? k0 : 0.0;
        out[1] = (gcand < M && gi + 1 < N) ? k1 : 0.0;
        *(double2_t *)(T + gcand * ldt + gi) = out;
    }
}

void launch_cross_k(hipStream_t s, double *T, long ldt, const double *Xs, long M, long Mpad, const double *X,
                    long N, long Npad, const KernParams &kp) {
    const int ntc = (int)(Mpad / GP_TILE), nti = (int)(Npad / GP_TILE);
    const int DU = kp.gower ? 0 : (kp.D <= 8 ? 8 : (kp.D <= 16 ? 16 : 0));
    const size_t shm = (size_t)2 * (DU ? DU : kp.D) * GP_TILE * sizeof(double);
    const dim3 grid((unsigned)((long)ntc * nti));
    if (DU == 8)
        GP_LAUNCH(cross_k_kernel<8>, grid, dim3(256), shm, s, T, ldt, Xs, M, X, N, kp, nti);
    else if (DU == 16)
        GP_LAUNCH(cross_k_kernel<16>, grid, dim3(256), shm, s, T, ldt, Xs, M, X, N, kp, nti);
    else
        GP_LAUNCH(cross_k_kernel<0>, grid, dim3(256), shm, s, T, ldt, Xs, M, X, N, kp, nti);
}

// ---- K(Xs, X) for a handful of candidate rows (the small-M path, smallm.hip): one training point per thread, every candidate ----
// The tile kernel above computes 128 candidate rows per workgroup whatever M is (15 us for one row at N = 512); here the work is
// M N covariance evaluations and nothing else.  Same arithmetic per entry as cross_k_kernel (inputs divided by the lengthscale, then
// differences, squares summed in dimension order): the same bits.
#define CKR_MAX_M 8
__global__ __launch_bounds__(256) void cross_k_rows_kernel(double *T, long ldt, const double *Xs, int M, const double *X, long N,
                                                           long Npad, KernParams kp) {
    extern __shared__ double xs[];   // [M][D], divided by the lengthscale
    const int D = kp.D;
    for (int idx = threadIdx.x; idx < M * D; idx += 256) xs[idx] = Xs[idx] / kp_div(kp, idx % D);
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    double acc[CKR_MAX_M];
#pragma unroll
    for (int m = 0; m < CKR_MAX_M; ++m) acc[m] = kp.gower ? 1.0 : 0.0;
    if (i < N) {
        for (int d = 0; d < D; ++d) {
            const double b = X[i * D + d] / kp_div(kp, d);
#pragma unroll
            for (int m = 0; m < CKR_MAX_M; ++m) {
                if (m < M) {
                    const double df = xs[m * D + d] - b;
                    if (kp.gower) acc[m] *= gower_factor(kp.kernel, kp.variance, df, kp.gdisc[d]);
                    else acc[m] = fma(df, df, acc[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < CKR_MAX_M; ++m)
        if (m < M) T[(long)m * ldt + i] = (i < N) ? (kp.gower ? acc[m] : k_of_r2(kp.kernel, kp.variance, acc[m])) : 0.0;
}

void launch_cross_k_rows(hipStream_t s, double *T, long ldt, const double *Xs, int M, const double *X, long N, long Npad,
                         const KernParams &kp) {
    if (M < 1 || M > CKR_MAX_M) return;
    GP_LAUNCH(cross_k_rows_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), (size_t)M * kp.D * sizeof(double), s, T, ldt, Xs, M,
              X, N, Npad, kp);
}

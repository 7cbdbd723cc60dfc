// Internal declarations shared by the gphip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>

#define GP_TILE 128          // tile edge of every blocked algorithm (rows/cols per workgroup tile)
#define GP_MAX_RHS 128       // RHS rows carried below the matrix (one tile)

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ---- launch / stream-operation checks ---------------------------------------------------------------------------------
// A refused kernel launch (resources, bad configuration) or a failed event record / stream wait reports through the HIP
// call's return value only; the launch_* helpers return void and sit many frames below the entry point.  Every launch and
// every stream-ordering call therefore NOTES its status in a per-thread slot (the first failure wins), and the entry
// points read that slot where they synchronise (GP_SYNC in api_internal.h): a failure anywhere in the call becomes
// GP_ERR_HIP with the kernel / call named, never a success over unwritten results.
void gp_note_hip(hipError_t e, const char *what, const char *file, int line);
#define GP_NOTE(call) gp_note_hip((call), #call, __FILE__, __LINE__)
#define GP_LAUNCH(kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                   \
        (void)hipGetLastError(); /* a stale status of an earlier, handled call is not this launch's */ \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                 \
        gp_note_hip(hipGetLastError(), #kernel, __FILE__, __LINE__);                       \
    } while (0)

// ---- kernel parameters of the stationary covariance (device copy) ----------
#define GP_MAX_D 64
struct KernParams {
    int kernel;            // GP_KERNEL_*
    int D;
    double variance;
    double ls[GP_MAX_D];       // lengthscale per dimension (iso: all equal)
    // the fork's "Gower" mixed-variable kernel (GPy/GPy/kern/src/stationary.py:116-135): K = prod_d k1(r_d),
    // r_d = |x_d - x'_d| / range_d for continuous dimensions, (x_d != x'_d) for discrete ones
    int gower;
    unsigned char gdisc[GP_MAX_D];  // 1: discrete dimension
    double gdiv[GP_MAX_D];          // staging divisor: range_d (continuous) or 1 (discrete)
};
// exp(x) for x <= 0 (the only arguments a stationary covariance has): n = rint(x log2 e), two-step reduction with the split
// ln 2, Taylor polynomial of degree 13 on |r| <= 0.347 (truncation 4e-18) in Horner form, scaling by v_ldexp_f64.  20
// instructions against ~35 of the library routine (no overflow / positive-range handling); at most 1 ulp from the exact value
// over [-745, 0] (20 M random arguments against an 80-bit reference on the host).  NaN stays NaN, -inf and x < -745.2 give 0.
__device__ __forceinline__ double gp_exp_nonpos(double x) {
    x = (x < -800.0) ? -800.0 : x;   // (a comparison, not fmax: NaN must pass)
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// the stationary covariance as a function of r^2, and with it g(r) = dK_dr(r) / r (finite at r = 0 for both kernels):
// RBF.K_of_r / dK_dr (rbf.py:50-54), Matern52.K_of_r / dK_dr (stationary.py:575-579)
__device__ __forceinline__ double gp_k_of_r2(int kernel, double variance, double r2) {
    if (kernel == 0) return variance * gp_exp_nonpos(-0.5 * r2);
    const double s5 = 2.23606797749978969640917366873128;  // sqrt(5)
    const double r = sqrt(r2);
    return variance * (1.0 + s5 * r + (5.0 / 3.0) * r2) * gp_exp_nonpos(-s5 * r);
}
__device__ __forceinline__ void gp_k_and_g(int kernel, double variance, double r2, double &k, double &g) {
    if (kernel == 0) {
        k = variance * gp_exp_nonpos(-0.5 * r2);
        g = -k;  // dK_dr = -r k
    } else {
        const double s5 = 2.23606797749978969640917366873128;
        const double r = sqrt(r2);
        const double e = gp_exp_nonpos(-s5 * r);
        k = variance * (1.0 + s5 * r + (5.0 / 3.0) * r2) * e;
        g = -(5.0 / 3.0) * variance * (1.0 + s5 * r) * e;  // (10/3 r - 5 r - 5 sqrt5/3 r^2) e / r
    }
}

// divisor used when staging inputs for a covariance evaluation
__host__ __device__ static inline double kp_div(const KernParams &kp, int d) { return kp.gower ? kp.gdiv[d] : kp.ls[d]; }

// ---- tile-set descriptor for the GEMM family ---------------------------------
// Enumerates output tiles (i, c): c in [c0, c1); rows i in [tri ? c : r0, r1).
struct TileSet {
    int r0, r1, c0, c1, tri;
};
static inline long tileset_count(const TileSet &t) {
    if (!t.tri) return (long)(t.r1 - t.r0) * (t.c1 - t.c0);
    long n = 0;
    for (int c = t.c0; c < t.c1; ++c) n += (t.r1 - c) > 0 ? (t.r1 - c) : 0;
    return n;
}

// C tile (i,c) at C + i*128*ldc + c*128
// A rows   at A + i*128*lda            (K columns)
// B rows   at B + (c*b_mul)*128*ldb    (K columns)
// mode 0: C = A B^T ; mode 1: C -= A B^T
struct GemmOpt {
    int k_tri = 0;      // k starts at the output row tile (A block upper-triangular) ...
    int k_sub = 0;      // ... minus k_sub tiles (the A / B pointers already sit at column tile k_sub)
    int k_end_tri = 0;  // k ends after column tile (tc - b_sub) (B block lower-triangular)
    int b_sub = 0;      // B row tile = (tc - b_sub) * b_mul
    int batch = 1;      // blockIdx.y instances with pointer strides sC, sA, sB (elements)
    long sC = 0, sA = 0, sB = 0;
    const short *tile_list = nullptr;  // device pointer, 2 shorts per tile, tileset_count(ts) tiles
    int stagger = 0;                   // odd-wave-slot workgroups start stagger * 1024 cycles late
    int small = 0;                     // 64x64 workgroup tiles (4 workgroups per 128-tile): latency-bound launches
    int waves8 = 0;                    // 128x128 tile on 8 waves (64x32 per wave, 4 waves/SIMD) instead of 4
    int inplace = 0;                   // C aliases A (tile-local product): the 128-tile must stay in one workgroup
    int rows64 = 0;                    // 64-row strips of the 128-tile (two workgroups per tile; safe in place)
    int pair = 0;                      // k_end_tri, rectangular tile set, small tiles: column tiles paired for equal K
};
// Host-side construction of an L2-friendly order: the tile set is cut into S x S super-tiles; the
// list is dealt so that each XCD's contiguous run (see the remap in gemm.hip) walks whole super-tiles.
std::vector<short> build_tile_list(const TileSet &ts, int S);
void launch_gemm_nt(hipStream_t s, int mode, double *C, long ldc, const double *A, long lda,
                    const double *B, long ldb, int b_mul, int K, TileSet ts, const GemmOpt &o = GemmOpt());

// Factor the 128x128 diagonal tile t of A (row-major, lda) in place (lower) and write
// its inverse (row-major 128x128, lower, zero above) to invL + t*128*128.
// info: device int, 0 = ok, else 1-based global column of the first non-positive pivot.
void launch_potrf_tile(hipStream_t s, double *A, long lda, int t, double *invL, int *info);
// The same for the diagonal tiles t and t + 1 in one launch, including L10 = A10 inv(L00)^T and A11 -= L10 L10^T between them.
void launch_potrf_pair(hipStream_t s, double *A, long lda, int t, double *invL, int *info);
// Both tile columns of the rows below a factored pair, one launch: for every 32-row strip of the row tiles [r0, r1)
//   X0 = A[., t] inv(L_tt)^T;  A[., t+1] -= X0 L[t+1, t]^T;  X1 = A[., t+1] inv(L_t+1,t+1)^T     (in place)
void launch_trsm2(hipStream_t s, double *A, long lda, int t, const double *invL, int r0, int r1);
void potrf_set_debug_lds(int bytes);   // test hook: dynamic LDS added to every diagonal-tile launch (a refused launch beyond ~9 KB)

// Ky lower tiles (incl. diagonal tiles in full) from X; padding rows get identity.
void launch_kbuild(hipStream_t s, double *A, long lda, const double *X, long N, long Npad,
                   const KernParams &kp, double diag_add, int full);
// RHS rows: A[(Npad + p)*lda + i] = Y[i*P + p] (zero for i >= N, and rows p >= P zero)
void launch_set_rhs(hipStream_t s, double *A, long lda, const double *Y, long N, long Npad, int P);
// T[c*ldt + i] = k(xs_c, x_i) for c < M, i < N; zero in the padding.
void launch_cross_k(hipStream_t s, double *T, long ldt, const double *Xs, long M, long Mpad,
                    const double *X, long N, long Npad, const KernParams &kp);

// the same for M <= 8 candidate rows only (rows 0 .. M-1 of T, columns 0 .. Npad-1; no padding rows): the small-M path
void launch_cross_k_rows(hipStream_t s, double *T, long ldt, const double *Xs, int M, const double *X, long N, long Npad,
                         const KernParams &kp);

// logdet = 2 * sum_i log A[i*lda+i], i < N (deterministic single-block reduction)
void launch_logdet(hipStream_t s, const double *A, long lda, long N, double *out);
// out[p] = sum_i z_p[i]^2 for the RHS rows z_p = A[(Npad+p)*lda + i]
void launch_rhs_sumsq(hipStream_t s, const double *A, long lda, long N, long Npad, int P, double *out);

// Backward substitution alpha = L^-T z using the inverse diagonal tiles.
//   z rows: Z + p*ldz (p < P), alpha rows: Aout + p*ldz; w: workspace P*Npad; invP: inverted diagonal panels.
void launch_trsv_backward(hipStream_t s, const double *L, long lda, const double *invP, int W, long Npad,
                          const double *Z, long ldz, int P, double *Aout, double *w);

// Row reductions over the solved candidate rows T[c, 0:N]:
//   var[c] = kss - sum_i T[c,i]^2 (+ noise_add), mean[c*P+p] = sum_i T[c,i] * Z[p*ldz + i]
void launch_predict_reduce(hipStream_t s, const double *T, long ldt, long M, long N, const double *Z,
                           long ldz, int P, double kss, double noise_add, double *mean, double *var);

// mu[i] = sum_j k(x_i, x_j) alpha[j]  (posterior mean at the training inputs, K generated on the fly)
void launch_train_mean(hipStream_t s, const double *X, long N, const KernParams &kp, const double *alpha,
                       double *mu);
// the same vector as y - d alpha (d = total diagonal added to K), O(N)
void launch_train_mean_identity(hipStream_t s, const double *Y, const double *alpha, double d, long N, double *mu);
// deterministic min / argbest reductions
void launch_min(hipStream_t s, const double *v, long n, double *out);

// acquisition values (negated) from mean/var; optional argbest
void launch_acq(hipStream_t s, int type, double par, double fmin, double y_mean, double y_std,
                const double *mean, const double *var, long M, double *out);
void launch_argbest(hipStream_t s, const double *v, long n, int sense, double *best_val, long long *best_idx,
                    double *scratch_val, long long *scratch_idx);

// ---- smallm.hip: a handful of candidate rows as matrix-vector work (the acquisition optimiser's one-row calls) ----------
// S[m, :] = T[m, :] L^-T for m < M by forward substitution over the panels (T consumed), inverted diagonal panels invP (W tiles)
void launch_small_forward_solve(hipStream_t s, const double *L, long lda, const double *invP, int W, long Npad, double *T,
                                double *S, long ldt, int M);
// beta[m, :] = Kx[m, :] Wi for m < M (Wi symmetric, Npad x Npad)
void launch_small_wi_product(hipStream_t s, const double *Wi, long ldw, long Npad, const double *Kx, long ldk, int M,
                             double *beta, long ldb);

// ---- onerow.hip: the acquisition optimiser's one-row calls as three launches over the explicit inverse factor ------------------
#define ROWS_MAX_M 4       // locations per pass of the fused path
#define ROWS_MAX_XS 128    // ... with M * D <= ROWS_MAX_XS doubles travelling in the kernel arguments
struct RowsX {
    int M;
    double xs[ROWS_MAX_XS];   // [M][D] row-major, as the caller gave them
};
struct RowsAcq {
    int on;                   // 0: posterior only
    int type;                 // GP_ACQ_*
    double par, fmin, y_mean, y_std;
    int lp, transform, nb;    // local penalisation (LP.py): on / log transform / batch size
    const double *Xb, *r0, *s0;
};
#define ROWS_OUT_DOUBLES (3 * ROWS_MAX_M * (1 + GP_MAX_D))   // result block; one more double behind it carries the call's ticket
struct RowsWork {             // device scratch (api_rows.hip sizes it): every partial has one writer
    double *wpart, *bpart, *meanpart, *vpart, *gpart;
    unsigned int *counter;    // arrival counter of the finishing kernels: never reset between calls, each pass counts from its own base
    unsigned int counter_base;
    double ticket;            // written behind the results by the workgroup that finishes the call: the host checks it (a launch that
                              // did not complete must not leave the previous call's numbers in the block)
};
long rows_tiles(int nt);
int rows_block_height(int nt);   // rows of the tile per workgroup: 32 for matrices of a few tiles, else 128
size_t rows_gpart_elems(long N);
// Workgroups of the pass's last launch -- the arrivals its counter waits for; the host adds the same number to the counter's base
// after every pass (api_rows.hip rows_wait), so launcher and host take it from here.
inline unsigned rows_finish_grid(long N) { return (unsigned)((N + 63) / 64); }       // rows_finish_kernel: 64 training rows per workgroup
inline unsigned rows_mean_grad_grid(long N) { return (unsigned)((N + 255) / 256); }  // rows_mean_grad_kernel: one row per thread
// results (host-visible block of 3 MV (1 + D) doubles, MV = 1 for M = 1 else ROWS_MAX_M):
//   [mean MV][var MV][acq MV][dmdx MV D][dvdx MV D][dacq MV D]
void launch_rows(hipStream_t s, const double *Li, long Npad, const RowsX &rx, const KernParams &kp, const double *X, long N,
                 const double *alpha, int want_grad, double kss, double noise_add, const RowsAcq &aq, const RowsWork &w,
                 double *out, int nt_loads);
// the mean's gradient alone: dmdx [M, D] at out + 3 MV (one pass over the training points, no inverse factor)
void launch_rows_mean_grad(hipStream_t s, const RowsX &rx, const KernParams &kp, const double *X, long N, const double *alpha,
                           const RowsWork &w, double *out);
void launch_transpose_tri(hipStream_t s, double *dst, const double *src, long n, int mode);

// ---- grad.hip ---------------------------------------------------------------------------------
#define GP_GRAD_CH 16
#define GP_GRAD_NACC (GP_GRAD_CH + 2)
void launch_set_identity(hipStream_t s, double *T, long ld, long n);
void launch_symmetrize(hipStream_t s, double *A, long ld, long n);
// A = scale * lower(A), mirrored into the upper triangle
void launch_symmetrize_scale(hipStream_t s, double *A, long ld, long n, double scale);
void launch_lml_grad(hipStream_t s, const double *X, long N, long Npad, const KernParams &kp, int ard, int d0,
                     const double *alpha, int P, const double *Wi, long ldw, double *partial, double *out);
void launch_predict_grad(hipStream_t s, const double *Xs, long M, const double *X, long N, const KernParams &kp,
                         const double *alpha, long lda_, int P, const double *beta, long ldb, double *dmdx,
                         double *dvdx);
void launch_acq_grad(hipStream_t s, int type, double par, double fmin, double y_mean, double y_std, const double *mean,
                     const double *var, const double *dmdx, const double *dvdx, long M, int D, double *out,
                     double *dout);
void launch_add_diag(hipStream_t s, double *A, long lda, long N, double v);
void launch_trace(hipStream_t s, const double *A, long lda, long N, double *out);   // out[0] = trace, out[1] = smallest diagonal entry

// nb blocks of n x n (row-major, leading dimension n, stacked): identity / transpose (dst_b = src_b^T)
void launch_set_identity_blocks(hipStream_t s, double *T, long n, int nb);
void launch_transpose_blocks(hipStream_t s, double *dst, const double *src, long n, int nb);
void launch_lp(hipStream_t s, const double *negacq, const double *Xs, long M, int D, const double *Xb, int nb,
               const double *r0, const double *s0, int transform, double *out);
void launch_lp_grad(hipStream_t s, double *negacq, double *dneg, const double *Xs, long M, int D, const double *Xb, int nb,
                    const double *r0, const double *s0, int transform);
void launch_mask(hipStream_t s, double *v, const long long *idx, int n, double fill);
// out[i, j] = 0.5 (sum_p alpha_p[i] alpha_p[j] - P Wi[i, j]),  i, j < N  (exact_gaussian_inference.py:70)
void launch_dldk(hipStream_t s, double *out, long ldo, const double *alpha, long lda_, int P, const double *Wi,
                 long ldw, long N);
// zero the strict upper triangle of the nt diagonal 128-tiles (the factor's tiles keep the symmetric input there)
void launch_zero_upper_diag(hipStream_t s, double *A, long lda, int nt);

// ---- rns.hip: fp64-equivalent contraction on the int8 matrix cores (option "emulate_fp64") -----------------------------
#define GP_RNS_T 14
#define GP_RNS_KMAX 8192   // longest contraction (bytes) one residue launch may take: see rns_reduce_f in rns.hip
int rns_init_constants(int device);
void launch_rns_convert(hipStream_t s, const double *src, long ld, long rows, long cols, signed char *dst,
                        long plane_stride, long ldd, double scale, int *flag);
void rns_set_interleave(int v);
void launch_rns_gemm256(hipStream_t s, const signed char *A, long lda, long a_plane, const signed char *B, long ldb,
                        long b_plane, signed char *R, int mt_all, int nt_all, int mt, int c0, int c1, int K,
                        int first, int tri = 0);
void launch_rns_reconstruct256(hipStream_t s, const signed char *R, int mt_all, int nt_all, int mt, int c0_128, int c1_128,
                               long rows, double *T, long ldt, double scale_2e, int tri = 0);

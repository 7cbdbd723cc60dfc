// Panel solve for TWO tile columns in one launch (the rows below a diagonal pair factored by potrf_pair_kernel).
//
// Reference step: the dtrsm inside LAPACK dpotrf (call site GPy/GPy/util/linalg.py:58), re-expressed as products with the
// inverted diagonal tiles like every other triangular solve here.  For the row tiles i in [r0, r1), with t, t+1 the pair:
//     X0 = A[i, t]   inv(L_tt)^T
//     A[i, t+1]     -= X0 L[t+1, t]^T
//     X1 = A[i, t+1] inv(L_t+1,t+1)^T                    (X0, X1 overwrite A[i, t], A[i, t+1])
// A row of the panel depends on no other row, so a workgroup takes a 32-row strip through all three products one after the
// other: the strip's intermediate results go to HBM and come back through this CU's own L1 (coherent within a workgroup), with
// a workgroup barrier -- which waits for the stores -- between two products.  One launch where the 128-column step needs
// two solve launches and one narrow update launch in between; the strips are as fine as the in-place solve's (4 per tile), so a
// few row tiles still spread over many CUs.
//
// Workgroup: 256 threads = 4 waves as 2 x 2 over the 32 x 128 strip (16 rows x 64 columns per wave: 4 accumulators of
// v_mfma_f64_16x16x4_f64), K = 128 in 8 stages of 16, operands staged global -> registers -> LDS (18-double pitch, double
// buffered) exactly as in gemm.hip; these launches are latency-bound (a strip is 3 x 1 MFLOP), the loop is the plain one.
#include "gphip_internal.h"

#define T2_BK 16
#define T2_LSTR 18

// MODE 0: C = A B^T, 1: C -= A B^T;  A: ROWS x 128 (lda), B: 128 x 128 (ldb), C: ROWS x 128 (ldc).  ROWS = 32: waves 2 x 2
// (16 rows x 64 columns each); ROWS = 16: waves 1 x 4 (16 rows x 32 columns each) -- twice the workgroups per tile for the launches
// with few row tiles, where the strip's own latency is all there is.
template <int MODE, int ROWS>
__device__ __forceinline__ void strip_product(const double *Ag, long lda, const double *Bg, long ldb, double *Cg, long ldc,
                                              double *smem, int tid) {
    constexpr int SBUF = (ROWS + GP_TILE) * T2_LSTR;
    constexpr int WN = ROWS == 32 ? 2 : 4;       // waves along the columns
    constexpr int NT = GP_TILE / WN / 16;        // accumulators per wave
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, lg = lane >> 4;
    const int srow = tid >> 3, sch = (tid & 7) * 2;           // staging: row tid/8 (+32 q), 16-byte chunk tid%8
    const bool a_loader = srow < ROWS;
    const double *ap = Ag + (long)(a_loader ? srow : 0) * lda + sch;
    const double *bp = Bg + (long)srow * ldb + sch;
    const long b32 = 32 * ldb;
    const int soff = srow * T2_LSTR + sch;
    double *Cw = Cg + (long)(wm * 16) * ldc + wn * (GP_TILE / WN);
    double4_t acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = MODE == 1 ? Cw[(long)(4 * r + lg) * ldc + 16 * n + li] : 0.0;
    double2_t ra, rb[4];
    ra = *(const double2_t *)ap;
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = *(const double2_t *)(bp + q * b32);
    if (a_loader) *(double2_t *)(smem + soff) = ra;
#pragma unroll
    for (int q = 0; q < 4; ++q) *(double2_t *)(smem + ROWS * T2_LSTR + soff + q * 32 * T2_LSTR) = rb[q];
    __syncthreads();
    const int aoff = (wm * 16 + li) * T2_LSTR + lg * 4;
    const int boff = ROWS * T2_LSTR + (wn * (GP_TILE / WN) + li) * T2_LSTR + lg * 4;
    constexpr int NK = GP_TILE / T2_BK;
#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < NK;
        if (more) {
            ap += T2_BK;
            bp += T2_BK;
            ra = *(const double2_t *)ap;
#pragma unroll
            for (int q = 0; q < 4; ++q) rb[q] = *(const double2_t *)(bp + q * b32);
        }
        const double *as = smem + buf * SBUF + aoff, *bs = smem + buf * SBUF + boff;
#pragma unroll
        for (int h = 0; h < 2; ++h) {            // lane group lg owns k = 4 lg .. 4 lg + 3 of the stage: two 16-byte reads per operand row
            const double2_t af = *(const double2_t *)(as + 2 * h);
            double2_t bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const double2_t *)(bs + n * 16 * T2_LSTR + 2 * h);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[e], bf[n][e], acc[n], 0, 0, MODE == 1 ? 1 : 0);
        }
        if (more) {
            double *ns = smem + (buf ^ 1) * SBUF;
            if (a_loader) *(double2_t *)(ns + soff) = ra;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(double2_t *)(ns + ROWS * T2_LSTR + soff + q * 32 * T2_LSTR) = rb[q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(long)(4 * r + lg) * ldc + 16 * n + li] = acc[n][r];
}

template <int ROWS>
__global__ __launch_bounds__(256) void trsm2_kernel(double *A, long lda, int t, const double *invL, int r0) {
    __shared__ __attribute__((aligned(16))) double smem[2 * (ROWS + GP_TILE) * T2_LSTR];
    constexpr int SPT = GP_TILE / ROWS;   // strips per row tile
    const int tid = threadIdx.x;
    const long row = ((long)r0 + (blockIdx.x / SPT)) * GP_TILE + (blockIdx.x % SPT) * ROWS;   // first row of this strip
    double *S0 = A + row * lda + (long)t * GP_TILE;        // the strip's columns of tile t ...
    double *S1 = S0 + GP_TILE;                              // ... and of tile t + 1
    const double *inv0 = invL + (long)t * GP_TILE * GP_TILE, *inv1 = inv0 + (long)GP_TILE * GP_TILE;
    const double *L10 = A + (long)(t + 1) * GP_TILE * lda + (long)t * GP_TILE;
    strip_product<0, ROWS>(S0, lda, inv0, GP_TILE, S0, lda, smem, tid);       // X0 = A0 inv00^T
    __syncthreads();   // (workgroup release / acquire: X0 is in memory before any wave stages it as an operand)
    strip_product<1, ROWS>(S0, lda, L10, lda, S1, lda, smem, tid);            // A1 -= X0 L10^T
    __syncthreads();
    strip_product<0, ROWS>(S1, lda, inv1, GP_TILE, S1, lda, smem, tid);       // X1 = A1 inv11^T
}

void launch_trsm2(hipStream_t s, double *A, long lda, int t, const double *invL, int r0, int r1) {
    if (r1 <= r0) return;
    // few row tiles: 16-row strips (8 workgroups per tile) -- the launch is as long as ONE strip's three products
    if (r1 - r0 <= 24)
        GP_LAUNCH((trsm2_kernel<16>), dim3((unsigned)(8 * (r1 - r0))), dim3(256), 0, s, A, lda, t, invL, r0);
    else
        GP_LAUNCH((trsm2_kernel<32>), dim3((unsigned)(4 * (r1 - r0))), dim3(256), 0, s, A, lda, t, invL, r0);
}

// Diagonal-tile Cholesky + triangular inverse for gfx950 (one workgroup per 128x128 tile).
//
// Replaces the unblocked part of LAPACK dpotrf (reference call site GPy/GPy/util/linalg.py:58)
// and supplies L11^-1 so that every panel / candidate triangular solve above it becomes a
// plain product on the MFMA GEMM kernel (dtrtrs call site posterior.py:294, dpotrs
// exact_gaussian_inference.py:60).
//
// The tile lives in LDS (pitch 130 doubles: the MFMA operand pattern row=lane&15, k=lane>>4 read
// with ds_read_b64 is conflict free since 130 = 2 mod 32).  Factorisation is right-looking over
// eight 16-column micro panels:
//   (1) wave 0 factors the 16x16 diagonal micro block entirely in registers (lane (i, g) owns
//       columns g, g+4, g+8, g+12 of row i; pivots/columns move by readlane / ds_bpermute; L D L^T
//       elimination so that only a reciprocal sits between consecutive pivots) and inverts it by
//       forward substitution (column per lane) -- while the other seven waves apply the previous
//       micro panel's trailing update (one micro block of look-ahead inside the tile);
//   (2) the micro panel below is multiplied by that inverse (4 x v_mfma_f64_16x16x4_f64 per block);
//   (3) the trailing 16x16 blocks take a rank-16 update (4 MFMAs each, A negated via the f64
//       MFMA neg modifier so the old block value rides in as the C operand).
// The tile inverse is then built block column by block column (one wave per column):
//   Inv[i][j] = -Dinv[i] * sum_{k=j}^{i-1} L[i][k] Inv[k][j], the inner sum staying in the
// accumulator and re-entering the next MFMA directly as its B operand (accumulator element s of
// lane (n, g) is row 4s+g, column n -- exactly the B fragment of k-step s).  Inverse blocks are
// parked transposed in the unused upper triangle of the LDS tile.
#include "gphip_internal.h"

#ifdef POTRF_STAMPS
// diagnostic build only (tools/micro/potrf_bench.hip): cycle stamps of wave 0 into a buffer of their own
__device__ unsigned long long g_potrf_stamps[64];
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        if (threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            g_potrf_stamps[i] = t_;                                                               \
        }                                                                                         \
    } while (0)
#else
#define STAMP(i)
#endif

#define TS 130   // LDS pitch of the tile (doubles)
#define DS 18    // LDS pitch of a 16x16 inverse micro block
#define DBLK (16 * DS)

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// 1/a from the hardware seed v_rcp_f64 and two Newton steps (4 dependent FMAs).
__device__ __forceinline__ double fast_rcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    double e = fma(-a, x, 1.0);
    x = fma(x, e, x);
    e = fma(-a, x, 1.0);
    x = fma(x, e, x);
    return x;
}
// sqrt(a) and 1/sqrt(a) from v_rsq_f64 and two coupled Newton (Goldschmidt) steps; within 1-2 ulp.
__device__ __forceinline__ void sqrt_rsqrt(double a, double &sq, double &rsq) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, a);  // residual correction of the root
    sq = fma(d, h, g);
    rsq = 2.0 * h;
}

// Factor the 16x16 micro block p of T in place and write its inverse to Dinv[p].  One wave.
// Lane (i = lane&15, g = lane>>4) owns columns g, g+4, g+8, g+12 of row i.
//
// The elimination runs in L D L^T form: at step k the trailing entries take
//   a_ij -= a_ik a_jk / a_kk
// with the UNSCALED column k, so the only arithmetic between two consecutive pivots is one reciprocal
// (seed + 2 Newton steps) and one FMA; the cross-lane fetches of a_ik, a_jk (ds_bpermute) are issued before
// the reciprocal is ready, and the square roots that turn L' D^1/2 into the Cholesky factor are taken
// afterwards, all 16 at once, off the chain.  Returns the first failing local column (0..15) or -1.
__device__ __forceinline__ int potrf16_inv16(double *T, double *Dinv, int p, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    double *blk = T + (p * 16 + li) * TS + p * 16;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = blk[lg + 4 * q];
    int fail = -1;
    double piv_own = 1.0;  // pivot d_i of this lane's row
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int kq = k >> 2, kg = k & 3;
        const double aik = __shfl(v[kq], li + 16 * kg);
        double t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = aik * __shfl(v[kq], lg + 4 * q + 16 * kg);
        const double akk = readlane_d(v[kq], k + 16 * kg);
        if (!(akk > 0.0) && fail < 0) fail = k;
        if (li == k) piv_own = akk;
        const double rk = fast_rcp(akk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lg + 4 * q;
            if (j > k && li >= j) v[q] = fma(-t[q], rk, v[q]);
        }
    }
    {
        const double a15 = readlane_d(v[3], 15 + 16 * 3);
        if (!(a15 > 0.0) && fail < 0) fail = 15;
        if (li == 15) piv_own = a15;
    }
    // scale: L[i][j] = a_ij / sqrt(d_j) (j < i), L[i][i] = sqrt(d_i)
    double sq_own, rs_own;
    sqrt_rsqrt(piv_own, sq_own, rs_own);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lg + 4 * q;
        const double rsj = __shfl(rs_own, j);  // lane j (g = 0) holds row j's pivot
        v[q] = (j < li) ? v[q] * rsj : ((j == li) ? sq_own : 0.0);
        blk[j] = v[q];
    }
    // inverse by forward substitution, column li per lane; L[i][k] comes back as an LDS broadcast read
    // (same address in every lane) of what this wave just stored -- DS operations of one wave execute in order.
    // Two partial sums per row halve the dependent FMA chain.
    const double *row = T + (p * 16) * TS + p * 16;
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s0 = (i == li) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            if (k & 1) s1 = fma(-row[i * TS + k], m[k], s1);
            else s0 = fma(-row[i * TS + k], m[k], s0);
        }
        const double rii = __shfl(rs_own, i);
        m[i] = (s0 + s1) * rii;
    }
    if (lg == 0) {
        double *dp = Dinv + p * DBLK + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i * DS] = m[i];
    }
    return fail;
}

// one 16x16 block product on the matrix pipe: acc (+)= sum_k A[i][k] B[j][k], operands in LDS with pitches
// (pa, pb); neg selects acc - A B^T through the f64 MFMA's neg-A modifier.
template <int NEG>
__device__ __forceinline__ double4_t mm16(const double *ap, const double *bp, int sb, double4_t acc) {
    const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
    const double b0 = bp[0], b1 = bp[4 * sb], b2 = bp[8 * sb], b3 = bp[12 * sb];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, NEG);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, NEG);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, NEG);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, NEG);
    return acc;
}

// Factor + invert tile t of A (see the header).
// T: GP_TILE * TS doubles of LDS, Dinv: 8 * DBLK doubles of LDS.  512 threads.
__device__ __forceinline__ void potrf_tile_body(double *A, long lda, int t, double *invL, int *info, double *T, double *Dinv) {

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    double *At = A + (long)t * GP_TILE * lda + (long)t * GP_TILE;
    STAMP(0);

    // load the lower part (whole rows: simpler and coalesced), 16 B per lane
    for (int q = 0; q < 16; ++q) {
        const int id = tid + 512 * q;
        const int r = id >> 6, c2 = (id & 63) * 2;
        const double2_t v = *(const double2_t *)(At + (long)r * lda + c2);
        *(double2_t *)(T + r * TS + c2) = v;
    }
    __syncthreads();
    STAMP(1);
    if (wave == 0) {
        const int fail = potrf16_inv16(T, Dinv, 0, lane);
        if (fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + fail + 1);
    }
    __syncthreads();

    // Right-looking over eight 16-column micro panels with one micro block of look-ahead:
    //   stage A  wave 0: row block p+1 of the panel solve;      waves 1..7: row blocks p+2..7
    //   stage B  wave 0: update of block (p+1,p+1), then its factorisation + inverse (the latency chain);
    //            waves 1..7: every other trailing block (i, j), p < j <= i, (i, j) != (p+1, p+1)
    STAMP(2);
    for (int p = 0; p < 7; ++p) {
        STAMP(3 + 3 * p);
        {   // stage A: X = T[rb][p] * Dinv[p]^T
            const int rb = p + 1 + wave;
            if (rb < 8) {
                const double *ap = T + (rb * 16 + li) * TS + p * 16 + lg;
                const double *bp = Dinv + p * DBLK + li * DS + lg;
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                acc = mm16<0>(ap, bp, 1, acc);
                double *cp = T + (rb * 16 + lg) * TS + p * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r];
            }
        }
        __syncthreads();
        STAMP(4 + 3 * p);
        if (wave == 0) {
            const int i = p + 1;
            const double *ap = T + (i * 16 + li) * TS + p * 16 + lg;
            double *cp = T + (i * 16 + lg) * TS + i * 16 + li;
            double4_t acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = cp[(4 * r) * TS];
            acc = mm16<1>(ap, ap, 1, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r];
            const int fail = potrf16_inv16(T, Dinv, i, lane);
            if (fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + i * 16 + fail + 1);
            STAMP(5 + 3 * p);
        } else {
            // blocks (i, j), p+1 <= j <= i <= 7 without (p+1, p+1): enumerate rows i = p+1 .. 7
            const int nb = 7 - p;
            const int cnt = nb * (nb + 1) / 2 - 1;
            for (int e = wave - 1; e < cnt; e += 7) {
                const int e1 = e + 1;  // skip entry 0 = (p+1, p+1)
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= e1) ++ii;
                const int jj = e1 - ii * (ii + 1) / 2;
                const int i = p + 1 + ii, j = p + 1 + jj;
                const double *ap = T + (i * 16 + li) * TS + p * 16 + lg;
                const double *bp = T + (j * 16 + li) * TS + p * 16 + lg;
                double *cp = T + (i * 16 + lg) * TS + j * 16 + li;
                double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = cp[(4 * r) * TS];
                acc = mm16<1>(ap, bp, 1, acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r];
            }
        }
        __syncthreads();
    }

    STAMP(24);
    // ---- tile inverse, block column j = wave; Inv[i][j] parked at T[(j16+n)][(i16+m)] = Inv_ij[m][n]
    for (int i = 1; i < 8; ++i) {
        const int j = wave;
        if (i > j) {
            double4_t P = {0.0, 0.0, 0.0, 0.0};
            {   // k = j term: L[i][j] * Dinv[j]
                const double *ap = T + (i * 16 + li) * TS + j * 16 + lg;
                const double *bp = Dinv + j * DBLK + lg * DS + li;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    P = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * s], bp[(4 * s) * DS], P, 0, 0, 0);
            }
            for (int k = j + 1; k < i; ++k) {
                const double *ap = T + (i * 16 + li) * TS + k * 16 + lg;
                const double *bp = T + (j * 16 + li) * TS + k * 16 + lg;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    P = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * s], bp[4 * s], P, 0, 0, 0);
            }
            // Y = -Dinv[i] * P ; P re-enters as the B operand straight from the accumulator
            double4_t Y = {0.0, 0.0, 0.0, 0.0};
            const double *dp = Dinv + i * DBLK + li * DS + lg;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dp[4 * s], P[s], Y, 0, 0, 1);
            double *cp = T + (j * 16 + li) * TS + i * 16 + lg;
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[4 * r] = Y[r];
        }
        __syncthreads();
    }

    STAMP(25);
    // ---- write back: L (lower incl. diagonal) in place, inverse tile to the workspace
    double *Iv = invL + (long)t * GP_TILE * GP_TILE;
    for (int q = 0; q < 32; ++q) {
        const int id = tid + 512 * q;
        const int r = id >> 7, c = id & 127;
        if (c <= r) At[(long)r * lda + c] = T[r * TS + c];
        double inv;
        const int rb = r >> 4, cb = c >> 4;
        if (rb == cb)
            inv = Dinv[rb * DBLK + (r & 15) * DS + (c & 15)];
        else if (rb > cb)
            inv = T[c * TS + r];
        else
            inv = 0.0;
        Iv[r * GP_TILE + c] = inv;
    }
    STAMP(26);
}


__global__ __launch_bounds__(512) void potrf_tile_kernel(double *A, long lda, int t, double *invL, int *info) {
    __shared__ __attribute__((aligned(16))) double T[GP_TILE * TS];
    __shared__ __attribute__((aligned(16))) double Dinv[8 * DBLK];
    potrf_tile_body(A, lda, t, invL, info, T, Dinv);
}

void launch_potrf_tile(hipStream_t s, double *A, long lda, int t, double *invL, int *info) {
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), 0, s, A, lda, t, invL, info);
}

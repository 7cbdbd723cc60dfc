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
//       columns g, g+4, g+8, g+12 of row i; pivots/columns move by readlane / ds_bpermute) and
//       inverts it by forward substitution (column per lane);
//   (2) the micro panel below is multiplied by that inverse (4 x v_mfma_f64_16x16x4_f64 per block);
//   (3) the trailing 16x16 blocks take a rank-16 update (4 MFMAs each, A negated via the f64
//       MFMA neg modifier so the old block value rides in as the C operand).
// The tile inverse is then built block column by block column (one wave per column):
//   Inv[i][j] = -Dinv[i] * sum_{k=j}^{i-1} L[i][k] Inv[k][j], the inner sum staying in the
// accumulator and re-entering the next MFMA directly as its B operand (accumulator element s of
// lane (n, g) is row 4s+g, column n -- exactly the B fragment of k-step s).  Inverse blocks are
// parked transposed in the unused upper triangle of the LDS tile.
#include "gphip_internal.h"

#define TS 130   // LDS pitch of the tile (doubles)
#define DS 18    // LDS pitch of a 16x16 inverse micro block
#define DBLK (16 * DS)

__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// Factor the 16x16 micro block p of T in place and write its inverse to Dinv[p].  Wave 0 only.
// Returns the first failing local column (0..15) or -1.
__device__ __forceinline__ int potrf16_inv16(double *T, double *Dinv, int p, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    double *blk = T + (p * 16 + li) * TS + p * 16;
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = blk[lg + 4 * q];
    double rinv[16];
    int fail = -1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int kq = k >> 2, kg = k & 3;
        const double akk = readlane_d(v[kq], k + 16 * kg);
        if (!(akk > 0.0) && fail < 0) fail = k;
        const double d = sqrt(akk);
        const double ri = 1.0 / d;
        rinv[k] = ri;
        if (lg == kg) {
            if (li > k) v[kq] = v[kq] * ri;
            else if (li == k) v[kq] = d;
        }
        const double lik = __shfl(v[kq], li + 16 * kg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lg + 4 * q;
            const double ljk = __shfl(v[kq], j + 16 * kg);
            if (j > k && li >= j) v[q] -= lik * ljk;
        }
    }
    // write the factor back (zero above the diagonal of the micro block)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = lg + 4 * q;
        blk[j] = (j <= li) ? v[q] : 0.0;
    }
    // inverse by forward substitution: lane li owns column li of M = L16^-1
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = (i == li) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const double lik = readlane_d(v[k >> 2], i + 16 * (k & 3));
            s -= lik * m[k];
        }
        m[i] = s * rinv[i];
    }
    if (lg == 0) {
        double *dp = Dinv + p * DBLK + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i * DS] = m[i];
    }
    return fail;
}

__global__ __launch_bounds__(512) void potrf_tile_kernel(double *A, long lda, int t, double *invL, int *info) {
    __shared__ __attribute__((aligned(16))) double T[GP_TILE * TS];
    __shared__ __attribute__((aligned(16))) double Dinv[8 * DBLK];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    double *At = A + (long)t * GP_TILE * lda + (long)t * GP_TILE;

    // load the lower part (whole rows: simpler and coalesced), 16 B per lane
    for (int q = 0; q < 16; ++q) {
        const int id = tid + 512 * q;
        const int r = id >> 6, c2 = (id & 63) * 2;
        const double2_t v = *(const double2_t *)(At + (long)r * lda + c2);
        *(double2_t *)(T + r * TS + c2) = v;
    }
    __syncthreads();

    for (int p = 0; p < 8; ++p) {
        if (wave == 0) {
            const int fail = potrf16_inv16(T, Dinv, p, lane);
            if (fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + p * 16 + fail + 1);
        }
        __syncthreads();
        // (2) micro panel: X = T[rb][p] * Dinv[p]^T, one row block per wave
        {
            const int rb = p + 1 + wave;
            if (rb < 8) {
                const double *ap = T + (rb * 16 + li) * TS + p * 16 + lg;
                const double *bp = Dinv + p * DBLK + li * DS + lg;
                double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
                double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
                double *cp = T + (rb * 16 + lg) * TS + p * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r];
            }
        }
        __syncthreads();
        // (3) trailing rank-16 update of blocks (i, j), p < j <= i <= 7
        {
            const int nb = 7 - p;
            const int cnt = nb * (nb + 1) / 2;
            for (int e = wave; e < cnt; e += 8) {
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
                const int jj = e - ii * (ii + 1) / 2;
                const int i = p + 1 + ii, j = p + 1 + jj;
                const double *ap = T + (i * 16 + li) * TS + p * 16 + lg;
                const double *bp = T + (j * 16 + li) * TS + p * 16 + lg;
                double *cp = T + (i * 16 + lg) * TS + j * 16 + li;
                double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = cp[(4 * r) * TS];
                double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
                double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 1);  // neg A: C - A B
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 1);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 1);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r];
            }
        }
        __syncthreads();
    }

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
}

void launch_potrf_tile(hipStream_t s, double *A, long lda, int t, double *invL, int *info) {
    hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(512), 0, s, A, lda, t, invL, info);
}

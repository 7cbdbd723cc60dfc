// Diagonal-tile Cholesky + triangular inverse for gfx950 (one workgroup of eight waves per 128x128 tile).
//
// Replaces the unblocked part of LAPACK dpotrf (reference call site GPy/GPy/util/linalg.py:58)
// and supplies L11^-1 so that every panel / candidate triangular solve above it becomes a
// plain product on the MFMA GEMM kernel (dtrtrs call site posterior.py:294, dpotrs
// exact_gaussian_inference.py:60).
//
// The tile lives in LDS (pitch 130 doubles: the MFMA operand pattern row=lane&15, k=lane>>4 read
// with ds_read_b64 is conflict free since 130 = 2 mod 32).  This kernel is the serial link of the factorisation's
// latency chain (one launch per 128 columns), so it is organised around its own critical path:
//
//  Factorisation, right-looking over eight 16-column micro panels p:
//   E(p)  ELIMINATION IN REGISTERS.  A wave holds, one row per lane, the 16 x 16 diagonal micro block (lanes 0-15) and up
//         to three 16-row blocks of the panel below it (lanes 16-63), all 16 columns of a row in registers.  The 16 pivot
//         steps  a_ij -= (a_ik / a_kk) a_jk  take a_kk and a_jk from the diagonal block's lanes through v_readlane (wave-
//         uniform scalars feeding the FMAs); the only arithmetic between two pivots is one reciprocal (L D L^T order).
//         The rows below the diagonal block come out SOLVED by the same steps -- no inverse of the micro block sits on the
//         chain (round 2 inverted it by substitution inside the chain: 7.4k cycles per micro panel for factor + inverse,
//         then a product per row block).  Columns are scaled by 1/sqrt(d_j) afterwards, all at once.
//   Uc(p) the 16-column block column p+1 takes panel p's rank-16 update on the matrix pipe (one block per wave, 4 MFMAs):
//         all that E(p+1) waits for.
//   Ul(p) WHILE E(p) runs, on waves that do not eliminate, block column p+1 takes the updates of the panels 0 .. p-1 (left-
//         looking for everything off the chain).
//  Inverse by 2 x 2 block recursion, Inv[J][I] = -Inv[J][J] L[J][I] Inv[I][I]: the eight 16 x 16 diagonal inverses by
//  substitution (seven of them beside E(4) .. E(7), the last beside the first level); then levels of 32, 64 and 128 rows, each as two rounds of independent block
//  products spread evenly over the eight waves (the intermediate L[J][I] Inv[I][I] overwrites L[J][I] in LDS: the factor
//  has gone to HBM by then).  Round 2 built it block column by block column: 168 dependent MFMAs on wave 0.
//  What bounds it (cycle stamps, tools/micro/potrf_check): E is bound by one wave's instruction issue (3.6k cycles per micro
//  panel: ~500 VALU instructions, a fifth of them f64); the trailing updates and the inverse by the f64 matrix instructions
//  themselves (about 90 cycles each per SIMD here, 736 of them), which share the SIMD's f64 units with E's FMAs.
//  Loads / stores cover the lower block triangle only (the inverse's upper blocks are zero from the allocation on) and ride
//  beside the arithmetic: block column 0 is loaded first and the rest arrives during E(0); the factor leaves block column
//  by block column during the following micro panel, the inverse level by level.
#include "gphip_internal.h"

#ifdef POTRF_STAMPS
// diagnostic build only (tools/micro/potrf_check.hip): cycle stamps of wave 0 into a buffer of their own
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


// ---- E: one wave eliminates micro panel p --------------------------------------------------------------------------------
// Lane (g = lane >> 4, i = lane & 15): g = 0 holds row i of the diagonal micro block p, g = 1..3 row i of the row blocks
// rb0 .. rb0+nrb-1 below it (lanes of missing blocks read a row of zeros at `zero` and write nothing).  Returns the first
// failing local column (0..15) or -1.  Right of its diagonal the diagonal block is left with meaningless values: nothing
// reads them (the factor's store, the substitution for its inverse and every block product stay on or below the diagonal).
__device__ __forceinline__ int eliminate_panel(double *T, const double *zero, int p, int rb0, int nrb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const bool valid = (g == 0) || (g <= nrb);
    const int rb = (g == 0) ? p : rb0 + g - 1;
    double *rowp = T + (rb * 16 + i) * TS + p * 16;
    const double *src = valid ? rowp : zero;
    double a[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double2_t v = *(const double2_t *)(src + 2 * q);
        a[2 * q] = v[0];
        a[2 * q + 1] = v[1];
    }
    double piv = 1.0;   // d_i collects in lane i (1.0 elsewhere)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double akk = readlane_d(a[k], k);
        const double rk = fast_rcp(akk);
        // a positive pivot too small to invert (subnormal: v_rcp_f64 overflows and the Newton steps give NaN) counts as a failed
        // pivot, like a non-positive one: the tile then goes through the jitter ladder instead of coming back as NaN with info 0
        if (lane == k) piv = (fabs(rk) < __builtin_inf()) ? akk : -1.0;
        const double m = a[k] * rk;                 // a_ik / a_kk
        // a_jk (row j of the diagonal block) through v_readlane, four at a time: the scalar results of one group are
        // consumed by the FMAs while the next group's reads issue (a VALU read of a just-written SGPR costs wait states)
#pragma unroll
        for (int j0 = k + 1; j0 < 16; j0 += 4) {
            double s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = (j0 + u < 16) ? readlane_d(a[k], j0 + u) : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u < 16) a[j0 + u] = fma(-m, s[u], a[j0 + u]);
        }
    }
    // (What bounds these 16 steps is the pivot-to-pivot chain -- update of column k+1, v_readlane, reciprocal seed, its
    // correction, the multiplier: six dependent f64 operations at ~22 cycles each, 2.1k cycles per micro panel -- not the
    // instruction count: a version with the column updates as v_fmac_f64_dpp row_newbcast (one instruction per (j, k) instead of
    // three, the diagonal block's column copied to every 16-lane row by ds_bpermute), software-pipelined and with a third-order
    // reciprocal correction, was correct and 0.1k cycles faster per micro panel; without any column update at all the steps
    // still take 3.0k.  The plain form stays.)
    const unsigned long long bad = __ballot(!(piv > 0.0)) & 0xFFFFull;
    const int fail = bad ? (int)__builtin_ctzll(bad) : -1;
    double sq, rs;
    sqrt_rsqrt(piv, sq, rs);
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] *= readlane_d(rs, j);   // L[i][j] = a_ij / sqrt(d_j)
    if (valid) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double2_t v;
            v[0] = a[2 * q];
            v[1] = a[2 * q + 1];
            *(double2_t *)(rowp + 2 * q) = v;
        }
        if (g == 0) rowp[i] = sq;   // L[i][i] = sqrt(d_i): lands after the row (one wave's LDS operations execute in order)
    }
    return fail;
}

// trailing block (i, j) -= X_i X_j^T with the solved micro panel p (rank 16: 4 MFMAs, as two independent pairs so that
// the dependent chain on the factorisation's critical path is two matrix instructions, not four)
__device__ __forceinline__ void update_block(double *T, int p, int i, int j, int li, int lg) {
    const double *ap = T + (i * 16 + li) * TS + p * 16 + lg;
    const double *bp = T + (j * 16 + li) * TS + p * 16 + lg;
    double *cp = T + (i * 16 + lg) * TS + j * 16 + li;
    const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
    const double b0 = bp[0], b1 = bp[4], b2 = bp[8], b3 = bp[12];
    double4_t acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cp[(4 * r) * TS];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 1);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc2, 0, 0, 1);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 1);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc2, 0, 0, 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r] + acc2[r];
}

// block (i, c) -= sum over the solved micro panels 0 .. np-1 of X_i X_c^T (left-looking: one read and one write of the block for
// np rank-16 updates; two accumulation chains)
__device__ __forceinline__ void update_block_panels(double *T, int np, int i, int c, int li, int lg) {
    const double *ap = T + (i * 16 + li) * TS + lg;
    const double *bp = T + (c * 16 + li) * TS + lg;
    double *cp = T + (i * 16 + lg) * TS + c * 16 + li;
    double4_t acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cp[(4 * r) * TS];
    for (int q = 0; q < np; ++q) {
        const double a0 = ap[q * 16], a1 = ap[q * 16 + 4], a2 = ap[q * 16 + 8], a3 = ap[q * 16 + 12];
        const double b0 = bp[q * 16], b1 = bp[q * 16 + 4], b2 = bp[q * 16 + 8], b3 = bp[q * 16 + 12];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 1);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc2, 0, 0, 1);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 1);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc2, 0, 0, 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[(4 * r) * TS] = acc[r] + acc2[r];
}

// ---- operand fetch / result store of the inverse's block products (element (row, col) of a 16 x 16 block) ------------------
//   A operand of MFMA step s: A[li][lg + 4s];  B operand: B[lg + 4s][li];  result r of lane (li, lg): D[4r + lg][li]
struct Frag { double v[4]; };
__device__ __forceinline__ Frag lda_rowmajor(const double *base, int pitch, int li, int lg) {    // A[m][k] at base + m pitch + k
    Frag f; const double *q = base + li * pitch + lg;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = q[4 * s];
    return f;
}
__device__ __forceinline__ Frag lda_transposed(const double *base, int pitch, int li, int lg) {  // A[m][k] at base + k pitch + m
    Frag f; const double *q = base + lg * pitch + li;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = q[(4 * s) * pitch];
    return f;
}
__device__ __forceinline__ Frag ldb_rowmajor(const double *base, int pitch, int li, int lg) {    // B[k][n] at base + k pitch + n
    Frag f; const double *q = base + lg * pitch + li;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = q[(4 * s) * pitch];
    return f;
}
__device__ __forceinline__ Frag ldb_transposed(const double *base, int pitch, int li, int lg) {  // B[k][n] at base + n pitch + k
    Frag f; const double *q = base + li * pitch + lg;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = q[4 * s];
    return f;
}
template <int NEG>
__device__ __forceinline__ double4_t mma(const Frag &a, const Frag &b, double4_t acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[s], b.v[s], acc, 0, 0, NEG);
    return acc;
}
__device__ __forceinline__ Frag acc_as_b(const double4_t &P) {   // an accumulator re-enters as the B operand: P[4s + lg][li]
    Frag f;
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = P[s];
    return f;
}
__device__ __forceinline__ void st_rowmajor(double *base, int pitch, int li, int lg, const double4_t &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) base[(4 * r + lg) * pitch + li] = acc[r];
}
__device__ __forceinline__ void st_transposed(double *base, int pitch, int li, int lg, const double4_t &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) base[li * pitch + 4 * r + lg] = acc[r];
}
// Where things live during the inverse:  L[i][j] (i >= j) row-major in T's lower blocks;  Dinv[b]: the inverse of
// diagonal micro block b, row-major, pitch DS;  Inv[i][j] (i > j) PARKED TRANSPOSED in T's upper block (j, i):
// T[(16 j + n) TS + 16 i + m] = Inv_ij[m][n];  the intermediate P[k][c] = (L[J][I] Inv[I][I])_kc overwrites L[k][c].
#define TBLK(i, j) (T + ((i) * 16) * TS + (j) * 16)

// Inverse of the 16x16 diagonal micro block p (forward substitution, column li per lane; L[i][k] comes back as LDS broadcast
// reads; two partial sums per row halve the dependent FMA chain).  One wave; lanes 0-15 write.
__device__ __forceinline__ void inv16(const double *T, double *Dinv, int p, int lane) {
    const int li = lane & 15, lg = lane >> 4;
    const double *row = T + (p * 16) * TS + p * 16;
    const double rinv = fast_rcp(row[li * TS + li]);   // 1 / L_ii of this lane's row
    double m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s0 = (i == li) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
            if (k & 1) s1 = fma(-row[i * TS + k], m[k], s1);
            else s0 = fma(-row[i * TS + k], m[k], s0);
        }
        m[i] = (s0 + s1) * readlane_d(rinv, i);
    }
    if (lg == 0) {
        double *dp = Dinv + p * DBLK + li;
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i * DS] = m[i];
    }
}

// the inverse's finished blocks go to the workspace as soon as a level is complete (stores overlap the next level's
// products).  nb blocks of 16 x 16, 128 16-byte chunks each; block b of a level sits at (rb, cb):
//   level 0: the diagonal micro blocks (b, b), from Dinv;  level 1: (2b+1, 2b);  level 2: (4Q+2+((b>>1)&1), 4Q+(b&1)),
//   Q = b>>2;  level 3: (4 + (b>>2), b&3) -- off-diagonal blocks are read back from where they are parked (transposed).
template <int LEVEL>
__device__ __forceinline__ void store_inverse_level(double *Iv, const double *T, const double *Dinv, int tid) {
    constexpr int nb = LEVEL == 0 ? 8 : (LEVEL == 1 ? 4 : (LEVEL == 2 ? 8 : 16));
#pragma unroll
    for (int q = 0; q < (nb * 128 + 511) / 512; ++q) {
        const int e = tid + 512 * q;
        if (e >= nb * 128) break;
        const int b = e >> 7, rr = (e & 127) >> 3, cc = (e & 7) * 2;
        int rb, cb;
        if (LEVEL == 0) { rb = b; cb = b; }
        else if (LEVEL == 1) { rb = 2 * b + 1; cb = 2 * b; }
        else if (LEVEL == 2) { rb = 4 * (b >> 2) + 2 + ((b >> 1) & 1); cb = 4 * (b >> 2) + (b & 1); }
        else { rb = 4 + (b >> 2); cb = b & 3; }
        const int r = rb * 16 + rr, c = cb * 16 + cc;
        double2_t v;
        if (LEVEL == 0) {
            v = *(const double2_t *)(Dinv + rb * DBLK + rr * DS + cc);
        } else {
            v[0] = T[c * TS + r];
            v[1] = T[(c + 1) * TS + r];
        }
        *(double2_t *)(Iv + r * GP_TILE + c) = v;
    }
}

// the factor's block column p (rows 16 p .. 127; on or below the diagonal only) goes to HBM: `nthreads` threads, this
// thread's index among them `th`
__device__ __forceinline__ void store_factor_column(double *At, long lda, const double *T, int p, int th, int nthreads) {
    const int nchunk = (128 - 16 * p) * 8;
    for (int e = th; e < nchunk; e += nthreads) {
        const int r = 16 * p + (e >> 3), c = 16 * p + (e & 7) * 2;
        const double2_t v = *(const double2_t *)(T + r * TS + c);
        if (c + 1 <= r) *(double2_t *)(At + (long)r * lda + c) = v;
        else if (c == r) At[(long)r * lda + c] = v[0];
    }
}

// Factor + invert tile t of A (see the header).  T: GP_TILE * TS doubles of LDS, Dinv: 8 * DBLK doubles of LDS, zrow: 16 zeros
// in LDS.  512 threads.
// PRELOADED: the tile (lower block triangle, diagonal micro blocks in full) already sits in T (potrf_pair_kernel's second tile).
template <bool PRELOADED = false>
__device__ __forceinline__ void potrf_tile_body(double *A, long lda, int t, double *invL, int *info, double *T, double *Dinv,
                                                const double *zrow) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    double *At = A + (long)t * GP_TILE * lda + (long)t * GP_TILE;
    double *Iv = invL + (long)t * GP_TILE * GP_TILE;
    STAMP(0);

    // ---- load: block column 0 first (all E(0) needs), the rest of the lower block triangle (diagonal micro blocks in full:
    //      the input is symmetric there) by waves 3..7 while waves 0..2 eliminate micro panel 0 ----
    if (PRELOADED) {
        if (wave < 3) {
            const int rb0 = 1 + 3 * wave;
            const int fail = eliminate_panel(T, zrow, 0, rb0, min(3, 8 - rb0), lane);
            if (wave == 0 && fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + fail + 1);
        }
        __syncthreads();
    } else {
        double2_t c0[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int id = tid + 512 * q;               // 128 rows x 8 chunks
            c0[q] = *(const double2_t *)(At + (long)(id >> 3) * lda + (id & 7) * 2);
        }
        double2_t rest[12];
        if (wave >= 3) {
            // chunks of block columns 1..7, rows from the column's diagonal block down: 28 blocks x 128 chunks = 3584,
            // enumerated block by block (block e >> 7 of the list: column cb = 1..7, row rb = cb..7), 320 threads x 12 (11.2)
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int e = (tid - 192) + 320 * q;
                if (e < 3584) {
                    const int b = e >> 7;
                    int cb = 1, rem = b;
                    while (rem >= 8 - cb) { rem -= 8 - cb; ++cb; }     // wave-divergent but tiny (<= 7 steps)
                    const int rb = cb + rem;
                    const int r = rb * 16 + ((e & 127) >> 3), c = cb * 16 + (e & 7) * 2;
                    rest[q] = *(const double2_t *)(At + (long)r * lda + c);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int id = tid + 512 * q;
            *(double2_t *)(T + (id >> 3) * TS + (id & 7) * 2) = c0[q];
        }
        __syncthreads();
        STAMP(1);
        // phase 1 of micro panel 0:  E(0) on waves 0..2  ||  the rest of the tile lands in LDS
        if (wave < 3) {
            const int rb0 = 1 + 3 * wave;
            const int fail = eliminate_panel(T, zrow, 0, rb0, min(3, 8 - rb0), lane);
            if (wave == 0 && fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + fail + 1);
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int e = (tid - 192) + 320 * q;
                if (e < 3584) {
                    const int b = e >> 7;
                    int cb = 1, rem = b;
                    while (rem >= 8 - cb) { rem -= 8 - cb; ++cb; }
                    const int rb = cb + rem;
                    const int r = rb * 16 + ((e & 127) >> 3), c = cb * 16 + (e & 7) * 2;
                    *(double2_t *)(T + r * TS + c) = rest[q];
                }
            }
        }
        __syncthreads();
    }

    // ---- factorisation ---------------------------------------------------------------------------------------------
    for (int p = 0; p < 8; ++p) {
        STAMP(2 + 3 * p);
        if (p >= 1) {
            // phase 1:  E(p) on waves 0 .. nE-1  ||  on the others: block column p+1 takes the updates of ALL earlier panels
            //           0 .. p-1 (left-looking: a trailing block is read and written once per micro panel it waits for, and the
            //           work per window is even -- (7-p) 4p matrix instructions -- instead of the 84, 60, 40, ... of applying
            //           panel p-1 to every trailing block at once), and the factor's finished block column p-1 leaves for HBM
            const int nrows = 7 - p;
            const int nE = nrows > 0 ? (nrows + 2) / 3 : 1;     // 2, 2, 2, 1, 1, 1, 1 for p = 1 .. 7
            // The eliminating waves are bound by their own instruction issue: the waves that share their SIMDs (wave w + 4 sits
            // on the SIMD of wave w) stay idle, the other 8 - 2 nE do the side work.  Worker index of a wave: its rank among
            // them, or -1.
            int worker = -1;
            const int nworkers = 8 - 2 * nE;
            if (wave >= nE && !(wave >= 4 && wave < 4 + nE)) worker = wave < 4 ? wave - nE : wave - 2 * nE;
            if (wave < nE) {
                __builtin_amdgcn_s_setprio(3);
                const int rb0 = p + 1 + 3 * wave;
                const int nrb = min(3, 8 - rb0);
                const int fail = eliminate_panel(T, zrow, p, rb0, nrb > 0 ? nrb : 0, lane);
                if (wave == 0 && fail >= 0 && lane == 0) atomicCAS(info, 0, t * GP_TILE + p * 16 + fail + 1);
                __builtin_amdgcn_s_setprio(0);
            } else if (worker >= 0) {
                for (int i = p + 1 + worker; i < 8; i += nworkers) update_block_panels(T, p, i, p + 1, li, lg);
                store_factor_column(At, lda, T, p - 1, worker * 64 + lane, nworkers * 64);
                // the diagonal micro blocks 0..6 are inverted here, two per window from micro panel 4 on (six workers and few
                // blocks by then; more than two substitutions at a time are bound by their LDS broadcast reads, and on the
                // eliminating wave's SIMD partner one takes 5k cycles at its low priority)
                if (p >= 4 && worker >= 4) {
                    const int blk = 2 * (p - 4) + (worker - 4);
                    if (blk < 7) inv16(T, Dinv, blk, lane);
                }
            }
            __syncthreads();
        }
        STAMP(3 + 3 * p);
        // phase 2:  Uc(p): block column p+1 takes panel p's update (block (p+1+wave, p+1))
        if (p < 7) {
            if (p + 1 + wave < 8) update_block(T, p, p + 1 + wave, p + 1, li, lg);
            __syncthreads();
        }
        STAMP(4 + 3 * p);
    }

    STAMP(26);
    // ---- inverse.  The diagonal micro blocks 0..6 were inverted beside E(4..7); block 7 follows now on wave 0 while waves 1..3
    //      take the level-1 pairs that do not need it and the factor's last block column goes out ----
    auto level1 = [&](int j) {   // pair (j, j+1):  Inv[i][j] = -Dinv[i] (L[i][j] Dinv[j]),  i = j + 1
        const int i = j + 1;
        double4_t P = {0.0, 0.0, 0.0, 0.0}, Y = {0.0, 0.0, 0.0, 0.0};
        P = mma<0>(lda_rowmajor(TBLK(i, j), TS, li, lg), ldb_rowmajor(Dinv + j * DBLK, DS, li, lg), P);
        Y = mma<1>(lda_rowmajor(Dinv + i * DBLK, DS, li, lg), acc_as_b(P), Y);
        st_transposed(TBLK(j, i), TS, li, lg, Y);
    };
    if (wave == 0) inv16(T, Dinv, 7, lane);
    else if (wave < 4) level1(2 * (wave - 1));
    else store_factor_column(At, lda, T, 7, tid - 256, 256);
    __syncthreads();
    STAMP(27);
    store_inverse_level<0>(Iv, T, Dinv, tid);
    if (wave == 0) level1(6);
    __syncthreads();
    STAMP(28);
    store_inverse_level<1>(Iv, T, Dinv, tid);
    // level 2: quads Q = 0, 1:  I = {4Q, 4Q+1}, J = {4Q+2, 4Q+3}
    {
        const int Q = wave >> 2, I0 = 4 * Q, J0 = 4 * Q + 2;
        const int k = J0 + ((wave >> 1) & 1), c = I0 + (wave & 1);
        // P[k][c] = sum_{m in I, m >= c} L[k][m] Inv[m][c]
        double4_t P = {0.0, 0.0, 0.0, 0.0};
        P = mma<0>(lda_rowmajor(TBLK(k, c), TS, li, lg), ldb_rowmajor(Dinv + c * DBLK, DS, li, lg), P);
        if (c == I0) P = mma<0>(lda_rowmajor(TBLK(k, I0 + 1), TS, li, lg), ldb_transposed(TBLK(I0, I0 + 1), TS, li, lg), P);
        __syncthreads();                       // every L block of the level has been read
        st_rowmajor(TBLK(k, c), TS, li, lg, P);
        __syncthreads();
        // Inv[r][c] = -sum_{k' in J, k' <= r} Inv[r][k'] P[k'][c]
        const int r = k;
        double4_t Y = {0.0, 0.0, 0.0, 0.0};
        if (r == J0 + 1) Y = mma<1>(lda_transposed(TBLK(J0, J0 + 1), TS, li, lg), ldb_rowmajor(TBLK(J0, c), TS, li, lg), Y);
        Y = mma<1>(lda_rowmajor(Dinv + r * DBLK, DS, li, lg), ldb_rowmajor(TBLK(r, c), TS, li, lg), Y);
        st_transposed(TBLK(c, r), TS, li, lg, Y);
    }
    __syncthreads();
    STAMP(29);
    store_inverse_level<2>(Iv, T, Dinv, tid);
    // level 3: I = {0..3}, J = {4..7}; every wave runs two independent accumulation chains (5 block products in all)
    {
        // P[k][c] = sum_{m = c..3} L[k][m] Inv[m][c]: wave w takes row k = 4 + (w >> 1), columns c1 = w & 1 and c2 = 3 - c1
        const int k = 4 + (wave >> 1);
        const int c1 = wave & 1, c2 = 3 - c1;
        double4_t P1 = {0.0, 0.0, 0.0, 0.0}, P2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (c1 + s < 4)
                P1 = mma<0>(lda_rowmajor(TBLK(k, c1 + s), TS, li, lg),
                            s == 0 ? ldb_rowmajor(Dinv + c1 * DBLK, DS, li, lg) : ldb_transposed(TBLK(c1, c1 + s), TS, li, lg), P1);
            if (c2 + s < 4)
                P2 = mma<0>(lda_rowmajor(TBLK(k, c2 + s), TS, li, lg),
                            s == 0 ? ldb_rowmajor(Dinv + c2 * DBLK, DS, li, lg) : ldb_transposed(TBLK(c2, c2 + s), TS, li, lg), P2);
        }
        __syncthreads();
        st_rowmajor(TBLK(k, c1), TS, li, lg, P1);
        st_rowmajor(TBLK(k, c2), TS, li, lg, P2);
        __syncthreads();
        // Inv[r][c] = -sum_{k' = 4..r} Inv[r][k'] P[k'][c]: wave w takes column c = w >> 1, rows r1 = 4 + (w & 1), r2 = 7 - (w & 1)
        const int c = wave >> 1;
        const int r1 = 4 + (wave & 1), r2 = 7 - (wave & 1);
        double4_t Y1 = {0.0, 0.0, 0.0, 0.0}, Y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = 4 + s;
            if (kk <= r2) {   // (r1 <= r2)
                const Frag b = ldb_rowmajor(TBLK(kk, c), TS, li, lg);
                if (kk <= r1)
                    Y1 = mma<1>(kk == r1 ? lda_rowmajor(Dinv + r1 * DBLK, DS, li, lg) : lda_transposed(TBLK(kk, r1), TS, li, lg), b, Y1);
                Y2 = mma<1>(kk == r2 ? lda_rowmajor(Dinv + r2 * DBLK, DS, li, lg) : lda_transposed(TBLK(kk, r2), TS, li, lg), b, Y2);
            }
        }
        st_transposed(TBLK(c, r1), TS, li, lg, Y1);
        st_transposed(TBLK(c, r2), TS, li, lg, Y2);
    }
    __syncthreads();
    STAMP(30);
    store_inverse_level<3>(Iv, T, Dinv, tid);
    STAMP(31);
}

// ---- two diagonal tiles per launch -----------------------------------------------------------------------------------------
// The factorisation's chain costs three dependent launches per 128 columns (this kernel, the panel solve, the update of the
// panel's remaining columns).  potrf_pair_kernel takes the 256 x 256 diagonal block [A00; A10 A11] in ONE launch and ONE
// workgroup, the tile resident in LDS throughout:
//   body(t)        A00 = L00 L00^T, inv00 = L00^-1                                   (as potrf_tile_kernel)
//   L10 = A10 inv00^T    every wave takes a 16-row block of A10: its operand fragments straight from global memory (32 bytes
//                        contiguous per lane and k block), inv00 from where body(t) left it in LDS (transposed in T's upper
//                        blocks, diagonal micro blocks in Dinv); 8 accumulators per wave, 144 matrix instructions
//   A11 -= L10 L10^T     L10 goes to LDS (and to HBM) and the lower block triangle of A11 takes the rank-128 update, 4-5
//                        micro blocks per wave, A11's old values as the accumulators' initial values
//   body(t+1)      with the tile already in LDS
// after which ONE launch solves both tile columns of the rows below (trsm2.hip) and ONE K = 256 launch updates the rest of the
// panel: half the dependent launches per column, and products twice as long.
__device__ __forceinline__ void pair_offdiag_solve(const double *A10, long lda, double *T, const double *Dinv, int wave, int li,
                                                   int lg, double4_t (&acc)[8]) {
    // A fragments of this wave's 16 rows: step e of k block Kb contracts k = 16 Kb + 4 lg + e (the same map on the B side); the
    // fragments of k block Kb + 1 are fetched while block Kb multiplies
    const double *ap = A10 + (long)(16 * wave + li) * lda + 4 * lg;
    double2_t n0 = *(const double2_t *)ap, n1 = *(const double2_t *)(ap + 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int Kb = 0; Kb < 8; ++Kb) {
        const double a[4] = {n0[0], n0[1], n1[0], n1[1]};
        if (Kb < 7) {
            n0 = *(const double2_t *)(ap + 16 * (Kb + 1));
            n1 = *(const double2_t *)(ap + 16 * (Kb + 1) + 2);
        }
        // B[k][n] = inv00[16 Jc + n][16 Kb + k]: Kb < Jc parked at T[(16 Kb + k) TS + 16 Jc + n], Kb == Jc in Dinv[Jc][n][k];
        // two column blocks at a time: two independent accumulation chains
#pragma unroll
        for (int Jc = Kb; Jc < 8; Jc += 2) {
            double b0[4], b1[4];
            if (Jc == Kb) {
                const double *q = Dinv + Jc * DBLK + li * DS + 4 * lg;
#pragma unroll
                for (int e = 0; e < 4; ++e) b0[e] = q[e];
            } else {
                const double *q = T + (16 * Kb + 4 * lg) * TS + 16 * Jc + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) b0[e] = q[e * TS];
            }
            if (Jc + 1 < 8) {
                const double *q = T + (16 * Kb + 4 * lg) * TS + 16 * (Jc + 1) + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) b1[e] = q[e * TS];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b0[e], acc[Jc], 0, 0, 0);
                if (Jc + 1 < 8) acc[Jc + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], b1[e], acc[Jc + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every block's operand reads to the top)
        }
    }
}

// micro block b of the lower block triangle enumerated row by row: (0,0), (1,0), (1,1), (2,0), ...
__device__ __forceinline__ void tri_block(int b, int &I, int &J) {
    I = 0;
    while ((I + 1) * (I + 2) / 2 <= b) ++I;
    J = b - I * (I + 1) / 2;
}

__global__ __launch_bounds__(512) void potrf_pair_kernel(double *A, long lda, int t, double *invL, int *info) {
    __shared__ __attribute__((aligned(16))) double T[GP_TILE * TS];
    __shared__ __attribute__((aligned(16))) double Dinv[8 * DBLK];
    __shared__ __attribute__((aligned(16))) double zrow[16];
    if (threadIdx.x < 16) zrow[threadIdx.x] = 0.0;
    potrf_tile_body<false>(A, lda, t, invL, info, T, Dinv, zrow);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    double *A10 = A + (long)(t + 1) * GP_TILE * lda + (long)t * GP_TILE;
    double *A11 = A10 + GP_TILE;
    // this wave's micro blocks of A11 (blocks wave, wave + 8, ... of the 36 of the lower block triangle, diagonal ones in full)
    // are requested now and arrive while L10 is computed
    double4_t upd[5];
    int bi[5], bj[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int b = wave + 8 * q;
        tri_block(b < 36 ? b : 0, bi[q], bj[q]);
        if (b < 36) {
#pragma unroll
            for (int r = 0; r < 4; ++r) upd[q][r] = A11[(long)(16 * bi[q] + 4 * r + lg) * lda + 16 * bj[q] + li];
        }
    }
    // ---- L10 = A10 inv00^T (the inverse's last level was stored from T / Dinv by every thread: they are complete and at rest) ----
    double4_t acc[8];
    pair_offdiag_solve(A10, lda, T, Dinv, wave, li, lg, acc);
    __syncthreads();                       // every wave has read the inverse out of T
#pragma unroll
    for (int Jc = 0; Jc < 8; ++Jc)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * wave + 4 * r + lg) * TS + 16 * Jc + li] = acc[Jc][r];
    __syncthreads();
    // L10 leaves for HBM (16-byte stores) while the update below runs
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 512 * q, r = e >> 6, c = (e & 63) * 2;
        *(double2_t *)(A10 + (long)r * lda + c) = *(const double2_t *)(T + r * TS + c);
    }
    // ---- A11 -= L10 L10^T on the lower block triangle ----
#pragma unroll 1
    for (int Kb = 0; Kb < 8; ++Kb) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (wave + 8 * q < 36) {
                const double *ap = T + (16 * bi[q] + li) * TS + 16 * Kb + 4 * lg;
                const double *bp = T + (16 * bj[q] + li) * TS + 16 * Kb + 4 * lg;
                const double2_t a0 = *(const double2_t *)ap, a1 = *(const double2_t *)(ap + 2);
                const double2_t b0 = *(const double2_t *)bp, b1 = *(const double2_t *)(bp + 2);
                upd[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], upd[q], 0, 0, 1);
                upd[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], upd[q], 0, 0, 1);
                upd[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], upd[q], 0, 0, 1);
                upd[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], upd[q], 0, 0, 1);
            }
        }
    }
    __syncthreads();                       // every wave has read L10 out of T
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (wave + 8 * q < 36) {
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 * bi[q] + 4 * r + lg) * TS + 16 * bj[q] + li] = upd[q][r];
        }
    __syncthreads();
    potrf_tile_body<true>(A, lda, t + 1, invL, info, T, Dinv, zrow);
}

__global__ __launch_bounds__(512) void potrf_tile_kernel(double *A, long lda, int t, double *invL, int *info) {
    __shared__ __attribute__((aligned(16))) double T[GP_TILE * TS];
    __shared__ __attribute__((aligned(16))) double Dinv[8 * DBLK];
    __shared__ __attribute__((aligned(16))) double zrow[16];   // the row of zeros E's idle lanes read
    if (threadIdx.x < 16) zrow[threadIdx.x] = 0.0;             // (visible after the first barrier of the body)
    potrf_tile_body<false>(A, lda, t, invL, info, T, Dinv, zrow);
}

// Test hook (option "debug_potrf_lds"): extra dynamic LDS requested with every diagonal-tile launch.  Beyond what the CU has
// left beside the kernel's 151 KB of static LDS the launch is REFUSED -- which is what the launch checks are there to catch.
static int g_debug_lds = 0;
void potrf_set_debug_lds(int bytes) { g_debug_lds = bytes; }

void launch_potrf_pair(hipStream_t s, double *A, long lda, int t, double *invL, int *info) {
    GP_LAUNCH(potrf_pair_kernel, dim3(1), dim3(512), (size_t)g_debug_lds, s, A, lda, t, invL, info);
}

void launch_potrf_tile(hipStream_t s, double *A, long lda, int t, double *invL, int *info) {
    GP_LAUNCH(potrf_tile_kernel, dim3(1), dim3(512), (size_t)g_debug_lds, s, A, lda, t, invL, info);
}

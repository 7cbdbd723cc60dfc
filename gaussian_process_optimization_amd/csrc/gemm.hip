// fp64 MFMA "NT" GEMM for gfx950:  C (op)= A * B^T  over a set of 128x128 output tiles.
//
// This one kernel carries every dense contraction of the path:
//   * Cholesky trailing update  A22 -= L21 L21^T      (dsyrk/dgemm inside LAPACK dpotrf,
//                                                      reference call site GPy/GPy/util/linalg.py:58)
//   * panel solve               L21  = A21 L11^-T     (as a product with the inverted diagonal tile)
//   * candidate solve           T   -= T_J L_J^T, T_b = T_b L_bb^-T
//                                                     (LAPACK dtrtrs, reference call site
//                                                      GPy/GPy/inference/latent_function_inference/posterior.py:294)
//
// Layout: everything row-major fp64.  A rows and B rows both have the contraction index k
// contiguous, so both operand tiles are staged the same way: 128 rows x BK(16) doubles per
// stage, global -> registers (16 B per lane, 128 B contiguous per row) -> LDS with an 18-double
// row pitch (16-B aligned rows; the operand fetch is one ds_read_b128 per lane = two k values
// of row lane&15, at worst 2-way conflicted, which is <15 % LDS occupancy next to the MFMAs).
//
// Work decomposition: 256 threads = 4 waves as 2x2; each wave owns a 64x64 sub-tile =
// 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  Per BK stage a wave issues 64 MFMAs
// against 16 ds_read_b128, i.e. the matrix pipe is the only busy unit; the next stage's global
// loads are issued before the MFMAs and written to the other LDS buffer BETWEEN the stage's two
// halves, in front of the stage's one barrier: the second half's fragments are in registers by
// then, so a wave released from the barrier issues MFMAs at once, and the fragments the first
// MFMAs of a half need are read in the middle of the half before it (round 3: 65.9 -> 69.0
// TFLOP/s for the kernel alone).  Two workgroups per CU (73.7 KB LDS each, <=256 VGPRs) keep a second
// wave per SIMD ready while the first sits at the barrier or in the C epilogue.
//
// Roofline: algorithmic flops per launch = 2 * 128*128 * K * ntiles; bound = fp64 MFMA.
#include "gphip_internal.h"
#include <algorithm>

#define BK 16
#define LSTR 18  // LDS row pitch in doubles

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
// C tile traffic goes through buffer instructions: one SGPR descriptor per wave, the lane part of the
// address as a single 32-bit VGPR offset and the row stride as an SGPR offset, instead of 16 64-bit
// VGPR row pointers that would otherwise live across the whole K loop.
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v[1], (int)v[0]);
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double d) {
    v2u_t v;
    v[0] = (unsigned)__double2loint(d);
    v[1] = (unsigned)__double2hiint(d);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

struct GemmArgs {
    double *C;
    long ldc;
    const double *A;
    long lda;
    const double *B;
    long ldb;
    int b_mul;
    int K;
    int r0, r1, c0, c1, tri;
    int k_tri;      // contraction starts at column max(0, ti - k_sub)*128 (A is block upper-triangular: lauum-type products)
    int k_sub;
    int k_end_tri;  // contraction ends after column tile (tc - b_sub): B is block lower-triangular
    int b_sub;      // B row tile = (tc - b_sub) * b_mul
    long sC, sA, sB;  // batch strides (elements) applied with blockIdx.y
    const short *tile_list;  // optional explicit (row tile, col tile) order (L2-friendly super-tiles)
    int stagger;             // > 0: odd-slot workgroups start stagger * 1024 cycles late
    int pair;                // k_end_tri products: a workgroup takes column tiles c1-1-p and c0+p (equal total K)
};

__device__ __forceinline__ void tile_from_linear(long t, const GemmArgs &a, int &i, int &c) {
    if (!a.tri) {
        int nr = a.r1 - a.r0;
        c = (int)(t / nr);
        // triangular-K products: the last column tiles contract the longest K, start them first
        c = a.k_end_tri ? a.c1 - 1 - c : a.c0 + c;
        i = a.r0 + (int)(t % nr);
        return;
    }
    // column j = c - c0 holds H - j tiles (rows c .. r1-1), H = r1 - c0; prefix S(j) = j*H - j(j-1)/2
    const long H = a.r1 - a.c0;
    const double b = 2.0 * (double)H + 1.0;
    long j = (long)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    if (j < 0) j = 0;
    long S = j * H - j * (j - 1) / 2;
    while (S > t) {
        --j;
        S = j * H - j * (j - 1) / 2;
    }
    while (S + (H - j) <= t) {
        S += (H - j);
        ++j;
    }
    c = a.c0 + (int)j;
    i = c + (int)(t - S);
}

// BT = block tile edge: 128 (throughput: 4x4 MFMA accumulators per wave, 2 workgroups/CU) or 64 (latency:
// 2x2 accumulators per wave, a quarter of the work per workgroup and up to 4 workgroups/CU -- used for the
// short launches of the factorisation's latency chain and the uneven triangular-K products).  With BT = 64
// every 128x128 tile of the tile set is computed by four workgroups.
// BM = 64 with BT = 128 (ROWS variant): the workgroup computes a 64-row strip of all 128 columns of the tile -- two
// workgroups per tile, each reading only its own rows of A, so an IN-PLACE product (C aliases A: the panel solve)
// can be split over two CUs.
template <int MODE, int BT, int NWN, bool PAIR = false, int BM = BT>
__global__ __launch_bounds__(128 * NWN, (BT == 128 ? (BM < BT ? 2 : NWN) : 4)) void gemm_nt_kernel(GemmArgs a) {
    constexpr int NTH = 128 * NWN;         // threads: 2 x NWN waves
    constexpr int MT = BM / 32;            // 16x16 MFMA tiles per wave along m
    constexpr int NT = BT / NWN / 16;      // ... and along n
    constexpr int WT = BM / 2;             // wave tile rows
    constexpr int WTN = BT / NWN;          // wave tile columns
    constexpr int LQA = BM * 8 / NTH;      // 16-byte staging loads per thread: A rows ...
    constexpr int LQ = BT * 8 / NTH;       // ... and B rows
    constexpr int LR = NTH / 8;            // rows covered by one staging load of the workgroup
    constexpr int SBUF = (BM + BT) * LSTR; // doubles per stage buffer: [A: BM x 18][B: BT x 18]
    __shared__ __attribute__((aligned(16))) double smem[2 * SBUF];

    // XCD-aware remap: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous
    // run of the tile list so that neighbouring tiles (same A row panel) hit the same L2.
    const long nwg = gridDim.x, bid = blockIdx.x;
    const long q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    long wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int sub = 0;
    if (BT == 64) {
        sub = (int)(wg & 3);
        wg >>= 2;
    } else if (BM < BT) {
        sub = (int)(wg % (BT / BM));
        wg /= (BT / BM);
    }

    int ti, tc, tc2 = -1;
    if (PAIR) {
        // balanced triangular-K product: column tiles c1-1-p (longest K) and c0+p (shortest) in one workgroup
        const int nr = a.r1 - a.r0;
        const int cp = (int)(wg / nr);
        ti = a.r0 + (int)(wg % nr);
        tc = a.c1 - 1 - cp;
        tc2 = a.c0 + cp;
    } else if (a.tile_list) {
        ti = a.tile_list[2 * wg];
        tc = a.tile_list[2 * wg + 1];
    } else {
        tile_from_linear(wg, a, ti, tc);
    }
    // the triangular decode goes through a VALU sqrt: tell the compiler the result is wave-uniform
    ti = __builtin_amdgcn_readfirstlane(ti);
    tc = __builtin_amdgcn_readfirstlane(tc);
    tc2 = __builtin_amdgcn_readfirstlane(tc2);
    const int sr = (BT == 64) ? (sub >> 1) * 64 : (BM < BT ? sub * BM : 0);  // row / column offset inside the 128-tile
    const int sc = (BT == 64) ? (sub & 1) * 64 : 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: C addressing stays scalar
    const int wm = wave / NWN, wn = wave % NWN;
    const int li = lane & 15, lg = lane >> 4;

    const int npass = (PAIR && tc2 != tc) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
    if (PAIR && pass == 1) tc = tc2;
    const long z = blockIdx.y;
    const int kstart = a.k_tri ? (ti > a.k_sub ? ti - a.k_sub : 0) * GP_TILE : 0;
    const int kend = a.k_end_tri ? (tc - a.b_sub + 1) * GP_TILE : a.K;
    const double *Ag = a.A + z * a.sA + ((long)ti * GP_TILE + sr) * a.lda + kstart;
    const double *Bg = a.B + z * a.sB + ((long)(tc - a.b_sub) * a.b_mul * GP_TILE + sc) * a.ldb + kstart;

    // staging map: BT rows x 8 chunks(16 B); thread handles rows (tid>>3) + LR*q, chunk tid&7
    const int srow = tid >> 3, sch = (tid & 7) * 2;
    const double *ap = Ag + (long)srow * a.lda + sch;
    const double *bp = Bg + (long)srow * a.ldb + sch;
    const long a32 = (long)LR * a.lda, b32 = (long)LR * a.ldb;
    const int soff = srow * LSTR + sch;

    // accumulator element r of tile (m,n) is C[row = lg + 4r][col = li] of that 16x16 tile
    double *Cw = a.C + z * a.sC + ((long)ti * GP_TILE + sr + wm * WT) * a.ldc + (long)tc * GP_TILE + sc + wn * WTN;  // uniform
    const unsigned cbyte = (unsigned)(lg * (int)a.ldc + li) * 8u;  // lane part of the address, bytes
    const unsigned crow = (unsigned)a.ldc * 8u;                    // row stride, bytes (wave tile spans <= 64 rows: < 2^31)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(Cw, 0, 0x7fffffff, 0x00020000);
    double4_t acc[MT][NT];
    if (MODE == 1) {
        // C -= A B^T: the old C rides in as the MFMA C operand (A negated by the f64 MFMA neg modifier), so
        // its HBM read overlaps the first operand-tile loads and the epilogue is stores only.
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[m][n][r] = buf_load_f64(crs, cbyte + n * 128, (unsigned)(m * 16 + 4 * r) * crow);
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }

    double2_t ra[LQA], rb[LQ];
    const int nk = (kend - kstart) / BK;

    // Two workgroups share a CU (one wave of each per SIMD) and start together, so they would reach their
    // barriers -- and leave the matrix pipe idle -- at the same moments.  The workgroup whose waves sit in the
    // odd hardware wave slot starts a fraction of a stage later (a stage is ~4000 cycles of MFMA per wave).
    if (BT == 128 && a.stagger) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (tid == 0) smem[0] = (double)(hwid & 1u);
        __syncthreads();
        const bool late = smem[0] != 0.0;
        __syncthreads();
        if (late) {
            for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 16 * 64 cycles each
        }
    }

#pragma unroll
    for (int q = 0; q < LQA; ++q) ra[q] = *(const double2_t *)(ap + q * a32);
#pragma unroll
    for (int q = 0; q < LQ; ++q) rb[q] = *(const double2_t *)(bp + q * b32);
    {
        double *As = smem, *Bs = smem + BM * LSTR;
#pragma unroll
        for (int q = 0; q < LQA; ++q) *(double2_t *)(As + soff + q * LR * LSTR) = ra[q];
#pragma unroll
        for (int q = 0; q < LQ; ++q) *(double2_t *)(Bs + soff + q * LR * LSTR) = rb[q];
    }
    __syncthreads();

    const int aoff = (wm * WT + li) * LSTR + lg * 4;
    const int boff = BM * LSTR + (wn * WTN + li) * LSTR + lg * 4;

    // Stage schedule: the fragments of the second half of a stage are read BEFORE the stage's barrier and its MFMAs issued after
    // it, so that a wave released from the barrier has 16 MFMAs to issue at once (no LDS latency behind the barrier's skew); the
    // first half of the next stage is read after them.  Every read of a buffer precedes the barrier in front of the stage that
    // overwrites it.  Lane group lg owns k = 4lg..4lg+3 of the stage; MFMA step (h,e) contracts k = 4g + 2h + e over the four lane
    // groups g, so two ds_read_b128 per operand row-tile feed four MFMA steps.
    // Fragment registers: the second half of a stage multiplies af[0..MT) x bf[0..NT); the first half multiplies
    // {pa[0..PFA), af[PFA..MT)} x pb[0..NT).  The fragments the first MFMAs of a half need -- PFA row tiles and every column tile --
    // are read in the MIDDLE of the half before it, the others right after its last MFMA: a half's first MFMAs never wait for
    // LDS.  PFA = 1 where registers are short (the 8-wave variant: 113 -> 125 VGPRs, four waves per SIMD), every row tile where
    // they are not (two waves per SIMD: a full second fragment set), none for the paired variant (two tile addresses to keep).
    constexpr int PFA = PAIR ? 0 : ((BT == 128 && BM == 128 && NWN == 2) ? MT : 1);
    constexpr bool PF = PFA > 0;
    double2_t af[MT], bf[NT], pa_own[PF ? PFA : 1], pb_own[NT];
    double2_t *const pa = PF ? pa_own : af;      // (without the prefetch the first half uses the plain set: one set of fragment registers)
    double2_t *const pb = PF ? pb_own : bf;
    constexpr int PA = PF ? PFA : 1;             // leading A fragments of the first half live in pa[0..PA)
    {
        const double *as = smem + aoff, *bs = smem + boff;
#pragma unroll
        for (int m = 0; m < PA; ++m) pa[m] = *(const double2_t *)(as + m * 16 * LSTR);
#pragma unroll
        for (int n = 0; n < NT; ++n) pb[n] = *(const double2_t *)(bs + n * 16 * LSTR);
#pragma unroll
        for (int m = PA; m < MT; ++m) af[m] = *(const double2_t *)(as + m * 16 * LSTR);
    }
#define GP_MMA(A_, B_, m_, n_, e_) acc[m_][n_] = __builtin_amdgcn_mfma_f64_16x16x4f64((A_)[e_], (B_)[e_], acc[m_][n_], 0, 0, MODE == 1 ? 1 : 0)
#define GP_A1(m_) ((m_) < PA ? pa[m_] : af[m_])
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
            ap += BK;
            bp += BK;
#pragma unroll
            for (int q = 0; q < LQA; ++q) ra[q] = *(const double2_t *)(ap + q * a32);
#pragma unroll
            for (int q = 0; q < LQ; ++q) rb[q] = *(const double2_t *)(bp + q * b32);
        }
        const double *as = smem + buf * SBUF + aoff;
        const double *bs = smem + buf * SBUF + boff;
        // ---- first half (k pairs 0, 1 of every lane group) ----
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) GP_MMA(GP_A1(m), pb[n], m, n, 0);
        if (PF) {
            __builtin_amdgcn_sched_barrier(0);
            // (af[0..PFA) and bf[] are free here: the second half's leading fragments)
#pragma unroll
            for (int m = 0; m < PFA; ++m) af[m] = *(const double2_t *)(as + m * 16 * LSTR + 2);
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const double2_t *)(bs + n * 16 * LSTR + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) GP_MMA(GP_A1(m), pb[n], m, n, 1);
        if (PF) __builtin_amdgcn_sched_barrier(0);
        if (!PF) {
            af[0] = *(const double2_t *)(as + 2);
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *(const double2_t *)(bs + n * 16 * LSTR + 2);
        }
#pragma unroll
        for (int m = PA; m < MT; ++m) af[m] = *(const double2_t *)(as + m * 16 * LSTR + 2);
        if (more) {
            double *As = smem + (buf ^ 1) * SBUF, *Bs = As + BM * LSTR;
#pragma unroll
            for (int q = 0; q < LQA; ++q) *(double2_t *)(As + soff + q * LR * LSTR) = ra[q];
#pragma unroll
            for (int q = 0; q < LQ; ++q) *(double2_t *)(Bs + soff + q * LR * LSTR) = rb[q];
        }
        __syncthreads();
        // ---- second half ----
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) GP_MMA(af[m], bf[n], m, n, 0);
        const double *as2 = smem + (buf ^ 1) * SBUF + aoff, *bs2 = smem + (buf ^ 1) * SBUF + boff;
        if (PF) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int m = 0; m < PFA; ++m) pa[m] = *(const double2_t *)(as2 + m * 16 * LSTR);
#pragma unroll
                for (int n = 0; n < NT; ++n) pb[n] = *(const double2_t *)(bs2 + n * 16 * LSTR);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) GP_MMA(af[m], bf[n], m, n, 1);
        if (PF) __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (!PF) {
                pa[0] = *(const double2_t *)(as2);
#pragma unroll
                for (int n = 0; n < NT; ++n) pb[n] = *(const double2_t *)(bs2 + n * 16 * LSTR);
            }
#pragma unroll
            for (int m = PA; m < MT; ++m) af[m] = *(const double2_t *)(as2 + m * 16 * LSTR);
        }
    }
#undef GP_A1
#undef GP_MMA

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                buf_store_f64(crs, cbyte + n * 128, (unsigned)(m * 16 + 4 * r) * crow, acc[m][n][r]);
    }  // pass
}

void launch_gemm_nt(hipStream_t s, int mode, double *C, long ldc, const double *A, long lda,
                    const double *B, long ldb, int b_mul, int K, TileSet ts, const GemmOpt &o) {
    long n = tileset_count(ts);
    if (n <= 0 || K <= 0 || o.batch <= 0) return;
    GemmArgs a;
    a.C = C; a.ldc = ldc; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb;
    a.b_mul = b_mul; a.K = K;
    a.r0 = ts.r0; a.r1 = ts.r1; a.c0 = ts.c0; a.c1 = ts.c1; a.tri = ts.tri;
    a.k_tri = o.k_tri; a.k_sub = o.k_sub; a.k_end_tri = o.k_end_tri; a.b_sub = o.b_sub;
    a.sC = o.sC; a.sA = o.sA; a.sB = o.sB;
    a.tile_list = o.tile_list;
    a.stagger = o.stagger;
    a.pair = o.pair;
    if (o.small && o.pair) {
        // pairs of column tiles: ceil(W/2) workgroup columns per row tile
        const long np = (long)(ts.r1 - ts.r0) * ((ts.c1 - ts.c0 + 1) / 2);
        dim3 grid((unsigned)(4 * np), (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 64, 2, true>), grid, dim3(256), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 64, 2, true>), grid, dim3(256), 0, s, a);
    } else if (o.rows64 == 32) {
        dim3 grid((unsigned)(4 * n), (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 128, 2, false, 32>), grid, dim3(256), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 128, 2, false, 32>), grid, dim3(256), 0, s, a);
    } else if (o.rows64) {
        dim3 grid((unsigned)(2 * n), (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 128, 2, false, 64>), grid, dim3(256), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 128, 2, false, 64>), grid, dim3(256), 0, s, a);
    } else if (o.waves8 && o.pair) {
        const long np = (long)(ts.r1 - ts.r0) * ((ts.c1 - ts.c0 + 1) / 2);
        dim3 grid((unsigned)np, (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 128, 4, true>), grid, dim3(512), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 128, 4, true>), grid, dim3(512), 0, s, a);
    } else if (o.small) {
        dim3 grid((unsigned)(4 * n), (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 64, 2>), grid, dim3(256), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 64, 2>), grid, dim3(256), 0, s, a);
    } else if (o.waves8) {
        dim3 grid((unsigned)n, (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 128, 4>), grid, dim3(512), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 128, 4>), grid, dim3(512), 0, s, a);
    } else {
        dim3 grid((unsigned)n, (unsigned)o.batch);
        if (mode == 0)
            GP_LAUNCH((gemm_nt_kernel<0, 128, 2>), grid, dim3(256), 0, s, a);
        else
            GP_LAUNCH((gemm_nt_kernel<1, 128, 2>), grid, dim3(256), 0, s, a);
    }
}

std::vector<short> build_tile_list(const TileSet &ts, int S) {
    std::vector<short> out;
    const int c_lo = ts.c0, c_hi = ts.c1;
    const int r_lo = ts.tri ? ts.c0 : ts.r0, r_hi = ts.r1;
    for (int sc = c_lo; sc < c_hi; sc += S)
        for (int sr = (ts.tri ? sc : r_lo); sr < r_hi; sr += S)
            for (int c = sc; c < std::min(sc + S, c_hi); ++c)
                for (int i = sr; i < std::min(sr + S, r_hi); ++i) {
                    if (ts.tri && i < c) continue;
                    out.push_back((short)i);
                    out.push_back((short)c);
                }
    return out;
}

// fp64-equivalent contraction on the int8 matrix cores ("emulate_fp64", default OFF; prototype for the candidate solve).
//
// Why: the dense contractions of the path are bound by the fp64 MFMA rate (v_mfma_f64_16x16x4_f64: 78.6 TFLOP/s peak), the
// int8 MFMA (v_mfma_i32_32x32x32_i8) runs at 64x that rate with EXACT int32 accumulation.  Scheme (Ozaki scheme II --
// integer modular emulation of matrix multiplication -- re-arranged so that the accumulator stays in residue form across
// all panel updates and is reconstructed once per column; model and constants: tools/rns_model.py):
//   * operands are fixed-point integers v = rint(x 2^(52-e)) (exact in fp64) with 2^(e-1) >= sqrt(max diag Ky), stored as
//     their symmetric residues v mod p_l (int8) for 14 pairwise coprime moduli p_l <= 253, P = prod p_l ~ 2^109.9.
//     Why 14 are enough: every contraction of the path is a product of two ROWS whose Euclidean norms the factorisation
//     itself bounds -- sum_k L_ck^2 = Ky_cc (Cholesky) and sum_k S_ik^2 <= k(x*, x*) (the posterior variance is >= 0) -- so by
//     Cauchy-Schwarz |sum_k a_k b_k| <= max diag Ky <= 2^(2e-2) for ANY contraction length, i.e. the exact integer
//     |X| <= (2^51 + sqrt(N)/2)^2 ~ 2^102 < P/2 = 2^108.9, seven bits to spare (a bound by entry magnitudes alone,
//     N 2^104, would need 16 moduli).  An entry beyond 2^(e-1) (which a valid factor cannot have) raises the range flag;
//   * T[:, c] -= sum_J S_J L[c, J]^T (the running right-hand side of dtrtrs, posterior.py:294) becomes, per modulus, an int8
//     GEMM whose int32 sums are reduced mod p_l and added to a one-byte residue accumulator R_l -- no rounding anywhere;
//   * before panel c of T is needed in fp64 (for the product with the inverted diagonal panel) the exact integer
//     X = sum a b (|X| < P/2, above) is recovered from its 14 residues by the CRT in fraction form,
//     X / P = centred_frac(sum_l w_l / p_l), w_l = r_l q_l mod p_l, with two fp64 accumulators (terms rounded to
//     multiples of 2^-44 sum exactly; the remainders carry a double-double reciprocal), and T -= X 2^(2e-104).
// The only approximation is the fixed-point rounding of the operands: absolute 2^(e-53) per entry, i.e. one fp64 ulp of
// the largest entries -- the same backward error an fp64 product commits.
//
// Reference call site of the contraction: LAPACK dtrtrs, GPy/GPy/inference/latent_function_inference/posterior.py:294.
#include "gphip_internal.h"

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16v_t __attribute__((ext_vector_type(16)));

#define RNS_T GP_RNS_T
// K bytes per stage BKB = 128 or 64; LDS row pitch BKB + 16 bytes (conflict-free ds_read_b128 over the rows of a lane group)

struct RnsConst {
    float p[RNS_T], inv_p[RNS_T];
    int q[RNS_T];
    double pd[RNS_T], inv_pd[RNS_T], ih[RNS_T], il[RNS_T];
    double P_scaled;  // P * 2^-104
};
__constant__ RnsConst c_rns;

// pairwise coprime, all <= 253: a quotient that is off by one near a half-way case still leaves |remainder| <= 127, so the
// symmetric residues need no fix-up step (0.505 * 253 < 128)
static const int h_moduli[RNS_T] = {253, 251, 249, 247, 245, 241, 239, 233, 229, 227, 223, 211, 199, 197};

static bool g_rns_const_done[64] = {false};
int rns_init_constants(int device) {
    if (device < 0 || device >= 64) return -1;
    if (g_rns_const_done[device]) return 0;
    RnsConst h;
    // P and q_l = (P / p_l)^-1 mod p_l in 128-bit integer arithmetic
    unsigned __int128 P = 1;
    for (int l = 0; l < RNS_T; ++l) P *= (unsigned)h_moduli[l];
    for (int l = 0; l < RNS_T; ++l) {
        const int p = h_moduli[l];
        unsigned __int128 Pl = P / (unsigned)p;
        const int r = (int)(Pl % (unsigned)p);
        int inv = 0;
        for (int x = 1; x < p; ++x)
            if ((r * x) % p == 1) {
                inv = x;
                break;
            }
        h.q[l] = inv;
        h.p[l] = (float)p;
        h.inv_p[l] = 1.0f / (float)p;
        h.pd[l] = (double)p;
        h.inv_pd[l] = 1.0 / (double)p;
        h.ih[l] = 1.0 / (double)p;
        // il = 1/p - ih exactly enough: one Newton residual in long double (64-bit mantissa) is below 2^-106 relative
        const long double res = (1.0L - (long double)h.ih[l] * (long double)p) / (long double)p;
        h.il[l] = (double)res;
    }
    long double Pl = 1.0L;
    for (int l = 0; l < RNS_T; ++l) Pl *= (long double)h_moduli[l];
    h.P_scaled = (double)(Pl * 0x1p-104L);
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_rns), &h, sizeof h) != hipSuccess) return -1;
    g_rns_const_done[device] = true;
    return 0;
}

// ---- fp64 -> residue planes ---------------------------------------------------------------------------------------
// src: rows x cols (row-major, ld); dst plane l: rows x ldd int8 at dst + l * plane_stride.  scale = 2^(52-e).
// flag: set to 1 when an entry does not fit (|x| scale > 2^51 (1 + 2^-8), see the header): the caller then refuses the
// emulated path.
__global__ __launch_bounds__(256) void rns_convert_kernel(const double *src, long ld, long rows, long cols4, signed char *dst,
                                                          long plane_stride, long ldd, double scale, int *flag) {
    const long c4 = blockIdx.x * 256L + threadIdx.x;  // group of 4 consecutive columns
    const long r = blockIdx.y;
    if (c4 >= cols4) return;
    const double *sp = src + r * ld + c4 * 4;
    const double2_t a = *(const double2_t *)sp, b = *(const double2_t *)(sp + 2);
    double v[4] = {rint(a[0] * scale), rint(a[1] * scale), rint(b[0] * scale), rint(b[1] * scale)};
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) bad |= !(fabs(v[i]) <= 0x1.01p51);
    if (bad) atomicOr(flag, 1);
#pragma unroll
    for (int l = 0; l < RNS_T; ++l) {
        const double p = c_rns.pd[l], ip = c_rns.inv_pd[l];
        unsigned packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double q = rint(v[i] * ip);
            double res = fma(-q, p, v[i]);  // exact: the true remainder is an integer of magnitude <= p
            const int ri = (int)res;   // |ri| <= 127 (see h_moduli)
            packed |= ((unsigned)ri & 0xffu) << (8 * i);
        }
        *(unsigned *)(dst + l * plane_stride + r * ldd + c4 * 4) = packed;
    }
}

void launch_rns_convert(hipStream_t s, const double *src, long ld, long rows, long cols, signed char *dst,
                        long plane_stride, long ldd, double scale, int *flag) {
    const long cols4 = cols / 4;
    dim3 grid((unsigned)((cols4 + 255) / 256), (unsigned)rows);
    GP_LAUNCH(rns_convert_kernel, grid, dim3(256), 0, s, src, ld, rows, cols4, dst, plane_stride, ldd, scale, flag);
}

// Accumulator residues are stored as UNSIGNED bytes, any representative in [0, p + 2] (<= 255): the accumulator is never
// an MFMA operand, and v_cvt_f32_ubyteN / v_cvt_pk_u8_f32 then unpack and pack a byte in one instruction each.
//   fold: s = hi 2^12 + lo (arithmetic shift, lo in [0, 4095])  ->  t = hi (4096 mod p) + lo = s (mod p) with
//   |t| <= |s| / 16 + 4096: three integer instructions bring ANY sum of a contraction of up to GP_RNS_KMAX = 8192 bytes
//   (|s| <= 8192 * 127^2 < 2^27) below 2^23.1, so one launch may contract several panels with NO intermediate reduction;
//   reduce: f = t + old byte (exact in f32); the f32 product f * (1/p) is off by at most 1.5 * 2^-24 * |f| / p < 0.004
//   (|f| / p < 2^23.1 / 191), so q = floor(f / p - 0.006) never exceeds the true quotient and falls short of it by one
//   only when the remainder is below 0.01 p < 2.6: r = f - q p (exact, one fma) is an integer in [0, p + 2] <= 255.
//   Contractions of at most 256 bytes (|s| < 2^22) skip the fold.
__device__ __forceinline__ float rns_reduce_f(int s, float oldf, float p, float inv_p, int c4096, bool fold) {
    int t = s;
    if (fold) t = (s >> 12) * c4096 + (s & 4095);
    const float f = (float)t + oldf;
    const float q = floorf(fmaf(f, inv_p, -0.006f));
    return fmaf(-q, p, f);
}


// ---- residue GEMM: R_l[ti, tc] = (R_l[ti, tc] + A_l[ti rows, 0:K] B_l[tc rows, 0:K]^T) mod p_l, 256 x 256 workgroup tiles ---
// A first version on 128 x 128 tiles (register-staged, 2 or 4 workgroups per CU) ran at 1.3 Pop/s whatever the occupancy;
// 256 x 256 tiles halve the bytes staged per op (256 op/B).  8 waves as 2 x 4, each 128 x 64 =
// 4 x 2 MFMA tiles (128 accumulator registers); K stages of 64 bytes (one workgroup per CU: latency has to be covered by
// the pipeline, not by occupancy).  R blocks are 64 KB per (plane, 256-row tile, 256-column tile), again in register order.
struct RnsGemm256Args {
    const signed char *A;  // plane l, row r: A + l * a_plane + r * lda   (K bytes, contiguous)
    long lda, a_plane;
    const signed char *B;  // already offset to the first contracted column; row c: B + l * b_plane + c * ldb
    long ldb, b_plane;
    signed char *R;        // block (l, ti, tc) at R + ((l * mt_all + ti) * nt_all + tc) * 65536
    int mt_all, nt_all;
    int mt, c0, c1;        // 256-row tiles 0..mt, 256-column tiles c0..c1
    int K;                 // multiple of 128
    int first;
    int tri;               // 1: only the tiles with ti >= tc (lower block triangle: the Cholesky's trailing update)
    int sr, sc;            // super-tile counts: 8 row tiles x 4 column tiles each
    int xcd_interleave;    // 1: the XCDs share a modulus at any time (see the kernel), 0: one contiguous chunk per XCD
};

__device__ __forceinline__ void rns_glds16(const signed char *src, unsigned char *lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_uniform, 16, 0, 0);
}

// K stages of 128 bytes (A 256 rows x 128 B, then B 256 rows x 128 B = 64 KB), double-buffered, filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers; the DMA of stage k+1 is issued at the start of stage k).  LDS-DMA writes
// a wave instruction's 64 x 16 B linearly, so rows cannot be padded: chunk c of row r sits in slot c ^ ((r >> 1) & 7) (the
// permutation is applied to the per-lane SOURCE address), which makes the fragment reads -- 16 consecutive rows x one
// chunk per ds_read_b128 lane group = 16 distinct 16-byte positions of the 256-byte bank row -- conflict-free.
// Inside a stage the fragments of k step j+1 are read in the MFMA gaps of k step j (one DS read per gap, T19 of the
// programming guide), and the stage barrier sits before the last k step, so the matrix pipe is busy while the waves
// synchronise.  Versions measured on the way (C3 candidate solve, ms): 128 x 128 tiles register-staged 37.7 (2 or 4
// workgroups per CU alike); 256 x 256 register-staged 42.3; LDS-DMA ring of 4 / 5 x 32 KB stages 34.6 / 32.4; this one 31.3
// (26.9 once one launch contracts 8 panels and reduces mod p once, rns_reduce_f above).
// Timing ablations of the ring version (wrong results by construction): no MFMA 26.9, no DMA 34.5, no accumulator traffic
// 32.8, none of the three 17.1 (11 ms of it this kernel's skeleton: barriers, fragment reads, reductions) -- the costs add
// up instead of overlapping.  Also measured and dropped: the same product as 4-wave workgroups on 128 x 256 tiles, two
// independent workgroups per CU (33.0 ms), and a row pitch of the L planes off the power of two (no change): neither
// barrier lockstep nor channel conflicts are the limiter.
// What IS the limiter (round 2, measured on a ping-pong variant of this kernel -- the programming guide's 8-phase
// schedule restated for 1-byte operands: 4 x 32 KB ring, the two halves of the workgroup one barrier apart so that one
// half reads fragments and issues DMA while the other issues MFMAs; same results, same 27 ms -- with its parts switched
// off; C3 candidate solve, GEMM share of the time): nothing but barriers + epilogue 5 ms, MFMAs only 11 ms (= the
// 10.9 ms the instruction count needs at 1.95 GHz), fragment reads only 8.6 ms, LDS-DMA ONLY 16.3 ms, all three 20.9 ms.
// The 174 GB a candidate solve stages from L2 into LDS move at 10.7 TB/s (42 GB/s per CU, between the guide's figures
// for L2- and Infinity-Cache-served gathers) with nothing else running: a 256 x 256 tile of 1-byte operands needs 1 byte
// per 256 operations, so the staging path caps this product at ~2.7 Pop/s whatever the schedule (the bare MFMA loop:
// 3.76 Pop/s, tools/micro/i8_peak.hip), and the kernel runs at 78 % of that cap.  Larger tiles are out of registers
// (the 256 x 256 int32 accumulators are half of the CU's register file).
// The staging path is sensitive to WHERE the bytes come from: with the eight XCDs on consecutive super-tiles of one
// modulus at a time (its planes fit the Infinity Cache) instead of one contiguous chunk of the grid -- two moduli -- per
// XCD, the candidate solve runs in 26.1 instead of 27.3 ms (option "rns_interleave", default 1).  A row pitch of the S
// planes off 6144 bytes changes nothing.
#define R256_STAGE 65536
__global__ __launch_bounds__(512, 2) void rns_gemm256_kernel(RnsGemm256Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * R256_STAGE];
    // Workgroups b, b+8, b+16, ... run on one XCD, in that order: 32 consecutive ones of an XCD form a super-tile (8 x 4
    // tiles sharing their operand panels in that XCD's L2), and the super-tiles the eight XCDs work on at the same time are
    // consecutive ones of the SAME modulus, so that what one XCD's L2 misses another has just brought into the
    // Infinity Cache (the planes of one modulus fit it; those of eight moduli -- one contiguous chunk per XCD -- do not).
    const long bid = blockIdx.x;
    const long q = bid >> 3;
    const int within = (int)(q & 31);
    // (rotated by the round number: with 8 row super-tiles an XCD would otherwise see the same row super-tile every round,
    // and in a triangular launch the XCD with the top rows would have nothing to do)
    const long st = a.xcd_interleave ? (q >> 5) * 8 + ((bid + (q >> 5)) & 7) : ((bid & 7) * ((long)gridDim.x >> 3) + q) >> 5;
    const int nst = a.sr * a.sc;
    const int l = (int)(st / nst);
    const int sti = (int)(st % nst);
    const int ti = (sti % a.sr) * 8 + (within & 7);
    const int tc = a.c0 + (sti / a.sr) * 4 + (within >> 3);
    if (l >= RNS_T || ti >= a.mt || tc >= a.c1 || (a.tri && ti < tc)) return;   // (the grid is rounded up to 8 super-tiles)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const float p = c_rns.p[l], inv_p = c_rns.inv_p[l];
    const signed char *Ag = a.A + (long)l * a.a_plane + (long)ti * 256 * a.lda;
    const signed char *Bg = a.B + (long)l * a.b_plane + (long)tc * 256 * a.ldb;
    // accumulator sub-blocks of this wave: rb = wm * 4 + m, cb = wn * 2 + n; tile t = m * 2 + n sits at Rb + RBOFF(t)
    signed char *Rb = a.R + (((long)l * a.mt_all + ti) * a.nt_all + tc) * 65536 + lane * 16;
    const int rb0 = wm * 4, cb0 = wn * 2;
#define RBOFF(t) ((((rb0 + ((t) >> 1)) * 8) + cb0 + ((t) & 1)) * 1024)

    // DMA map: wave w fills rows w*32 .. w*32+31 of A and of B, four instructions each (8 rows x 128 B per instruction)
    const signed char *ga[4], *gb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wave * 32 + j * 8 + (lane >> 3);
        const int ch = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
        ga[j] = Ag + (long)row * a.lda + ch;
        gb[j] = Bg + (long)row * a.ldb + ch;
    }
    unsigned char *la = smem + wave * 32 * 128;
    const int nk = a.K / 128;
    const bool fold = a.K > 256;   // see rns_reduce_f
    const int c4096 = 4096 % (int)p;
    auto issue = [&](int kt) {
        unsigned char *sb = la + (kt & 1) * R256_STAGE;
        const long ko = (long)kt * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) rns_glds16(ga[j] + ko, sb + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) rns_glds16(gb[j] + ko, sb + 32768 + j * 1024);
    };
    issue(0);

    int16v_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0;

    const int frow = lane & 31, fh = lane >> 5;
    int arow[4], brow[2], asw[4], bsw[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = wm * 128 + m * 32 + frow;
        arow[m] = r * 128;
        asw[m] = (r >> 1) & 7;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int r = wn * 64 + n * 32 + frow;
        brow[n] = 32768 + r * 128;
        bsw[n] = (r >> 1) & 7;
    }
    int4_t fa[2][4], fb[2][2];
    auto read_frags = [&](int set, int ks, int kt) {
        const unsigned char *sb = smem + (kt & 1) * R256_STAGE;
        const int c = 2 * ks + fh;
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[set][m] = *(const int4_t *)(sb + arow[m] + ((c ^ asw[m]) << 4));
#pragma unroll
        for (int n = 0; n < 2; ++n) fb[set][n] = *(const int4_t *)(sb + brow[n] + ((c ^ bsw[n]) << 4));
    };
    auto mfma_step = [&](int set) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[set][m], fb[set][n], acc[m][n], 0, 0, 0);
    };
    auto interleave6 = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const bool next = kt + 1 < nk;
        if (next) issue(kt + 1);   // the other buffer: every wave left it at the previous barrier, its reads retired
        read_frags(1, 1, kt);
        mfma_step(0);
        interleave6();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, 2, kt);
        mfma_step(1);
        interleave6();
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, 3, kt);
        mfma_step(0);
        interleave6();
        __builtin_amdgcn_sched_barrier(0);
        if (next) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stage kt+1 landed; this wave's reads of stage kt retired
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_frags(0, 0, kt + 1);
        }
        mfma_step(1);
        if (next) interleave6();
        __builtin_amdgcn_sched_barrier(0);
    }
    int4_t cold[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) cold[t] = int4_t{0, 0, 0, 0};
    if (!a.first) {
#pragma unroll
        for (int t = 0; t < 8; ++t) cold[t] = *(const int4_t *)(Rb + RBOFF(t));
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int t = m * 2 + n;
            int4_t out;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned packed = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float oldf = (float)(((unsigned)cold[t][w] >> (8 * b)) & 0xffu);   // v_cvt_f32_ubyteN
                    const float r = rns_reduce_f(acc[m][n][w * 4 + b], oldf, p, inv_p, c4096, fold);
                    packed = __builtin_amdgcn_cvt_pk_u8_f32(r, b, packed);
                }
                out[w] = (int)packed;
            }
            *(int4_t *)(Rb + RBOFF(t)) = out;
        }
#undef RBOFF
}

static int g_rns_interleave = 1;
void rns_set_interleave(int v) { g_rns_interleave = v ? 1 : 0; }

void launch_rns_gemm256(hipStream_t s, const signed char *A, long lda, long a_plane, const signed char *B, long ldb,
                        long b_plane, signed char *R, int mt_all, int nt_all, int mt, int c0, int c1, int K, int first,
                        int tri) {
    if (mt <= 0 || c1 <= c0 || K <= 0) return;
    if (K > GP_RNS_KMAX || K % 128) {   // callers bound K (api.hip): sums of a longer contraction could leave the exact range
        fprintf(stderr, "gphip: residue contraction of %d bytes not launched\n", K);
        return;
    }
    RnsGemm256Args a;
    a.A = A; a.lda = lda; a.a_plane = a_plane; a.B = B; a.ldb = ldb; a.b_plane = b_plane; a.R = R;
    a.mt_all = mt_all; a.nt_all = nt_all; a.mt = mt; a.c0 = c0; a.c1 = c1; a.K = K; a.first = first; a.tri = tri; a.xcd_interleave = g_rns_interleave;
    a.sr = (mt + 7) / 8;
    a.sc = (c1 - c0 + 3) / 4;
    const long nwg = ((long)RNS_T * a.sr * a.sc + 7) / 8 * 8 * 32;   // whole rounds of eight super-tiles (one per XCD)
    GP_LAUNCH(rns_gemm256_kernel, dim3((unsigned)nwg), dim3(512), 0, s, a);
}

// reconstruction from the 64 KB blocks: one workgroup (512 threads, the GEMM's thread layout) per (256-row tile,
// 256-column tile); only the 128-column tiles c0_128 .. c1_128 are touched, rows below `rows` only
__global__ __launch_bounds__(512) void rns_reconstruct256_kernel(const signed char *R, int mt_all, int nt_all, int tc0, int ntc,
                                                                 int c0_128, int c1_128, long rows, double *T, long ldt,
                                                                 double out_scale, int tri) {
    const int blk = blockIdx.x >> 3;   // eight workgroups per 64 KB block: one MFMA tile of each wave's eight per workgroup
    const int ti = blk / ntc, tc = tc0 + blk % ntc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int col128 = (tc * 256 + wn * 64) >> 7;
    if (col128 < c0_128 || col128 >= c1_128) return;   // wave-uniform
    if (tri && ti < tc) return;                        // block never written (lower block triangle only)
    const long plane = (long)mt_all * nt_all * 65536;
    const signed char *Rb = R + ((long)ti * nt_all + tc) * 65536 + lane * 16;
    const double MAGIC = 0x1.8p8;  // 1.5 * 2^(52-44): rounds to multiples of 2^-44
    {
        const int t = blockIdx.x & 7;
        const int m = t >> 1, n = t & 1;
        const int sub = ((wm * 4 + m) * 8 + wn * 2 + n) * 1024;   // sub-block (rb, cb) = (wm*4 + m, wn*2 + n)
        const long row0 = (long)ti * 256 + wm * 128 + m * 32 + 4 * (lane >> 5);
        if ((long)ti * 256 + wm * 128 + m * 32 >= rows) return;   // wave-uniform
        // everything this tile needs is requested before any of it is used: the 14 residue vectors and the 16 fp64
        // values of T to be updated (one batch of loads instead of 32 dependent round trips)
        int4_t v[RNS_T];
#pragma unroll
        for (int l = 0; l < RNS_T; ++l) v[l] = *(const int4_t *)(Rb + l * plane + sub);
        const long col = (long)tc * 256 + wn * 64 + n * 32 + (lane & 31);
        double told[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + (r & 3) + 8 * (r >> 2);
            told[r] = row < rows ? T[row * ldt + col] : 0.0;
        }
        double H[16], L[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) H[r] = L[r] = 0.0;
#pragma unroll
        for (int l = 0; l < RNS_T; ++l) {
            const float p = c_rns.p[l], inv_p = c_rns.inv_p[l];
            const int q = c_rns.q[l];
            const double ih = c_rns.ih[l], il = c_rns.il[l];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int res = (int)(((unsigned)v[l][r >> 2] >> (8 * (r & 3))) & 0xffu);   // any representative in [0, 255]
                const float wf0 = (float)(res * q);                 // r q < 2^16
                const float k = rintf(wf0 * inv_p);
                const double w = (double)fmaf(-k, p, wf0);          // representative of r q mod p, |w| <= p
                const double t1 = w * ih;
                const double hi = (t1 + MAGIC) - MAGIC;
                double lo = fma(w, ih, -hi);
                lo = fma(w, il, lo);
                H[r] += hi;
                L[r] += lo;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double f = (H[r] - rint(H[r])) + L[r];
            f -= rint(f);
            const long row = row0 + (r & 3) + 8 * (r >> 2);
            if (row < rows) T[row * ldt + col] = told[r] - f * out_scale;
        }
    }
}

void launch_rns_reconstruct256(hipStream_t s, const signed char *R, int mt_all, int nt_all, int mt, int c0_128, int c1_128,
                               long rows, double *T, long ldt, double scale_2e, int tri) {
    if (mt <= 0 || c1_128 <= c0_128) return;
    long double Pl = 1.0L;
    for (int l = 0; l < RNS_T; ++l) Pl *= (long double)h_moduli[l];
    const double P_scaled = (double)(Pl * 0x1p-104L);
    const int tc0 = c0_128 / 2, tc1 = (c1_128 + 1) / 2;
    GP_LAUNCH(rns_reconstruct256_kernel, dim3((unsigned)(8 * mt * (tc1 - tc0))), dim3(512), 0, s, R, mt_all, nt_all, tc0,
                       tc1 - tc0, c0_128, c1_128, rows, T, ldt, P_scaled * scale_2e, tri);
}

// Shared declarations of the api_*.hip translation units: the context behind gp_t, error helpers, and the internal
// entry points each unit offers the others.  (include/gphip.h is the public C ABI; gphip_internal.h the kernels.)
#pragma once
#include "gphip_internal.h"
#include "../../include/gphip.h"

#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(GP_ERR_HIP, "%s -> %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NCCLCHK(x)                                                                                   \
    do {                                                                                             \
        ncclResult_t r_ = (x);                                                                       \
        if (r_ != ncclSuccess) return fail(GP_ERR_RCCL, "%s -> %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
// Drain a stream, then report (and clear) what gp_note_hip recorded since the last report: the point where an entry point's
// results are about to be read.
int gp_pending_error();
#define GP_SYNC(stream)                                                                            \
    do {                                                                                            \
        hipError_t e_ = hipStreamSynchronize(stream);                                               \
        int p_ = gp_pending_error();                                                                \
        if (p_) return p_;                                                                          \
        if (e_ != hipSuccess) return fail(GP_ERR_HIP, "hipStreamSynchronize(%s) -> %s (%s:%d)", #stream, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define GP_ERR_RANGE (-1000)   // internal: an operand of the residue path left the fixed-point range (the caller repeats in fp64)
void gp_clear_stale_note();
#define GP_DEAD_CHECK(g)                                                                            \
    do {                                                                                            \
        gp_clear_stale_note();                                                                      \
        if ((g)->dead)                                                                              \
            return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one"); \
    } while (0)

struct Phase {
    const char *name;
    hipEvent_t e0, e1;
    double flops, bytes;
    bool used;
};
#define MAX_PHASES 16

struct gp_ctx {
    int device = 0;
    hipStream_t s = nullptr;       // main stream
    hipStream_t s_panel = nullptr; // look-ahead (panel chain) stream: high priority, all CUs
    hipStream_t s_bulk = nullptr;  // trailing-update stream of the look-ahead Cholesky: masked off the reserved CUs
    int bulk_reserved = -1;        // reserved-CU count s_bulk was created with
    hipStream_t s_inv = nullptr, s_pred = nullptr;  // pipelined candidate solve (gp_fit_predict), low priority
    // stream-ordering events of the look-ahead factorisation, one dense vector per role (EV_* below)
    std::vector<hipEvent_t> la_events[6];
    bool ev_error = false;         // a stream-ordering event could not be created (la_event); checked by la_events_ok
    // data
    long N = 0, Npad = 0;
    int D = 0, P = 0;
    double *dX = nullptr, *dY = nullptr;
    double *dA = nullptr;     // (Npad + 128) x Npad: Ky / L (lower) and, below it, the RHS rows (Y^T -> z^T)
    double *dInvL = nullptr;  // nt tiles of 128 x 128: inverted diagonal tiles of L
    double *dAlpha = nullptr; // P x Npad
    double *dW = nullptr;     // P x Npad workspace
    double *dMu = nullptr;    // (1 + TM_SPLIT) * N : training mean + partials
    int *dInfo = nullptr;
    double *dScal = nullptr;  // small scalars: [0] logdet, [8..8+P) sumsq / dot
    double *dRedV = nullptr;  // 512 doubles of reduction scratch
    long long *dRedI = nullptr;
    long capN = 0;
    int capP = 0;
    // params
    KernParams kp{};
    int ard = 0;
    double noise = 0.0;
    bool have_data = false, have_params = false, fitted = false;
    double jitter = 0.0, lml = 0.0, logdet = 0.0;
    bool fmin_valid = false;
    double fmin = 0.0;
    // candidates
    long M = 0;
    double *dXs = nullptr;
    long capM = 0;
    double *dT = nullptr;  // Mc_pad x Npad
    long capT = 0;         // elements
    double *dMean = nullptr, *dVar = nullptr, *dAcq = nullptr;
    long capOut = 0;
    bool predicted = false;
    int predicted_noise = -1;
    // Wi
    double *dWi = nullptr;
    long capWi = 0;
    bool wi_valid = false;
    double *dT2 = nullptr;   // solved candidate rows S = K(Xs,X) L^-T (the running right-hand side stays in dT)
    long capT2 = 0;
    bool w_in_t2 = false;    // dT2 still holds W = L^-T of the current factor (ensure_linv / ensure_wi)
    // the explicit inverse factor and the scratch of the fused one-row path (onerow.hip, api_rows.hip)
    double *dLi = nullptr;   // L^-1, lower triangular, Npad x Npad row-major, zeros above the diagonal
    long capLi = 0;
    bool li_valid = false;
    double *dRows = nullptr; // RowsWork partials
    long capRows = 0;
    unsigned int *dRowsCounter = nullptr;
    double *hRowsOut = nullptr;          // pinned, device-visible result block of the fused path (+ the ticket behind it)
    double rows_ticket = 0.0;            // counts the fused passes; the finishing workgroup writes it back
    unsigned int rows_counter_base = 0;  // arrivals the counter holds from the passes before this one
    std::vector<double> lp_cache;        // local-penalisation batch as last uploaded (Xb | r | s), skipped when unchanged
    int lp_cache_nb = -1;
    long rows_fused_calls = 0, rows_fallback_calls = 0;
    long rows_calls_since_fit = 0;       // *_rows calls that wanted the inverse factor since the last fit (build policy, api_rows.hip)
    int rows_build = -1;                 // inverse factor of the one-location path: -1 by the rule of api_rows.hip, 0 never, 1 at the first call (option "rows_build")
    int rows_nt = -1;                    // fused one-row path: non-temporal loads of the inverse factor (option "rows_nt"; -1: when its
                                         // lower triangle exceeds the 256 MiB Infinity Cache, N > 8192 -- measured -10 % at N = 16384,
                                         // +10 % at N = 4096 where the next call finds the factor cached: profiles/r05_small_calls.txt)
    double *dLp = nullptr;   // local-penalisation batch (centres, radii, scales)
    double *dX2 = nullptr, *dK2 = nullptr;  // gp_cross_kernel_matrix: second input set and K(X, X2)
    long capX2 = 0, capK2 = 0;
    long capLp = 0;
    double *dCov = nullptr;  // full covariance / beta scratch
    long capCov = 0;
    double *dInvP = nullptr, *dInvPw = nullptr;  // inverted diagonal panels L_JJ^-1 (+ build workspace)
    long capInvP = 0, capInvPw = 0;
    int invp_W = 0;
    bool invp_valid = false;
    double *dDm = nullptr, *dDv = nullptr, *dDacq = nullptr;
    long capD = 0;
    // options
    int panel_tiles = 6;
    int lookahead = 1;
    int inner_min_rows = 0;         // ... only while at least this many row tiles lie below the pair (below that the 128-column step's shorter launches win)
    // look-ahead factorisation, columns owned by the chain stream (factor_lookahead): the bulk stream keeps own_keep_base + own_keep_per_row * n
    // tiles of a trailing update with n row tiles below the look-ahead panel -- what lasts as long as the chain is busy with that panel --
    // and the rest, the far columns, is updated on the chain stream once the panel is done (every CU).  own_keep_per_row = 0: off.
    // Defaults from the sweep of round 4 (N = 8192 ... 32768, profiles/r04_own_columns.txt)
    int own_keep_per_row = 36, own_keep_base = 200;
    int own_keep_pipe_pct = 0;      // ... scaled by this while pipelined candidate stages share the bulk stream's CUs (0: the owned range stops shrinking there; fused step -0.3 ms)
    int inner_tiles = 1;            // tile columns per step of the in-panel factorisation (2: potrf_pair_kernel + trsm2 + K = 256 update;
                                    // measured in round 4: the same wall time as 1 at every size, profiles/r04_pair_step_experiment.txt)
    int lookahead_min_tiles = 40;   // gp_fit: matrices of at most this many tiles (N <= 5120) take the single-stream factorisation
    int reserve_cus = 32;
    long mc_max = 16384;
    long small_m = 8;        // up to this many candidates take the matrix-vector solve (smallm.hip) instead of the tile path
    // profiling
    Phase phases[MAX_PHASES];
    int nphases = 0;
    bool profiling = false;
    int profile_class = 0;   // which kernel symbol gp_profile brackets: 0 = the 8-wave 128-tile update, 1 = the 64 x 64 work-unit update
    std::vector<hipEvent_t> gemm_events;
    std::vector<hipEvent_t> rns_events;   // gp_profile: start / end of every residue GEMM launch (rns_gemm256_kernel)
    size_t rns_ev_used = 0;
    double rns_ops = 0.0;                 // int8 multiply-adds x 2 of those launches
    std::vector<long> gemm_tiles;
    std::map<std::array<int, 5>, short *> tile_lists;  // cached L2-friendly tile orders (device)
    int supertile = 8;  // long rectangular / triangular launches walk 8 x 8 super-tiles per XCD (fabric traffic 5.35 -> 3.72 GB per launch, same time)
    int small_below = 1400;  // launches with fewer 128-tiles than this use 64x64 workgroup tiles
    int chain_small_below = 400;  // ... the same threshold for the launches of the factorisation's chain stream
    int lauum_panels = 1;    // Ky^-1 product accumulated per k-panel (0: one launch over the whole contraction)
    int side_alpha = 1;      // alpha / log det on the side stream while stages of the one-call entry points still run
    int pair_panels = 1;     // candidate solve: two panels per update launch (K = 2 x panel width), bitwise the same result
    int pair_tri = 2;        // triangular-K products: pair column tiles c and W-1-c in one workgroup (1: 64x64 units only)
    int fmin_direct = 0;     // gp_fmin through the N^2 product K(X,X) alpha instead of y - d alpha
    int trsm_rows64 = 32;    // in-place panel solves as 64- or 32-row strips of the tile (2 or 4 workgroups per tile)
    int waves8 = 1;
    int stagger = 3;  // see gemm.hip: odd-slot workgroups start 3 * 1024 cycles late (+1.5 % measured)
    int pipe_stages_grad = 0, pipe_start_pct_grad = 40;  // the same for gp_fit_grad (stages of the solve for L^-T)
    int pipe_stages = 0;         // gp_fit_predict: candidate stages that ride behind the factorisation (rest afterwards)
    int pipe_done = 0;           // ... how many did, in the last factorisation
    int pipe_start_pct = -1;     // ... released once this share of the panels is factored (the chain sets the pace from there); -1: 32 % up to 24 panels, 40 % beyond (measured N = 8192 ... 32768)
    std::vector<int> gemm_K;
    size_t gemm_ev_used = 0;
    long gemm_launches = 0;
    double gemm_flops = 0.0;      // flops of the event-bracketed launches
    double gemm_flops_all = 0.0;  // flops of every GEMM launch since gp_profile(1)
    long profile_min_tiles = 1024;
    // comm
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    double *dComm = nullptr;  // gather scratch of the top-k exchange
    long capComm = 0;
    // fp64 emulation on the int8 matrix cores (rns.hip)
    int emulate_fp64 = 0;
    int rns_group = 8; // panels per residue launch of the emulated candidate solve
    int rns_group_fit = 8;  // ... and of the emulated trailing update of the factorisation
    signed char *dLr = nullptr, *dSr = nullptr, *dRr = nullptr;  // residue planes of L, of the current S panel, accumulator
    signed char *dRm = nullptr;                                   // residue accumulator of the trailing matrix (factorisation)
    signed char *dWr = nullptr;                                   // residue planes of W = L^-T (emulated Ky^-1)
    long capWr = 0;
    long capLr = 0, capSr = 0, capRr = 0, capRm = 0;
    bool lr_valid = false;          // dLr belongs to the current factor
    std::vector<char> lr_done;      // ... per panel: rows below the panel's diagonal block converted
    int lr_W = 0, lr_e = 0;
    double jitter_try = 0.0;        // jitter of the factorisation attempt in progress (fixes the fixed-point scale)
    int emulate_fit = 1;            // emulate_fp64 also covers the factorisation's trailing update
    bool emu_off_call = false;      // this call fell back to true fp64 (an operand left the fixed-point range)
    long emu_fallbacks = 0;         // how often that happened
    bool dead = false;  // gp_shutdown ran: the device's streams are gone, only gp_destroy is still valid
};

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// the factorisation's trailing update runs in residue form (option emulate_fp64 with emulate_fit; panel edges on 256-column blocks)
static inline bool emu_fit_applies(const gp_ctx *g) {
    const long PB = (long)g->panel_tiles * GP_TILE;
    return g->emulate_fp64 && g->emulate_fit && !g->emu_off_call && (PB % 256 == 0) && PB <= GP_RNS_KMAX;
}

static inline GemmOpt inplace_opt() {
    GemmOpt o;
    o.inplace = 1;
    return o;
}

// Pipelined candidate solve (gp_fit_predict): as soon as panel J of L is final (chain(J) done), two more
// streams run, behind the factorisation and at low priority,
//   s_inv : invP_J = L_JJ^-1 (the per-panel build of ensure_panel_inv),
//   s_pred: S[:, J] = T[:, J] invP_J^T ;  T[:, > J] -= S[:, J] L[> J, J]^T
// so that the candidates' N^2 M flops fill the CUs the latency chain of the late panels leaves idle.
struct PredPipe {
    bool on = false;
    int mt = 0;          // candidate row tiles
    double *T = nullptr, *S = nullptr;
    std::function<void(hipStream_t)> init;  // fills T (cross covariance / identity) on the candidate stream, beside the factorisation's head
    bool trapezoid = false;  // T is block upper-triangular (the identity: the solve for L^-T), row tiles above the panel's end only
    int stages = 0, start_pct = 0;
};

// The candidate solve S = T L^-T with the running right-hand side's updates  T[:, > J] -= S_J L[> J, J]^T  carried in
// residue form on the int8 matrix cores (rns.hip; option "emulate_fp64").  Per panel J: the fp64 columns of T are
// rebuilt from the exact integer accumulator, S_J = T_J invP_J^T runs in fp64 as before (5 % of the flops), S_J is
// converted to residues and ONE int8 launch (14 moduli) applies it to every column to the right.
// Shared state of the residue paths: fixed-point scale, residue planes of L (zeroed padding), per-panel conversion.
struct RnsGeom {
    int e = 0;
    double scale = 1.0, back = 1.0;
    long Lrows = 0, Lplane = 0, Lpitch = 0;   // row pitch of the residue planes of L in bytes (= Npad, api_rns.hip)
    int nt256 = 0;
};

// trapezoid (the solve of the identity, for Ky^-1): row tiles beyond a panel's end are still zero, so every step of panel J
// covers the row tiles [0, J1) only; the solved panels' residues are KEPT, all side by side in planes of N columns (Wr, for
// the product W W^T afterwards), and S = L^-T has its own fixed-point scale: its rows have norm sqrt((Ky^-1)_ii) <=
// 1 / sqrt(noise + 1e-8 + jitter), which takes the place of sqrt(max diag Ky) in the bound of rns.hip.
struct RnsSolveOpt {
    bool trapezoid = false;
    int eS = -1;                 // exponent of S's scale (2^(eS-1) >= the largest row norm of S); < 0: that of L
    signed char *Wr = nullptr;   // full residue planes of S: Wrows x wpitch bytes per plane, zero beyond the written rows
    long wpitch = 0, wplane = 0;
};

// One set of HIP streams per device for the whole process, created once in a fixed order and never destroyed.
// Hardware queues are dealt over the command processor's pipes in creation order, and two queues on one pipe do
// not overlap (a 6000-workgroup dispatch holds the pipe until its last workgroup is issued).  Measured: a context
// created after an earlier one was closed, or a re-created bulk stream, put the chain and the trailing update on
// one pipe and the factorisation went from 34 to 45 ms.  Order here: main, chain, bulk, inverse, candidates
// -> pipes 0,1,2,3,0.
struct DevStreams {
    hipStream_t s = nullptr, panel = nullptr, bulk = nullptr, inv = nullptr, pred = nullptr;
    int reserved = -1;
};

enum { EV_CHAIN = 0, EV_BULK = 1, EV_INVP = 2, EV_MISC = 3, EV_FAR = 4, EV_CONV = 5 };  // chain(J) done, bulk(J) done, invP_J built,
                                                                                    // fork/join/side, far launch of group g done, residues of panel J written

// ---- internal entry points (defined in the api_*.hip unit named in the comment of each group) ----
void shutdown_all();
int make_bulk_stream(int device, int reserve, hipStream_t *out);
int get_streams(int device, int reserve, DevStreams *out);
int phase_begin(gp_ctx *g, const char *name, double flops, double bytes);
void phase_end(gp_ctx *g, int id);
void gemm(gp_ctx *g, hipStream_t s, int mode, double *C, long ldc, const double *A, long lda, const double *B, long ldb, int b_mul, int K, TileSet ts, const GemmOpt &o = GemmOpt());
void rns_gemm(gp_ctx *g, hipStream_t s, const signed char *A, long lda, long a_plane, const signed char *B, long ldb, long b_plane, signed char *R, int mt_all, int nt_all, int mt, int c0, int c1, int K, int first, int tri = 0);
int dev_realloc(double **p, long *cap, long need);
void destroy_ctx_events(gp_ctx *g);
void factor_buf(gp_ctx *g, double *A, long lda, int nt, int R1, double *invL, int *info, bool side_inv = false);
int factor(gp_ctx *g);
void build_panel_inv_one(gp_ctx *g, hipStream_t s, int J, int W, int nt);
int byte_realloc(signed char **p, long *cap, long need);
int rns_prepare(gp_ctx *g, double jitter, RnsGeom *r);
void rns_convert_panel(gp_ctx *g, hipStream_t s, const RnsGeom &r, int J, int *flag);
int factor_lookahead(gp_ctx *g, const PredPipe &pp = PredPipe());
int ensure_bulk_stream(gp_ctx *g);
hipEvent_t la_event(gp_ctx *g, int kind, size_t i);
int la_events_ok(gp_ctx *g);
int ensure_panel_inv(gp_ctx *g);
void solve_rows(gp_ctx *g, double *T, double *S, int mt, int trapezoid, int J_from = 0);
int solve_rows_rns(gp_ctx *g, double *T, double *S, int mt, const RnsSolveOpt &opt = RnsSolveOpt());
int fit_impl(gp_ctx *g, int maxtries, int pipe, int include_noise);
int run_predict(gp_ctx *g, int include_noise, bool tiles_only = false);   // tiles_only: never the small-M path (the caller uses dT2 as a padded tile operand)
int ensure_out(gp_ctx *g);
int run_acq(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std);
struct LpBatch { double *X = nullptr, *r = nullptr, *s = nullptr; };
int upload_lp_batch(gp_ctx *g, const double *Xb, int nb, const double *r0, const double *s0, LpBatch *b);
int run_acq_lp(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std, int transform, const double *Xb, int nb, const double *r0, const double *s0);
int wi_lauum(gp_ctx *g);
int wi_rns(gp_ctx *g);
int ensure_wi(gp_ctx *g);
int ensure_linv(gp_ctx *g);
int lml_grad_impl(gp_ctx *g, double *dvariance, double *dlengthscale, double *dnoise, bool reset_phases);
int ensure_grad_buffers(gp_ctx *g, long elemsBeta, long M);
int run_predict_grad(gp_ctx *g);

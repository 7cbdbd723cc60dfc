// C-ABI of libgphip.so and the host-side orchestration of the blocked algorithms.
// See include/gphip.h for the contract (reference file:line per entry point).
#include "gphip_internal.h"
#include "../../include/gphip.h"

#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <functional>
#include <array>
#include <map>
#include <mutex>
#include <set>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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
    double *dLp = nullptr;   // local-penalisation batch (centres, radii, scales)
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
    int reserve_cus = 32;
    long mc_max = 16384;
    // profiling
    Phase phases[MAX_PHASES];
    int nphases = 0;
    bool profiling = false;
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
    int inner_left_rows = 1 << 30;  // panels with at least this many row tiles update their columns left-looking
    int trsm_waves8 = 0;     // in-place panel solves on the 8-wave variant
    int trsm_rows64 = 32;    // in-place panel solves as 64- or 32-row strips of the tile (2 or 4 workgroups per tile)
    int panel_tiles_tail = 0, tail_rows = 0;  // narrower factorisation panels once fewer than tail_rows row tiles remain
    int waves8 = 1;
    int stagger = 3;  // see gemm.hip: odd-slot workgroups start 3 * 1024 cycles late (+1.5 % measured)
    int pipe_stages_grad = 0, pipe_start_pct_grad = 40;  // the same for gp_fit_grad (stages of the solve for L^-T)
    int pipe_stages = 0;         // gp_fit_predict: candidate stages that ride behind the factorisation (rest afterwards)
    int pipe_done = 0;           // ... how many did, in the last factorisation
    int pipe_start_pct = 40;     // ... released once this share of the panels is factored (the chain sets the pace from there)
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
    // fp64 emulation on the int8 matrix cores (rns.hip), candidate solve only
    // cooperative tail of the factorisation (potrf.hip: chol_tail_kernel)
    int tail_tiles = 0;      // trailing tile columns handed to ONE persistent launch (0 = off)
    int tail_wgs = 0;        // its workgroup count (0 = automatic)
    int ncu = 0;
    unsigned *dSync = nullptr;
    int emulate_fp64 = 0;
    int rns_group = 8; // panels per residue launch of the emulated candidate solve
    int rns_group_fit = 8;  // ... and of the emulated trailing update of the factorisation
    int rns_pad = 0;   // bytes added to the row pitch of L's residue planes (measured: no effect)
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

static int ensure_bulk_stream(gp_ctx *g);
enum { EV_CHAIN = 0, EV_BULK = 1, EV_INVP = 2, EV_MISC = 3, EV_FAR = 4, EV_CONV = 5 };  // chain(J) done, bulk(J) done, invP_J built,
                                                                                    // fork/join/side, far launch of group g done, residues of panel J written
static hipEvent_t la_event(gp_ctx *g, int kind, size_t i);

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
static std::mutex g_ds_mu;
static std::map<int, DevStreams> g_ds;
static std::set<gp_ctx *> g_live;  // contexts created and not yet destroyed
static bool g_atexit_registered = false;

static void destroy_ctx_events(gp_ctx *g);

// Ordered shutdown (exported as gp_shutdown, and registered with atexit() at the first stream creation so that it runs
// BEFORE the HIP runtime's and a profiler's own exit handlers, which were registered earlier): quiesce every device the
// library touched, destroy the events recorded on the shared streams, then the streams.  Without it the five
// process-lifetime queues per device -- two of them created with hipExtStreamCreateWithCUMask -- were still alive when
// the runtime's static destructors ran; under rocprofv3 the runtime's queue teardown then called into the already
// finalised tool and the process died with SIGSEGV inside __cxa_finalize (round 1: every profiled run after the
// per-device stream set was introduced; plain runs exited 0).  See DESIGN.md, "Lifecycle".
static void shutdown_all() {
    std::lock_guard<std::mutex> lk(g_ds_mu);
    for (auto &kv : g_ds) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        hipDeviceSynchronize();
    }
    for (gp_ctx *g : g_live) {
        hipSetDevice(g->device);
        destroy_ctx_events(g);
        g->s = g->s_panel = g->s_bulk = g->s_inv = g->s_pred = nullptr;
        g->dead = true;
    }
    for (auto &kv : g_ds) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        DevStreams &d = kv.second;
        for (hipStream_t *st : {&d.pred, &d.inv, &d.bulk, &d.panel, &d.s}) {
            if (*st) hipStreamDestroy(*st);
            *st = nullptr;
        }
    }
    g_ds.clear();
}
static void shutdown_atexit() { shutdown_all(); }

static int make_bulk_stream(int device, int reserve, hipStream_t *out) {
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, device));
    const int ncu = pr.multiProcessorCount;
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> mask(words, 0xffffffffu);
    // CU bits are dealt round-robin over the XCDs (measured: tools/micro/cumask.hip), so clearing the
    // lowest R bits reserves R/8 CUs on every XCD.
    for (int i = 0; i < reserve && i < ncu - 8; ++i) mask[i / 32] &= ~(1u << (i % 32));  // reserve may be large (half the chip)
    if (reserve > 0) {
        hipError_t e = hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data());
        if (e != hipSuccess) return fail(GP_ERR_HIP, "hipExtStreamCreateWithCUMask -> %s", hipGetErrorString(e));
    } else {
        HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    return 0;
}

static int get_streams(int device, int reserve, DevStreams *out) {
    std::lock_guard<std::mutex> lk(g_ds_mu);
    if (const char *e = getenv("GPHIP_RESERVE_CUS")) reserve = std::max(0, std::min(64, atoi(e)));
    DevStreams &d = g_ds[device];
    if (!d.s) {
        if (!g_atexit_registered) {
            atexit(shutdown_atexit);
            g_atexit_registered = true;
        }
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        HIPCHK(hipStreamCreateWithPriority(&d.s, hipStreamNonBlocking, lo));
        HIPCHK(hipStreamCreateWithPriority(&d.panel, hipStreamNonBlocking, hi));
        int rc = make_bulk_stream(device, reserve, &d.bulk);
        if (rc) return rc;
        d.reserved = reserve;
        HIPCHK(hipStreamCreateWithPriority(&d.inv, hipStreamNonBlocking, lo));
        // the pipelined candidate solve launches thousands of workgroups too: keep it off the reserved CUs as well,
        // or the chain's diagonal-tile workgroup (which needs an EMPTY CU) waits for a whole candidate update
        {
            int rp = reserve;
            if (const char *e = getenv("GPHIP_PRED_RESERVE")) rp = atoi(e);
            rc = make_bulk_stream(device, rp, &d.pred);
            if (rc) return rc;
        }
    }
    // never re-created: the replacement queue lands on another command-processor pipe (creation order), and
    // when that is the chain stream's pipe the two can no longer overlap
    *out = d;
    return 0;
}

static inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// ---- phase timing -----------------------------------------------------------------------------
static int phase_begin(gp_ctx *g, const char *name, double flops, double bytes) {
    if (g->nphases >= MAX_PHASES) return -1;
    Phase &p = g->phases[g->nphases];
    p.name = name;
    p.flops = flops;
    p.bytes = bytes;
    if (!p.used) {
        hipEventCreate(&p.e0);
        hipEventCreate(&p.e1);
        p.used = true;
    }
    hipEventRecord(p.e0, g->s);
    return g->nphases++;
}
static void phase_end(gp_ctx *g, int id) {
    if (id >= 0) hipEventRecord(g->phases[id].e1, g->s);
}

// ---- GEMM wrapper with accounting ---------------------------------------------------------------
static void gemm(gp_ctx *g, hipStream_t s, int mode, double *C, long ldc, const double *A, long lda, const double *B,
                 long ldb, int b_mul, int K, TileSet ts, const GemmOpt &o = GemmOpt()) {
    const long n = tileset_count(ts) * o.batch;
    if (n <= 0 || K <= 0) return;
    GemmOpt oo = o;
    if (n >= 1024 && !oo.stagger) oo.stagger = g->stagger;
    if (n >= 1024 && g->waves8) oo.waves8 = 1;  // 4 waves/SIMD: +2 % on the long launches (measured)
    // in-place panel solves are at most one workgroup per CU: eight waves hide the single tile's LDS/barrier latency
    if (oo.inplace && g->trsm_waves8 && n <= 512) oo.waves8 = 1;
    // ... or split each tile into two 64-row strips (a strip reads only its own rows of A: still safe in place)
    if (oo.inplace && g->trsm_rows64 && !oo.waves8) oo.rows64 = g->trsm_rows64;  // 1 / 64: two strips, 32: four
    // short launches (the factorisation's latency chain, the uneven triangular-K products) run as 64x64 work units
    const int small_thr = (s == g->s_panel) ? g->chain_small_below : g->small_below;
    if (small_thr > 0 && n < small_thr && !oo.inplace) oo.small = 1;
    // products with an inverted panel: column tile c contracts c+1 K-blocks; pairing c with W-1-c gives every
    // workgroup the same W+1 blocks (one balanced round of workgroups instead of a long and a short one)
    if (g->pair_tri && (oo.small || (oo.waves8 && g->pair_tri >= 2)) && o.k_end_tri && !ts.tri && !o.tile_list && ts.c1 - ts.c0 >= 2)
        oo.pair = 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // events only around the launches that carry the flops (>= 1024 output tiles): bracketing every one of the
    // ~700 small launches of an iteration stalls the latency chain (34 -> 53 ms per factorisation, measured)
    // ... and only the launches of ONE kernel symbol, gemm_nt_kernel<1, 128, 4, false, 128> (C -= A B^T, 8 waves), so that the
    // average agrees with that symbol's row in a rocprofv3 --stats summary of the same command
    const bool timed = g->profiling && n >= g->profile_min_tiles &&
                       (g->profile_min_tiles < 1024 || (mode == 1 && oo.waves8 && !oo.small));  // tracing tools lower the threshold
    if (timed) {
        if (g->gemm_ev_used + 2 > g->gemm_events.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            g->gemm_events.push_back(a);
            g->gemm_events.push_back(b);
        }
        e0 = g->gemm_events[g->gemm_ev_used++];
        e1 = g->gemm_events[g->gemm_ev_used++];
        hipEventRecord(e0, s);
        g->gemm_tiles.push_back(n);
        g->gemm_K.push_back((o.k_tri || o.k_end_tri) ? -K : K);
    }
    if (g->supertile > 1 && !o.tile_list && !o.k_end_tri && o.batch == 1 && tileset_count(ts) >= 2048) {
        const std::array<int, 5> key{ts.r0, ts.r1, ts.c0, ts.c1, ts.tri};
        auto it = g->tile_lists.find(key);
        if (it == g->tile_lists.end()) {
            std::vector<short> l = build_tile_list(ts, g->supertile);
            short *d = nullptr;
            if (hipMalloc((void **)&d, l.size() * sizeof(short)) == hipSuccess) {
                hipMemcpy(d, l.data(), l.size() * sizeof(short), hipMemcpyHostToDevice);
                it = g->tile_lists.emplace(key, d).first;
            }
        }
        if (it != g->tile_lists.end()) oo.tile_list = it->second;
    }
    launch_gemm_nt(s, mode, C, ldc, A, lda, B, ldb, b_mul, K, ts, oo);
    if (timed) {
        hipEventRecord(e1, s);
        g->gemm_launches++;
        g->gemm_flops += 2.0 * GP_TILE * GP_TILE * (double)K * (double)n * ((o.k_tri || o.k_end_tri) ? 0.5 : 1.0);
    }
    g->gemm_flops_all += 2.0 * GP_TILE * GP_TILE * (double)K * (double)n * ((o.k_tri || o.k_end_tri) ? 0.5 : 1.0);
}

// residue GEMM launch with the same accounting (gp_profile): events on the launch's own stream, int8 operations of the
// blocks the launch really computes (a triangular launch skips the blocks above the diagonal)
static void rns_gemm(gp_ctx *g, hipStream_t s, const signed char *A, long lda, long a_plane, const signed char *B, long ldb,
                     long b_plane, signed char *R, int mt_all, int nt_all, int mt, int c0, int c1, int K, int first, int tri = 0) {
    if (mt <= 0 || c1 <= c0 || K <= 0) return;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g->profiling) {
        if (g->rns_ev_used + 2 > g->rns_events.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            g->rns_events.push_back(a);
            g->rns_events.push_back(b);
        }
        e0 = g->rns_events[g->rns_ev_used++];
        e1 = g->rns_events[g->rns_ev_used++];
        long blocks = 0;
        for (int c = c0; c < c1; ++c) blocks += tri ? std::max(0, mt - c) : mt;
        g->rns_ops += 2.0 * 65536.0 * (double)K * (double)blocks * GP_RNS_T;
        hipEventRecord(e0, s);
    }
    launch_rns_gemm256(s, A, lda, a_plane, B, ldb, b_plane, R, mt_all, nt_all, mt, c0, c1, K, first, tri);
    if (e1) hipEventRecord(e1, s);
}

static inline GemmOpt inplace_opt() {
    GemmOpt o;
    o.inplace = 1;
    return o;
}

// ---- memory helpers -----------------------------------------------------------------------------
static int dev_realloc(double **p, long *cap, long need) {
    if (need <= *cap && *p) return 0;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)p, (size_t)need * sizeof(double));
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hipMalloc(%ld doubles) -> %s", need, hipGetErrorString(e));
    *cap = need;
    return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" {

const char *gp_last_error(void) { return g_err.c_str(); }
const char *gp_version(void) { return "gphip 0.1 (gfx950, fp64 MFMA)"; }

int gp_device_count(int *count) {
    if (!count) return fail(GP_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(GP_ERR_HIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    *count = n;
    return 0;
}

int gp_device_info(int device, char *name, int cap, int *cus, int64_t *hbm_bytes) {
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, device));
    if (name && cap > 0) {
        snprintf(name, cap, "%s (%s)", pr.name, pr.gcnArchName);
    }
    if (cus) *cus = pr.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)pr.totalGlobalMem;
    return 0;
}

int gp_create(gp_t **out, int device) {
    if (!out) return fail(GP_ERR_ARG, "out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(GP_ERR_ARG, "device %d out of range (%d visible)", device, n);
    HIPCHK(hipSetDevice(device));
    gp_ctx *g = new gp_ctx();
    g->device = device;
    for (int i = 0; i < MAX_PHASES; ++i) g->phases[i].used = false;
    {
        DevStreams d;
        int rcs = get_streams(device, g->reserve_cus, &d);
        if (rcs) {
            delete g;
            return rcs;
        }
        g->s = d.s;
        g->s_panel = d.panel;
        g->s_bulk = d.bulk;
        g->bulk_reserved = d.reserved;
        g->s_inv = d.inv;
        g->s_pred = d.pred;
    }
    {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) == hipSuccess) g->ncu = pr.multiProcessorCount;
    }
    hipError_t e = hipMalloc((void **)&g->dInfo, sizeof(int) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dSync, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dScal, sizeof(double) * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dRedV, sizeof(double) * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dRedI, sizeof(long long) * 1024);
    if (e != hipSuccess) {
        if (g->dInfo) hipFree(g->dInfo);
        if (g->dSync) hipFree(g->dSync);
        for (double *p : {g->dScal, g->dRedV})
            if (p) hipFree(p);
        if (g->dRedI) hipFree(g->dRedI);
        delete g;
        return fail(GP_ERR_HIP, "gp_create: hipMalloc -> %s", hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lk(g_ds_mu);
        g_live.insert(g);
    }
    // run an unmodified host program (e.g. the whole GPU test suite) with the emulated contractions on
    if (const char *ev = getenv("GPHIP_EMULATE_FP64")) g->emulate_fp64 = atoi(ev) ? 1 : 0;
    *out = g;
    return 0;
}

int gp_shutdown(void) {
    shutdown_all();
    return 0;
}

static void destroy_ctx_events(gp_ctx *g) {
    for (int i = 0; i < MAX_PHASES; ++i)
        if (g->phases[i].used) {
            hipEventDestroy(g->phases[i].e0);
            hipEventDestroy(g->phases[i].e1);
            g->phases[i].used = false;
        }
    g->nphases = 0;
    for (hipEvent_t e : g->gemm_events) hipEventDestroy(e);
    g->gemm_events.clear();
    for (hipEvent_t e : g->rns_events) hipEventDestroy(e);
    g->rns_events.clear();
    g->rns_ev_used = 0;
    g->gemm_ev_used = 0;
    g->gemm_tiles.clear();
    g->gemm_K.clear();
    for (auto &v : g->la_events) {
        for (hipEvent_t e : v) hipEventDestroy(e);
        v.clear();
    }
}

int gp_destroy(gp_t *g) {
    if (!g) return 0;
    {
        std::lock_guard<std::mutex> lk(g_ds_mu);
        if (!g_live.erase(g)) return 0;  // not a live context (double destroy)
    }
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    if (g->comm) ncclCommDestroy(g->comm);
    double *ptrs[] = {g->dX, g->dY, g->dA, g->dInvL, g->dAlpha, g->dW, g->dMu, g->dScal, g->dRedV,
                      g->dXs, g->dT, g->dMean, g->dVar, g->dAcq, g->dWi, g->dT2, g->dDm, g->dDv, g->dDacq, g->dCov, g->dInvP, g->dInvPw, g->dLp, g->dComm};
    for (double *p : ptrs)
        if (p) hipFree(p);
    if (g->dInfo) hipFree(g->dInfo);
    if (g->dSync) hipFree(g->dSync);
    if (g->dRedI) hipFree(g->dRedI);
    for (signed char *p : {g->dLr, g->dSr, g->dRr, g->dRm, g->dWr})
        if (p) hipFree(p);
    for (auto &kv : g->tile_lists) hipFree(kv.second);
    // events recorded on the shared streams go first; the streams themselves belong to the per-device set shared by
    // every context of the process and are destroyed by gp_shutdown / the exit hook
    destroy_ctx_events(g);
    delete g;
    return 0;
}

int gp_set_option(gp_t *g, const char *name, int64_t value) {
    if (!g || !name) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!strcmp(name, "panel_tiles")) {
        if (value < 1 || value > 64) return fail(GP_ERR_ARG, "panel_tiles out of range");
        g->panel_tiles = (int)value;
    } else if (!strcmp(name, "lookahead")) {
        g->lookahead = value ? 1 : 0;
    } else if (!strcmp(name, "pipe_start_pct")) {
        if (value < 0 || value > 100) return fail(GP_ERR_ARG, "pipe_start_pct out of range");
        g->pipe_start_pct = (int)value;
    } else if (!strcmp(name, "small_below")) {
        g->small_below = (int)value;
    } else if (!strcmp(name, "profile_min_tiles")) {
        if (value < 0) return fail(GP_ERR_ARG, "profile_min_tiles < 0");
        g->profile_min_tiles = value;
    } else if (!strcmp(name, "pipe_stages_grad")) {
        if (value < 0) return fail(GP_ERR_ARG, "pipe_stages_grad < 0");
        g->pipe_stages_grad = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "pipe_start_pct_grad")) {
        if (value < 0 || value > 100) return fail(GP_ERR_ARG, "pipe_start_pct_grad out of range");
        g->pipe_start_pct_grad = (int)value;
    } else if (!strcmp(name, "pipe_stages")) {
        if (value < 0) return fail(GP_ERR_ARG, "pipe_stages < 0");
        g->pipe_stages = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "lauum_panels")) {
        g->lauum_panels = (int)value;
        g->wi_valid = false;
    } else if (!strcmp(name, "side_alpha")) {
        g->side_alpha = (int)value;
    } else if (!strcmp(name, "pair_panels")) {
        g->pair_panels = value ? 1 : 0;
    } else if (!strcmp(name, "pair_tri")) {
        g->pair_tri = (int)value;
    } else if (!strcmp(name, "fmin_direct")) {
        g->fmin_direct = (int)value;
        g->fmin_valid = false;
    } else if (!strcmp(name, "chain_small_below")) {
        g->chain_small_below = (int)value;
    } else if (!strcmp(name, "waves8")) {
        g->waves8 = value ? 1 : 0;
    } else if (!strcmp(name, "stagger")) {
        g->stagger = (int)value;
    } else if (!strcmp(name, "supertile")) {
        if (value < 0 || value > 64) return fail(GP_ERR_ARG, "supertile out of range");
        g->supertile = (int)value;
    } else if (!strcmp(name, "reserve_cus")) {
        if (value != g->bulk_reserved)
            return fail(GP_ERR_ARG, "reserve_cus is fixed when the device's streams are created (%d); set GPHIP_RESERVE_CUS "
                                    "before the first gp_create", g->bulk_reserved);
    } else if (!strcmp(name, "panel_tiles_tail")) {
        if (value < 0 || value > 16) return fail(GP_ERR_ARG, "panel_tiles_tail out of range");
        g->panel_tiles_tail = (int)value;
    } else if (!strcmp(name, "tail_rows")) {
        g->tail_rows = (int)value;
    } else if (!strcmp(name, "inner_left_rows")) {
        g->inner_left_rows = (int)value;
    } else if (!strcmp(name, "trsm_rows64")) {
        g->trsm_rows64 = (int)value;
    } else if (!strcmp(name, "trsm_waves8")) {
        g->trsm_waves8 = (int)value;
    } else if (!strcmp(name, "tail_tiles")) {
        if (value < 0 || value > 4096) return fail(GP_ERR_ARG, "tail_tiles out of range");
        g->tail_tiles = (int)value;
    } else if (!strcmp(name, "tail_wgs")) {
        if (value < 0 || value > 1024) return fail(GP_ERR_ARG, "tail_wgs out of range");
        g->tail_wgs = (int)value;
    } else if (!strcmp(name, "rns_pad")) {
        if (value < 0 || value % 16) return fail(GP_ERR_ARG, "rns_pad must be a non-negative multiple of 16");
        g->rns_pad = (int)value;
        g->lr_valid = false;
        if (g->dLr) { hipFree(g->dLr); g->dLr = nullptr; g->capLr = 0; }
    } else if (!strcmp(name, "rns_interleave")) {
        rns_set_interleave((int)value);   // process-wide A/B switch of the residue GEMM's workgroup order (default 1)
    } else if (!strcmp(name, "rns_group_fit")) {
        if (value < 1 || value > 16) return fail(GP_ERR_ARG, "rns_group_fit must be in [1, 16]");
        g->rns_group_fit = (int)value;
    } else if (!strcmp(name, "rns_group")) {
        if (value < 1 || value > 16) return fail(GP_ERR_ARG, "rns_group must be in [1, 16]");
        g->rns_group = (int)value;
    } else if (!strcmp(name, "emulate_fit")) {
        g->emulate_fit = value ? 1 : 0;
    } else if (!strcmp(name, "emulate_fp64")) {
        g->emulate_fp64 = value ? 1 : 0;
        g->predicted = false;
    } else if (!strcmp(name, "mc_max")) {
        if (value < GP_TILE) return fail(GP_ERR_ARG, "mc_max < 128");
        g->mc_max = round_up(value, GP_TILE);
    } else
        return fail(GP_ERR_ARG, "unknown option %s", name);
    return 0;
}

int gp_synchronize(gp_t *g) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->s_panel));
    if (g->s_bulk) HIPCHK(hipStreamSynchronize(g->s_bulk));
    if (g->s_inv) HIPCHK(hipStreamSynchronize(g->s_inv));
    if (g->s_pred) HIPCHK(hipStreamSynchronize(g->s_pred));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_set_data(gp_t *g, const double *X, const double *Y, int64_t N, int D, int P) {
    if (!g || !X || !Y) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (N < 1 || D < 1 || D > GP_MAX_D || P < 1 || P > GP_MAX_RHS)
        return fail(GP_ERR_ARG, "bad shape N=%ld D=%d P=%d (D <= %d, P <= %d)", (long)N, D, P, GP_MAX_D, GP_MAX_RHS);
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->s));
    const long Npad = round_up(N, GP_TILE);
    if (Npad > g->capN || P > g->capP || !g->dA) {
        double **bufs[] = {&g->dX, &g->dY, &g->dA, &g->dInvL, &g->dAlpha, &g->dW, &g->dMu};
        for (double **b : bufs) {
            if (*b) hipFree(*b);
            *b = nullptr;
        }
        const long capN = Npad;
        const int capP = std::max(P, g->capP);
        HIPCHK(hipMalloc((void **)&g->dX, sizeof(double) * capN * GP_MAX_D));
        HIPCHK(hipMalloc((void **)&g->dY, sizeof(double) * capN * capP));
        HIPCHK(hipMalloc((void **)&g->dA, sizeof(double) * (capN + GP_MAX_RHS) * capN));
        HIPCHK(hipMalloc((void **)&g->dInvL, sizeof(double) * capN * GP_TILE));
        HIPCHK(hipMalloc((void **)&g->dAlpha, sizeof(double) * capN * capP));
        HIPCHK(hipMalloc((void **)&g->dW, sizeof(double) * capN * capP));
        HIPCHK(hipMalloc((void **)&g->dMu, sizeof(double) * capN * 16));
        g->capN = capN;
        g->capP = capP;
    }
    g->N = N;
    g->Npad = Npad;
    g->D = D;
    g->P = P;
    HIPCHK(hipMemcpyAsync(g->dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice, g->s));
    HIPCHK(hipMemcpyAsync(g->dY, Y, sizeof(double) * N * P, hipMemcpyHostToDevice, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    g->have_data = true;
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    g->kp.D = D;
    return 0;
}

// The fork's Gower kernel option (GPy/GPy/kern/src/stationary.py:61-65,116-135; lengthscales = variable ranges,
// GPyOpt/GPyOpt/core/task/space.py:351-362).  Only K is Gower: Kdiag stays `variance` and the gradient
// formulas stay Euclidean in the fork; the former is reproduced, the latter are refused (gp_lml_grad /
// gp_predict_grad return GP_ERR_STATE for a Gower model).
int gp_set_gower(gp_t *g, int enable, const int *is_discrete, const double *range) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data) return fail(GP_ERR_STATE, "gp_set_data first");
    if (enable && (!is_discrete || !range)) return fail(GP_ERR_ARG, "null argument");
    g->kp.gower = enable ? 1 : 0;
    for (int d = 0; d < g->D && enable; ++d) {
        g->kp.gdisc[d] = is_discrete[d] ? 1 : 0;
        g->kp.gdiv[d] = is_discrete[d] ? 1.0 : range[d];
        if (!is_discrete[d] && !(range[d] > 0.0)) return fail(GP_ERR_ARG, "range of dimension %d must be positive", d);
    }
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

int gp_set_params(gp_t *g, int kernel, int ard, double variance, const double *lengthscale, double noise) {
    if (!g || !lengthscale) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data) return fail(GP_ERR_STATE, "gp_set_data must precede gp_set_params");
    if (kernel != GP_KERNEL_RBF && kernel != GP_KERNEL_MATERN52) return fail(GP_ERR_ARG, "unknown kernel %d", kernel);
    g->kp.kernel = kernel;
    g->kp.D = g->D;
    g->kp.variance = variance;
    for (int d = 0; d < g->D; ++d) g->kp.ls[d] = ard ? lengthscale[d] : lengthscale[0];
    g->ard = ard ? 1 : 0;
    g->noise = noise;
    g->have_params = true;
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

// ---- blocked right-looking Cholesky (two-level: 128-column steps inside panel_tiles-wide panels) ----
// Workgroups of the cooperative tail: enough to spread a column's rank-128 update, never more than the CUs (every
// workgroup has to be resident: the kernel synchronises through grid barriers)
static int tail_workgroups(gp_ctx *g, int t0, int R1) {
    if (g->tail_wgs > 0) return std::min(g->tail_wgs, std::max(1, g->ncu));
    const long tiles = (long)(R1 - t0) * (R1 - t0) / 4;
    return (int)std::max<long>(1, std::min<long>(tiles, std::max(1, g->ncu)));
}

// A: nt x nt tiles (lower) plus R1 - nt extra row tiles that ride through the panel solves and updates (the RHS rows)
static void factor_buf(gp_ctx *g, double *A, long lda, int nt, int R1, double *invL, int *info) {
    const int W = g->panel_tiles;
    hipStream_t s = g->s;
    for (int J0 = 0; J0 < nt; J0 += W) {
        if (g->tail_tiles > 0 && nt - J0 <= g->tail_tiles) {   // the rest in one persistent launch
            launch_chol_tail(s, A, lda, invL, info, J0, nt, R1, g->dSync, tail_workgroups(g, J0, R1));
            return;
        }
        const int J1 = std::min(J0 + W, nt);
        for (int j = J0; j < J1; ++j) {
            launch_potrf_tile(s, A, lda, j, invL, info);
            // panel solve: A[i, j] <- A[i, j] * inv(L_jj)^T for the row tiles below (and the RHS tile)
            gemm(g, s, 0, A, lda, A + (long)j * GP_TILE, lda, invL + (long)j * GP_TILE * GP_TILE, GP_TILE, 0,
                 GP_TILE, TileSet{j + 1, R1, j, j + 1, 0}, inplace_opt());
            // update of the remaining columns of this panel (K = 128)
            if (j + 1 < J1)
                gemm(g, s, 1, A, lda, A + (long)j * GP_TILE, lda, A + (long)j * GP_TILE, lda, 1, GP_TILE,
                     TileSet{0, R1, j + 1, J1, 1});
        }
        // trailing update with the whole panel (K = W * 128): the dense contraction on MFMA
        if (J1 < nt)
            gemm(g, s, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, (J1 - J0) * GP_TILE,
                 TileSet{0, R1, J1, nt, 1});
    }
}

static void factor(gp_ctx *g) {
    const int nt = (int)(g->Npad / GP_TILE);
    factor_buf(g, g->dA, g->Npad, nt, nt + 1, g->dInvL, g->dInfo);
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

static void build_panel_inv_one(gp_ctx *g, hipStream_t s, int J, int W, int nt) {
    const long lda = g->Npad;
    const long PB = (long)W * GP_TILE;
    const int J0 = J * W, Wp = std::min(W, nt - J0);
    double *Wb = g->dInvPw + (long)J * PB * PB;
    const double *Lb = g->dA + (long)J * (PB * lda + PB);
    const double *Ib = g->dInvL + (long)J0 * GP_TILE * GP_TILE;
    launch_set_identity_blocks(s, Wb, PB, 1);
    for (int b = 0; b < Wp; ++b) {
        gemm(g, s, 0, Wb, PB, Wb + (long)b * GP_TILE, PB, Ib + (long)b * GP_TILE * GP_TILE, GP_TILE, 0, GP_TILE,
             TileSet{0, b + 1, b, b + 1, 0}, inplace_opt());
        if (b + 1 < Wp)
            gemm(g, s, 1, Wb, PB, Wb + (long)b * GP_TILE, PB, Lb + (long)b * GP_TILE, lda, 1, GP_TILE,
                 TileSet{0, b + 1, b + 1, Wp, 0});
    }
    launch_transpose_blocks(s, g->dInvP + (long)J * PB * PB, Wb, PB, 1);
}

static int byte_realloc(signed char **p, long *cap, long need) {
    if (need <= *cap && *p) return 0;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)p, (size_t)need);
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hipMalloc(%ld bytes) -> %s", need, hipGetErrorString(e));
    *cap = need;
    return 0;
}

// The candidate solve S = T L^-T with the running right-hand side's updates  T[:, > J] -= S_J L[> J, J]^T  carried in
// residue form on the int8 matrix cores (rns.hip; option "emulate_fp64").  Per panel J: the fp64 columns of T are
// rebuilt from the exact integer accumulator, S_J = T_J invP_J^T runs in fp64 as before (5 % of the flops), S_J is
// converted to residues and ONE int8 launch (14 moduli) applies it to every column to the right.
// Shared state of the residue paths: fixed-point scale, residue planes of L (zeroed padding), per-panel conversion.
struct RnsGeom {
    int e = 0;
    double scale = 1.0, back = 1.0;
    long Lrows = 0, Lplane = 0, Lpitch = 0;   // row pitch of the residue planes of L in bytes: NOT a power of two
    int nt256 = 0;
};

static int rns_prepare(gp_ctx *g, double jitter, RnsGeom *r) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = std::min(g->panel_tiles, nt);
    if ((long)W * GP_TILE > GP_RNS_KMAX) return fail(GP_ERR_ARG, "emulate_fp64: panel_tiles too wide for one residue contraction");
    if (g->N > (1L << 20)) return fail(GP_ERR_ARG, "emulate_fp64 needs N <= 2^20");
    if (rns_init_constants(g->device)) return fail(GP_ERR_HIP, "rns constants");
    r->Lrows = round_up(Npad, 256);
    // a row pitch of exactly Npad bytes (a power of two at the benchmark sizes) would put the 256 rows of an operand
    // tile's K slice on the same memory channel
    r->Lpitch = Npad + g->rns_pad;
    r->Lplane = r->Lrows * r->Lpitch;
    r->nt256 = (int)(r->Lrows / 256);
    // common power-of-two scale: |L_ij| <= sqrt(max diag of Ky), |S_ik| <= sqrt(prior variance); one spare bit
    const double diag0 = (g->kp.gower ? std::pow(g->kp.variance, g->D) : g->kp.variance) + g->noise + 1e-8 + jitter;
    int e = 1 + (int)std::ceil(std::log2(std::sqrt(std::max(diag0, 1e-300))));
    if (e < 0) e = 0;
    r->e = e;
    r->scale = std::ldexp(1.0, 52 - e);
    r->back = std::ldexp(1.0, 2 * e);
    const long need = (long)GP_RNS_T * r->Lplane;
    if (need > g->capLr || !g->dLr) {
        int rc = byte_realloc(&g->dLr, &g->capLr, need);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(g->dLr, 0, (size_t)need, g->s));
        HIPCHK(hipStreamSynchronize(g->s));
        g->lr_valid = false;
    }
    const int nJ = (nt + W - 1) / W;
    if (!g->lr_valid || g->lr_W != W || g->lr_e != e || (int)g->lr_done.size() != nJ) {   // (rns_pad frees dLr)
        g->lr_done.assign(nJ, 0);
        g->lr_W = W;
        g->lr_e = e;
        g->lr_valid = true;
    }
    return 0;
}

// residues of L's panel J (rows strictly below its diagonal block), once per factor
static void rns_convert_panel(gp_ctx *g, hipStream_t s, const RnsGeom &r, int J, int *flag) {
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE), W = g->lr_W;
    const int J0 = J * W, J1 = std::min(J0 + W, nt);
    if (J1 >= nt || g->lr_done[J]) return;
    launch_rns_convert(s, g->dA + (long)J1 * GP_TILE * lda + (long)J0 * GP_TILE, lda, Npad - (long)J1 * GP_TILE,
                       (long)(J1 - J0) * GP_TILE, g->dLr + (long)J1 * GP_TILE * r.Lpitch + (long)J0 * GP_TILE, r.Lplane, r.Lpitch,
                       r.scale, flag);
    g->lr_done[J] = 1;
}

static inline long Npad_rows(gp_ctx *g) { return g->Npad; }

static int factor_lookahead(gp_ctx *g, const PredPipe &pp = PredPipe()) {
    int rc;
    if ((rc = ensure_bulk_stream(g))) return rc;
    const long lda = g->Npad;
    const int nt = (int)(g->Npad / GP_TILE);
    const int R1 = nt + 1;
    const int W = g->panel_tiles;
    double *A = g->dA;
    hipStream_t sp = g->s_panel, sb = g->s_bulk;
    // fork
    hipEvent_t e0 = la_event(g, EV_MISC, 0);
    hipEventRecord(e0, g->s);
    hipStreamWaitEvent(sp, e0, 0);
    hipStreamWaitEvent(sb, e0, 0);
    const long PB = (long)W * GP_TILE;
    if (pp.on) {
        hipStreamWaitEvent(g->s_inv, e0, 0);
        hipStreamWaitEvent(g->s_pred, e0, 0);
        if (pp.init) pp.init(g->s_pred);
    }
    int next_pred = 0;
    const int nJu = (nt + W - 1) / W;
    // Only the first `pipe_stages` candidate stages ride behind the factorisation (on the CU-masked stream, released
    // at panel pred_start); the caller runs the rest on the main stream, on every CU, once the factor is complete.
    const int pstages = pp.on ? std::max(1, std::min(nJu, pp.stages)) : 0;
    const int pred_start = std::max(0, std::min(nJu - 1, nJu * pp.start_pct / 100));
    // panel boundaries: W tiles while the trailing matrix is tall; `panel_tiles_tail` once fewer than `tail_rows`
    // row tiles remain (the chain sets the pace there and a narrower panel means a shorter look-ahead update
    // between two chains).  The pipelined candidate solve needs the uniform panels its inverses are built on.
    std::vector<int> pb;
    for (int j = 0; j < nt;) {
        pb.push_back(j);
        const int w = (!pp.on && g->panel_tiles_tail > 0 && nt - j < g->tail_rows) ? g->panel_tiles_tail : W;
        j += w;
    }
    pb.push_back(nt);
    pb.push_back(nt);
    const int nJ = (int)pb.size() - 2;
    // "emulate_fp64": the trailing update (the launches of the bulk stream) in residue form on the int8 matrix cores
    // (rns.hip).  The Schur complement right of the look-ahead panel lives as Ky (untouched, in dA) minus an exact integer
    // accumulator dRm; a panel's columns are rebuilt in fp64 once, right before they become the look-ahead target.  The
    // chain (diagonal tiles, panel solves, in-panel and look-ahead updates) and the right-hand-side tile row stay fp64.
    const bool emu = g->emulate_fp64 && g->emulate_fit && !g->emu_off_call && (PB % 256 == 0) && PB <= GP_RNS_KMAX &&
                     !(g->panel_tiles_tail > 0) && !(g->tail_tiles > 0);
    RnsGeom rg;
    int *rflag = g->dInfo + 2;
    if (emu) {
        if ((rc = rns_prepare(g, g->jitter_try, &rg))) return rc;
        const long need = (long)GP_RNS_T * rg.nt256 * rg.nt256 * 65536;
        if ((rc = byte_realloc(&g->dRm, &g->capRm, need))) return rc;
        HIPCHK(hipMemsetAsync(rflag, 0, sizeof(int), g->s));
    }
    // emulated: panels per residue launch (the far launches ride on the otherwise idle candidate stream)
    const int Gf = (emu && !pp.on) ? std::max(1, std::min(g->rns_group_fit, (int)(GP_RNS_KMAX / PB))) : 1;
    hipStream_t sfar = g->s_pred;
    std::vector<char> far_issued(nJ / std::max(1, Gf) + 2, 0);
    if (Gf > 1) hipStreamWaitEvent(sfar, e0, 0);
    for (int J = 0; J < nJ; ++J) {
        const int J0 = pb[J], J1 = pb[J + 1], J2 = pb[J + 2];
        bool bulk_recorded = false;
        if (!pp.on && g->tail_tiles > 0 && nt - J0 <= g->tail_tiles) {
            // the trailing columns J0 .. nt-1 in one persistent launch on the chain stream: panel J's columns are
            // complete in stream order (look-ahead update J-1), everything right of them once bulk(J-1) is
            if (J >= 1 && J0 < nt) hipStreamWaitEvent(sp, la_event(g, EV_BULK, J - 1), 0);
            launch_chol_tail(sp, A, lda, g->dInvL, g->dInfo, J0, nt, R1, g->dSync, tail_workgroups(g, J0, R1));
            break;
        }
        // in-panel updates: left-looking while the panel is tall (column j receives columns J0..j-1 in ONE
        // contraction of K = 128 (j-J0): a third of the C traffic of j-J0 rank-128 updates and a longer K, which is
        // what counts while the chain shares the chip with the trailing update), right-looking once the panel is
        // short and the chain alone sets the pace (the rank-128 update is the shorter launch)
        const bool left = (R1 - J0) >= g->inner_left_rows;
        for (int j = J0; j < J1; ++j) {
            if (left && j > J0)
                gemm(g, sp, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, (j - J0) * GP_TILE,
                     TileSet{0, R1, j, j + 1, 1});
            launch_potrf_tile(sp, A, lda, j, g->dInvL, g->dInfo);
            gemm(g, sp, 0, A, lda, A + (long)j * GP_TILE, lda, g->dInvL + (long)j * GP_TILE * GP_TILE, GP_TILE, 0,
                 GP_TILE, TileSet{j + 1, R1, j, j + 1, 0}, inplace_opt());
            if (!left && j + 1 < J1)
                gemm(g, sp, 1, A, lda, A + (long)j * GP_TILE, lda, A + (long)j * GP_TILE, lda, 1, GP_TILE,
                     TileSet{0, R1, j + 1, J1, 1});
        }
        hipEvent_t eF = la_event(g, EV_CHAIN, J);
        hipEventRecord(eF, sp);
        const int K = (J1 - J0) * GP_TILE;
        if (pp.on) {
            if (J < pstages) {
                hipStreamWaitEvent(g->s_inv, eF, 0);
                build_panel_inv_one(g, g->s_inv, J, W, nt);
                hipEventRecord(la_event(g, EV_INVP, J), g->s_inv);
            }
            // Two concurrent MFMA-bound launches run slower than one after the other (measured 51 vs 63 TFLOP/s), and
            // the candidate stream is CU-masked like the trailing update (the diagonal-tile workgroup needs an empty
            // CU), which costs it 1/8 of the chip.  So only the first `pipe_stages` stages ride here, released once
            // the factorisation turns latency-bound (panel >= pred_start): they fill the CUs the chain leaves idle in
            // the tail.  The rest run after the join on the main stream, on every CU (fit_impl).  Measured at C3:
            // 73.4 ms against 77.0 for gp_fit + gp_predict; every stage pipelined: 78.1.
            if (J >= pred_start) {
                for (; next_pred <= J && next_pred < pstages; ++next_pred) {
                    const int Q = next_pred, Q0 = Q * W, Q1 = std::min(Q0 + W, nt);
                    const int KQ = (Q1 - Q0) * GP_TILE;
                    const int prow = pp.trapezoid ? std::min(pp.mt, Q1) : pp.mt;
                    hipStreamWaitEvent(g->s_pred, la_event(g, EV_CHAIN, J), 0);
                    hipStreamWaitEvent(g->s_pred, la_event(g, EV_INVP, Q), 0);
                    GemmOpt o;
                    o.k_end_tri = 1;
                    o.b_sub = Q0;
                    gemm(g, g->s_pred, 0, pp.S, g->Npad, pp.T + (long)Q0 * GP_TILE, g->Npad, g->dInvP + (long)Q * PB * PB,
                         PB, 1, KQ, TileSet{0, prow, Q0, Q1, 0}, o);
                    if (Q1 < nt)
                        gemm(g, g->s_pred, 1, pp.T, g->Npad, pp.S + (long)Q0 * GP_TILE, g->Npad, A + (long)Q0 * GP_TILE, lda,
                             1, KQ, TileSet{0, prow, Q1, nt, 0});
                }
            }
        }
        if (J1 >= nt) break;
        // the look-ahead update is on the critical path: enqueue it before the trailing update so that its
        // workgroups reach the dispatcher first once bulk(J-1) has drained
        if (J >= 1) hipStreamWaitEvent(sp, la_event(g, EV_BULK, J - 1), 0);
        // (emulated: the look-ahead panel's columns took everything the residue accumulator holds for them -- panels
        // 0 .. J-1 -- on the bulk stream, before bulk(J-1) was recorded)
        gemm(g, sp, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
             TileSet{0, R1, J1, J2, 1});
        if (J2 < nt) {
            hipStreamWaitEvent(sb, eF, 0);
            if (emu) {
                rns_convert_panel(g, sb, rg, J, rflag);
                // the right-hand-side tile row rides in fp64
                gemm(g, sb, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
                     TileSet{nt, R1, J2, nt, 0});
                auto rlaunch = [&](hipStream_t st, int Jfirst, int t0, int t1, int first) {   // panels Jfirst..J -> tiles [t0, t1)
                    t1 = std::min(t1, nt);
                    if (t0 >= t1) return;
                    const int T0 = pb[Jfirst];
                    rns_gemm(g, st, g->dLr + (long)T0 * GP_TILE, rg.Lpitch, rg.Lplane, g->dLr + (long)T0 * GP_TILE, rg.Lpitch,
                                       rg.Lplane, g->dRm, rg.nt256, rg.nt256, rg.nt256, t0 / 2, (t1 + 1) / 2, (J1 - T0) * GP_TILE,
                                       first, 1);
                };
                auto pbi = [&](int k) { return pb[std::min(k, nJ + 1)]; };
                // rebuild the columns of panel J+2 in fp64 (Ky minus everything accumulated for them: panels 0 .. J) as soon as
                // the last residue launch into them is enqueued -- on this stream, off the chain
                auto rebuild_next = [&]() {
                    if (pbi(J + 2) < nt)
                        launch_rns_reconstruct256(sb, g->dRm, rg.nt256, rg.nt256, rg.nt256, pbi(J + 2), std::min(pbi(J + 3), nt),
                                                  Npad_rows(g), A, lda, rg.back, 1);
                };
                if (Gf == 1) {
                    rlaunch(sb, J, J2, nt, J == 0 ? 1 : 0);
                    rebuild_next();
                } else {
                    // Panels in groups of Gf (all panel edges sit on 256-column accumulator blocks).  Pair (panel j, column
                    // panel c >= j+2; c = j+1 is the fp64 look-ahead) is served exactly once, by
                    //   near(J)  on the bulk stream, every iteration: the group's panels so far -> the columns of panel J+2,
                    //   mid(g)   on the bulk stream, at the group's last panel: the whole group -> the next Gf column panels,
                    //   far(g)   on a stream of its own: the whole group -> everything right of that,
                    // so the accumulator makes one round trip per group for the far columns and the long launch (K = Gf PB)
                    // overlaps the next group's chain.  Ordering: near(J) and mid(g) accumulate into blocks far(g-1) / far(g-2)
                    // wrote (mid waits for far(g-1); near follows mid(g-1) in stream order); far(g) follows far(g-1) in stream
                    // order; the chain's reconstruction of panel J+1's columns waits for bulk(J-1) = near(J-1), recorded
                    // BEFORE mid so that the chain does not wait for it.  The integers summed are those of Gf = 1.
                    const int gi = J / Gf, Jg = gi * Gf;
                    const int first = gi == 0 ? 1 : 0;
                    hipEventRecord(la_event(g, EV_CONV, J), sb);
                    rlaunch(sb, Jg, pbi(J + 2), pbi(J + 3), first);
                    rebuild_next();
                    hipEventRecord(la_event(g, EV_BULK, J), sb);
                    bulk_recorded = true;
                    if (J % Gf == Gf - 1) {
                        if (gi >= 1 && far_issued[gi - 1]) hipStreamWaitEvent(sb, la_event(g, EV_FAR, gi - 1), 0);
                        rlaunch(sb, Jg, pbi(J + 3), pbi(J + 3 + Gf), first);
                        if (pbi(J + 3 + Gf) < nt) {
                            hipStreamWaitEvent(sfar, la_event(g, EV_CONV, J), 0);
                            rlaunch(sfar, Jg, pbi(J + 3 + Gf), nt, first);
                            hipEventRecord(la_event(g, EV_FAR, gi), sfar);
                            far_issued[gi] = true;
                        }
                    }
                }
            } else {
                gemm(g, sb, 1, A, lda, A + (long)J0 * GP_TILE, lda, A + (long)J0 * GP_TILE, lda, 1, K,
                     TileSet{0, R1, J2, nt, 1});
            }
            if (!bulk_recorded) hipEventRecord(la_event(g, EV_BULK, J), sb);
        }
    }
    // join
    hipEvent_t ep = la_event(g, EV_MISC, 1), eb = la_event(g, EV_MISC, 2);
    hipEventRecord(ep, sp);
    hipEventRecord(eb, sb);
    hipStreamWaitEvent(g->s, ep, 0);
    hipStreamWaitEvent(g->s, eb, 0);
    if (Gf > 1) {   // (every far launch ends before the factor is complete: mid of the next group waits for it; join anyway)
        hipEvent_t ef = la_event(g, EV_MISC, 7);
        hipEventRecord(ef, sfar);
        hipStreamWaitEvent(g->s, ef, 0);
    }
    g->pipe_done = pstages;
    if (pp.on) {
        hipEvent_t eq = la_event(g, EV_MISC, 3), ei = la_event(g, EV_MISC, 4);
        hipEventRecord(eq, g->s_pred);
        hipEventRecord(ei, g->s_inv);
        hipStreamWaitEvent(g->s, eq, 0);
        hipStreamWaitEvent(g->s, ei, 0);
    }
    return 0;
}

// ---- the same factorisation with one panel of look-ahead on three streams ----------------------------
// s_panel (high priority, every CU): the latency chain of panel J -- potrf tile, panel solve, in-panel
//          updates -- then the update of panel J+1's columns with panel J (so that chain J+1 can start);
// s_bulk  (masked off `reserve_cus` CUs, which therefore stay free for the chain's single-workgroup potrf
//          kernel whose 148 KB of LDS needs an otherwise empty CU): the update of every column right of
//          panel J+1 with panel J -- the dense contraction, >90 % of the flops;
// s       : everything before and after.
// Ordering: bulk(J) after chain(J); look-ahead update(J) after bulk(J-1) (both touch panel J+1's columns);
// bulk(J) after bulk(J-1) (stream order).  Column sets of concurrent kernels are disjoint by construction.
static int ensure_bulk_stream(gp_ctx *g) {
    DevStreams d;
    int rc = get_streams(g->device, g->reserve_cus, &d);
    if (rc) return rc;
    g->s_bulk = d.bulk;
    g->bulk_reserved = d.reserved;
    return 0;
}

static hipEvent_t la_event(gp_ctx *g, int kind, size_t i) {
    std::vector<hipEvent_t> &v = g->la_events[kind];
    while (v.size() <= i) {
        hipEvent_t e;
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
        v.push_back(e);
    }
    return v[i];
}

// ---- inverted diagonal panels ---------------------------------------------------------------------
// invP_J = L_JJ^-1 for every panel J of W tiles (PB = W*128 rows), so that every triangular solve
// against L -- candidates (dtrtrs, posterior.py:294), alpha (dpotrs, exact_gaussian_inference.py:60),
// Ky^-1 (dpotri, linalg.py:127-145) -- is ONE product per panel on the MFMA GEMM instead of a chain of
// W dependent 128-column steps.  Built batched over all panels at once: the solve of the identity
// against L_JJ (2W-1 small launches, each covering every panel) gives L_JJ^-T, then one transpose.
static int ensure_panel_inv(gp_ctx *g) {
    if (g->invp_valid && g->invp_W == g->panel_tiles) return 0;
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = std::min(g->panel_tiles, nt);
    const long PB = (long)W * GP_TILE;
    const int nJ = (nt + W - 1) / W, nF = nt / W, Wl = nt % W;
    int rc;
    if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJ * PB * PB))) return rc;
    if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJ * PB * PB))) return rc;
    double *Wk = g->dInvPw;
    hipStream_t s = g->s;
    launch_set_identity_blocks(s, Wk, PB, nJ);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: the nF full panels as one batch; pass 1: the ragged last panel (Wl tiles)
        const int batch = pass == 0 ? nF : (Wl ? 1 : 0), Wp = pass == 0 ? W : Wl;
        if (batch == 0) continue;
        const long z0 = pass == 0 ? 0 : nF;
        double *Wb = Wk + z0 * PB * PB;
        const double *Lb = g->dA + z0 * (PB * lda + PB);
        const double *Ib = g->dInvL + z0 * (long)W * GP_TILE * GP_TILE;
        for (int b = 0; b < Wp; ++b) {
            GemmOpt o;
            o.batch = batch;
            o.inplace = 1;
            o.sC = o.sA = PB * PB;
            o.sB = (long)W * GP_TILE * GP_TILE;
            gemm(g, s, 0, Wb, PB, Wb + (long)b * GP_TILE, PB, Ib + (long)b * GP_TILE * GP_TILE, GP_TILE, 0, GP_TILE,
                 TileSet{0, b + 1, b, b + 1, 0}, o);
            if (b + 1 < Wp) {
                o.inplace = 0;
                o.sB = PB * lda + PB;
                gemm(g, s, 1, Wb, PB, Wb + (long)b * GP_TILE, PB, Lb + (long)b * GP_TILE, lda, 1, GP_TILE,
                     TileSet{0, b + 1, b + 1, Wp, 0}, o);
            }
        }
    }
    launch_transpose_blocks(s, g->dInvP, Wk, PB, nJ);
    g->invp_W = W;
    g->invp_valid = true;
    return 0;
}

// Row solve  S = T L^-T  for `mt` row tiles of T (row-major, ld = Npad); T is consumed as the running
// right-hand side.  trapezoid = 1: T is block upper-triangular (row tile r is zero left of column tile r:
// the identity, for L^-T), so panel J only touches the row tiles above its end.
static void solve_rows(gp_ctx *g, double *T, double *S, int mt, int trapezoid, int J_from = 0) {
    const long Npad = g->Npad, lda = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    const long PB = (long)W * GP_TILE;
    const double *L = g->dA;
    hipStream_t s = g->s;
    auto panel_solve = [&](int J, int J0, int J1, int rows) {
        GemmOpt o;
        o.k_end_tri = 1;
        o.b_sub = J0;
        // S[:, J] = T[:, J] invP_J^T   (invP_J lower triangular: column tile c contracts k <= c)
        gemm(g, s, 0, S, Npad, T + (long)J0 * GP_TILE, Npad, g->dInvP + (long)J * PB * PB, PB, 1, (J1 - J0) * GP_TILE,
             TileSet{0, rows, J0, J1, 0}, o);
    };
    for (int J0 = J_from * W, J = J_from; J0 < nt;) {
        const int J1 = std::min(J0 + W, nt), J2 = std::min(J1 + W, nt);
        const int Kp = (J1 - J0) * GP_TILE;
        const int rows = trapezoid ? std::min(mt, J1) : mt;
        panel_solve(J, J0, J1, rows);
        if (J1 >= nt) break;
        // Two panels per update (full row sets only): panel J+1's columns take panel J's update as a small launch of
        // their own, then ONE launch contracts both panels (K = 2 PB) into everything right of them -- half the round
        // trips of the running right-hand side through HBM and a contraction twice as long.  The accumulator sees the
        // same products in the same order as with one launch per panel: bitwise the same result.
        const bool two = g->pair_panels && !trapezoid && J2 > J1 && J2 < nt;
        if (!two) {
            // T[:, > J] -= S[:, J] L[> J, J]^T
            gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, Kp,
                 TileSet{0, rows, J1, nt, 0});
            J0 = J1;
            ++J;
            continue;
        }
        gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, Kp, TileSet{0, rows, J1, J2, 0});
        panel_solve(J + 1, J1, J2, rows);
        gemm(g, s, 1, T, Npad, S + (long)J0 * GP_TILE, Npad, L + (long)J0 * GP_TILE, lda, 1, (J2 - J0) * GP_TILE,
             TileSet{0, rows, J2, nt, 0});
        J0 = J2;
        J += 2;
    }
}

#define GP_ERR_RANGE (-1000)   // internal to this file: an operand of the residue path left the fixed-point range

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

static int solve_rows_rns(gp_ctx *g, double *T, double *S, int mt, const RnsSolveOpt &opt = RnsSolveOpt()) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    const long PB = (long)W * GP_TILE, Mcpad = (long)mt * GP_TILE;
    int rc;
    RnsGeom r;
    if (W != std::min(g->panel_tiles, nt)) return fail(GP_ERR_STATE, "emulate_fp64: panel width changed since the fit");
    if ((rc = rns_prepare(g, g->jitter, &r))) return rc;
    // 256 x 256 workgroup tiles: rows / columns padded to multiples of 256 (zero residues in the padding)
    const long Mc256 = round_up(Mcpad, 256);
    const int mt256 = (int)(Mc256 / 256), nt256 = r.nt256;
    // Panels are taken in groups of G (option "rns_group"): their S panels sit side by side in the residue buffer, the
    // columns of group panel i take ONE launch that contracts the i panels before it (K = i PB), and ONE launch then
    // contracts the whole group (K = G PB) into every column right of it.  Each accumulator block is therefore read,
    // reduced and written once per group instead of once per panel, and the contraction is G times as long; the
    // products summed are the same integers, so the result does not depend on G.  Panel edges must sit on 256-column
    // accumulator blocks for the launches of one group to touch disjoint blocks: odd panel widths take G = 1.
    int G = std::max(1, std::min(g->rns_group, (int)(GP_RNS_KMAX / PB)));
    if (W % 2) G = 1;
    const bool keep = opt.Wr != nullptr;
    const long KS = keep ? opt.wpitch : G * PB;   // row pitch of the S planes
    auto zalloc = [&](signed char **p, long *cap, long need) -> int {
        if (need <= *cap && *p) return 0;
        int r2 = byte_realloc(p, cap, need);
        if (r2) return r2;
        if (hipMemsetAsync(*p, 0, (size_t)need, g->s) != hipSuccess) return fail(GP_ERR_HIP, "hipMemsetAsync");
        return 0;
    };
    if (PB > GP_RNS_KMAX) return fail(GP_ERR_ARG, "emulate_fp64: panel_tiles too wide for one residue contraction");
    if (!keep && (rc = zalloc(&g->dSr, &g->capSr, (long)GP_RNS_T * Mc256 * KS))) return rc;
    if ((rc = zalloc(&g->dRr, &g->capRr, (long)GP_RNS_T * mt256 * nt256 * 65536))) return rc;
    hipStream_t s = g->s;
    int *flag = g->dInfo + 2;
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), s));
    const int eS = opt.eS >= 0 ? opt.eS : r.e;
    const double scaleS = std::ldexp(1.0, 52 - eS), back = std::ldexp(1.0, eS + r.e);
    const long Lplane = r.Lplane, Splane = keep ? opt.wplane : Mc256 * KS;
    signed char *Sr = keep ? opt.Wr : g->dSr;
    for (int J = 0; J < (int)g->lr_done.size(); ++J) rns_convert_panel(g, s, r, J, flag);
    // trapezoid: a launch's row blocks are those that hold a non-zero row of its S panels; blocks the accumulator has
    // never seen must read as zero, so it starts zeroed and no launch overwrites ("first")
    if (opt.trapezoid) HIPCHK(hipMemsetAsync(g->dRr, 0, (size_t)GP_RNS_T * mt256 * nt256 * 65536, s));
    auto rows_of = [&](int Jend) { return opt.trapezoid ? std::min(mt, Jend) : mt; };   // row tiles (128) of a step
    auto panel_solve = [&](int Ja, int Jb, int Jidx) {   // S[:, Ja..Jb) = T[:, Ja..Jb) invP^T, fp64
        GemmOpt o;
        o.k_end_tri = 1;
        o.b_sub = Ja;
        gemm(g, s, 0, S, Npad, T + (long)Ja * GP_TILE, Npad, g->dInvP + (long)Jidx * PB * PB, PB, 1, (Jb - Ja) * GP_TILE,
             TileSet{0, rows_of(Jb), Ja, Jb, 0}, o);
    };
    bool first = !opt.trapezoid;   // no launch has written the accumulator yet: the first group's launches overwrite their blocks
    for (int J0 = 0, J = 0; J0 < nt;) {
        int done = 0;    // panels of this group solved and converted; they span tiles [J0, Ja)
        bool last = false;
        for (int i = 0; i < G; ++i) {
            const int Ja = J0 + i * W, Jb = std::min(Ja + W, nt);
            if (Ja >= nt) break;
            const int rt = rows_of(Jb), rb = (rt + 1) / 2;                 // rows of this panel's steps: tiles, 256-blocks
            const long rrows = (long)rt * GP_TILE;
            // the group's earlier panels -> this panel's columns (whole 256-column blocks: Ja, Jb are even); their S rows
            // beyond tile Ja are zero, so are the products: the launch stops at the blocks that hold rows < Ja
            if (i > 0)
                rns_gemm(g, s, Sr + (keep ? (long)J0 * GP_TILE : 0), KS, Splane, g->dLr + (long)J0 * GP_TILE, r.Lpitch, Lplane,
                         g->dRr, mt256, nt256, opt.trapezoid ? (std::min(mt, Ja) + 1) / 2 : mt256, Ja / 2, (Jb + 1) / 2,
                         (Ja - J0) * GP_TILE, first ? 1 : 0);
            if (opt.trapezoid ? (J0 > 0 || i > 0) : (!first || i > 0))
                launch_rns_reconstruct256(s, g->dRr, mt256, nt256, opt.trapezoid ? rb : mt256, Ja, Jb,
                                          opt.trapezoid ? std::min(rrows, Mcpad) : Mcpad, T, Npad, back);
            panel_solve(Ja, Jb, J + i);
            done = i + 1;
            if (Jb >= nt) { last = true; break; }
            launch_rns_convert(s, S + (long)Ja * GP_TILE, Npad, opt.trapezoid ? rrows : Mcpad, (Jb - Ja) * GP_TILE,
                               Sr + (keep ? (long)Ja * GP_TILE : (long)i * PB), Splane, KS, scaleS, flag);
        }
        if (last) {
            // the last panel's residues are still wanted by the product W W^T
            if (keep) {
                const int Ja = J0 + (done - 1) * W, Jb = std::min(Ja + W, nt);
                launch_rns_convert(s, S + (long)Ja * GP_TILE, Npad, (long)rows_of(Jb) * GP_TILE, (Jb - Ja) * GP_TILE,
                                   Sr + (long)Ja * GP_TILE, Splane, KS, scaleS, flag);
            }
            break;
        }
        const int Jg = J0 + done * W;   // < nt here
        // the whole group -> every column right of it
        rns_gemm(g, s, Sr + (keep ? (long)J0 * GP_TILE : 0), KS, Splane, g->dLr + (long)J0 * GP_TILE, r.Lpitch, Lplane, g->dRr,
                 mt256, nt256, opt.trapezoid ? (std::min(mt, Jg) + 1) / 2 : mt256, G == 1 ? Jg / 2 : (Jg + 1) / 2, nt256,
                 (Jg - J0) * GP_TILE, first ? 1 : 0);
        if (!opt.trapezoid) first = false;
        J0 = Jg;
        J += done;
    }
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return GP_ERR_RANGE;   // (internal: the caller repeats the solve in true fp64)
    return 0;
}

__global__ void dot_ay_kernel(const double *alpha, long lda_, const double *Y, long N, int P, double *out) {
    __shared__ double sh[16];
    const int p = blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < N; i += 1024) s = fma(alpha[p * lda_ + i], Y[i * P + p], s);
    // block reduce
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < 16; ++i) r += sh[i];
        out[p] = r;
    }
}

static int ensure_out(gp_ctx *g);
static int wi_lauum(gp_ctx *g);

// Shared body of gp_fit and gp_fit_predict.  pipe != 0: the candidate solve of the resident candidates is
// pipelined behind the factorisation (PredPipe above) and the posterior reductions are appended.
// Any error return of fit_impl after work was forked onto the side streams must leave them joined: the guard waits for
// every stream of the context unless the normal exit (where the joins are stream-ordered) dismissed it.
struct QuiesceOnError {
    gp_ctx *g;
    bool armed = true;
    ~QuiesceOnError() {
        if (!armed) return;
        for (hipStream_t st : {g->s_panel, g->s_bulk, g->s_inv, g->s_pred, g->s})
            if (st) hipStreamSynchronize(st);
    }
};

static int fit_impl(gp_ctx *g, int maxtries, int pipe, int include_noise) {
    HIPCHK(hipSetDevice(g->device));
    QuiesceOnError guard{g};
    const long N = g->N, Npad = g->Npad, lda = g->Npad;
    const int P = g->P;
    const int nt_ = (int)(Npad / GP_TILE);
    // pipe 1: candidate solve of the resident candidates; pipe 2: the solve of the identity (L^-T, for Ky^-1)
    const long mcpad = pipe == 1 ? round_up(g->M, GP_TILE) : (pipe == 2 ? Npad : 0);
    PredPipe pp;
    if (pipe) {
        int rc;
        const int W = std::min(g->panel_tiles, nt_);
        const long PB = (long)W * GP_TILE;
        const int nJ = (nt_ + W - 1) / W;
        if (pipe == 1 && (rc = ensure_out(g))) return rc;
        if ((rc = dev_realloc(&g->dT, &g->capT, mcpad * Npad))) return rc;
        if ((rc = dev_realloc(&g->dT2, &g->capT2, mcpad * Npad))) return rc;
        if (pipe == 2 && (rc = dev_realloc(&g->dWi, &g->capWi, Npad * Npad))) return rc;
        if ((rc = dev_realloc(&g->dInvP, &g->capInvP, (long)nJ * PB * PB))) return rc;
        if ((rc = dev_realloc(&g->dInvPw, &g->capInvPw, (long)nJ * PB * PB))) return rc;
        pp.on = true;
        pp.mt = (int)(mcpad / GP_TILE);
        pp.T = g->dT;
        pp.S = g->dT2;
        pp.trapezoid = (pipe == 2);
        // 0 = automatic: the share of the panels that was best at N = 16384 (3 of 22 candidate stages, 8 of 22 L^-T stages)
        pp.stages = pipe == 2 ? (g->pipe_stages_grad > 0 ? g->pipe_stages_grad : std::max(1, (nJ * 36 + 50) / 100))
                              : (g->pipe_stages > 0 ? g->pipe_stages : std::max(1, (nJ * 14 + 50) / 100) + (nJ <= 12 ? 1 : 0));  // small N: the chain is everything
        pp.start_pct = pipe == 2 ? g->pipe_start_pct_grad : g->pipe_start_pct;
    }
    const double diag_add = g->noise + 1e-8;  // exact_gaussian_inference.py:56
    const double diag0 = (g->kp.gower ? std::pow(g->kp.variance, g->D) : g->kp.variance) + diag_add;
    g->nphases = 0;
    g->emu_off_call = false;
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;

    double jitter = 0.0;
    int tries = 0;  // number of jittered attempts so far
    int info = 0;
    for (;;) {
        g->lr_valid = false;     // residue planes of L belong to one factorisation attempt
        g->jitter_try = jitter;
        int ph = phase_begin(g, "kbuild", 0.0, 8.0 * N * g->D + 8.0 * (double)N * N / 2);
        launch_kbuild(g->s, g->dA, lda, g->dX, N, Npad, g->kp, diag_add, 0);
        // jitchol retries factor (Ky + jitter I): the jitter lands on the assembled diagonal (linalg.py:69)
        if (jitter != 0.0) launch_add_diag(g->s, g->dA, lda, N, jitter);
        launch_set_rhs(g->s, g->dA, lda, g->dY, N, Npad, P);
        phase_end(g, ph);
        HIPCHK(hipMemsetAsync(g->dInfo, 0, sizeof(int) * 4, g->s));
        if (pipe == 1) {
            pp.init = [g, mcpad, N, Npad](hipStream_t st) {
                launch_cross_k(st, g->dT, Npad, g->dXs, g->M, mcpad, g->dX, N, Npad, g->kp);
            };
            ph = phase_begin(g, "cholesky+cand_solve", (double)N * N * N / 3.0 + (double)N * N * g->M, 0.0);
            int rcf = factor_lookahead(g, pp);
            if (rcf) return rcf;
        } else if (pipe == 2) {
            ph = phase_begin(g, "cholesky+potri_stages", (double)N * N * N / 3.0, 0.0);
            pp.init = [g, Npad](hipStream_t st) { launch_set_identity(st, g->dT, Npad, Npad); };
            int rcf = factor_lookahead(g, pp);
            if (rcf) return rcf;
        } else {
            ph = phase_begin(g, "cholesky", (double)N * N * N / 3.0, 0.0);
            if (g->lookahead && Npad / GP_TILE > g->panel_tiles) {
                int rcf = factor_lookahead(g);
                if (rcf) return rcf;
            } else {
                factor(g);
            }
        }
        phase_end(g, ph);
        HIPCHK(hipMemcpyAsync(&info, g->dInfo, sizeof(int), hipMemcpyDeviceToHost, g->s));
        unsigned sync_words[2] = {0, 0};
        if (g->tail_tiles > 0) HIPCHK(hipMemcpyAsync(sync_words, g->dSync, sizeof sync_words, hipMemcpyDeviceToHost, g->s));
        HIPCHK(hipStreamSynchronize(g->s));
        if (sync_words[1] != 0)
            return fail(GP_ERR_HIP, "cooperative tail kernel: grid barrier timed out (a workgroup was not resident)");
        if (g->emulate_fp64 && !g->emu_off_call && info == 0) {
            int bad = 0;
            HIPCHK(hipMemcpy(&bad, g->dInfo + 2, sizeof(int), hipMemcpyDeviceToHost));
            if (bad) {
                // an entry of L outside the fixed-point range (non-finite data): the same attempt again in true fp64, whose
                // result is what the reference would return for such data
                g->emu_off_call = true;
                ++g->emu_fallbacks;
                g->nphases = 0;
                continue;
            }
        }
        if (info == 0) break;
        // jitter ladder, GPy/GPy/util/linalg.py:62-75
        if (!(diag0 > 0.0)) return fail(GP_ERR_NOT_PD_DIAG, "not pd: non-positive diagonal elements");
        if (tries == 0)
            jitter = diag0 * 1e-6;
        else
            jitter *= 10.0;
        ++tries;
        if (tries > maxtries || !std::isfinite(jitter)) {
            g_err = "not positive definite, even with jitter.";
            return info > 0 ? info : 1;
        }
        g->nphases = 0;
    }
    g->jitter = jitter;
    // alpha = L^-T z, log det and alpha'y: 45 short dependent launches (latency-bound, 1 ms).  When candidate / L^-T
    // stages are still to run on the main stream they go on the side stream instead, beside those long launches.
    bool side_alpha = false;
    auto alpha_lml = [&](hipStream_t st) {
        launch_logdet(st, g->dA, lda, N, g->dScal);
        launch_trsv_backward(st, g->dA, lda, g->dInvP, g->invp_W, Npad, g->dA + Npad * lda, lda, P, g->dAlpha, g->dW);
        hipLaunchKernelGGL(dot_ay_kernel, dim3(P), dim3(1024), 0, st, g->dAlpha, Npad, g->dY, N, P, g->dScal + 8);
    };
    if (pipe) {
        const int W = std::min(g->panel_tiles, nt_);
        const int nJ = (nt_ + W - 1) / W;
        if (g->pipe_done >= nJ) {  // every inverted panel was built by the pipeline
            g->invp_W = W;
            g->invp_valid = true;
        } else {  // the remaining stages on the main stream, every CU
            int phr = phase_begin(g, pipe == 2 ? "potri_solve_rest" : "cand_solve_rest", 0.0, 0.0);
            int rci = ensure_panel_inv(g);
            if (rci) return rci;
            if (g->s_inv && g->side_alpha) {
                hipEvent_t eI = la_event(g, EV_MISC, 5);
                hipEventRecord(eI, g->s);
                hipStreamWaitEvent(g->s_inv, eI, 0);
                alpha_lml(g->s_inv);
                hipEventRecord(la_event(g, EV_MISC, 6), g->s_inv);
                side_alpha = true;
            }
            solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), pipe == 2 ? 1 : 0, g->pipe_done);
            phase_end(g, phr);
        }
        if (pipe == 2) {
            int rcl = wi_lauum(g);
            if (rcl) return rcl;
            g->wi_valid = true;
        }
    }

    int ph = phase_begin(g, "alpha_lml", 2.0 * (double)N * N * P, 8.0 * (double)N * N / 2);
    if (side_alpha) {
        hipStreamWaitEvent(g->s, la_event(g, EV_MISC, 6), 0);
    } else {
        int rci = ensure_panel_inv(g);
        if (rci) return rci;
        alpha_lml(g->s);
    }
    phase_end(g, ph);
    if (pipe == 1) {
        ph = phase_begin(g, "reduce", 0.0, 8.0 * (double)N * g->M);
        launch_predict_reduce(g->s, g->dT2, Npad, g->M, N, g->dA + Npad * Npad, Npad, P, g->kp.variance,
                              include_noise ? g->noise : 0.0, g->dMean, g->dVar);
        phase_end(g, ph);
    }
    std::vector<double> sc(8 + P);
    HIPCHK(hipMemcpyAsync(sc.data(), g->dScal, sizeof(double) * (8 + P), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    double fit = 0.0;
    for (int p = 0; p < P; ++p) fit += sc[8 + p];
    g->logdet = sc[0];
    const double log_2_pi = std::log(2.0 * M_PI);
    g->lml = 0.5 * (-(double)N * P * log_2_pi - P * g->logdet - fit);  // exact_gaussian_inference.py:62
    g->fitted = true;
    if (pipe == 1) {
        g->predicted = true;
        g->predicted_noise = include_noise ? 1 : 0;
    }
    guard.armed = false;
    return 0;
}

int gp_fit(gp_t *g, int maxtries, double *lml, double *logdet, double *jitter_used) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit");
    int rc = fit_impl(g, maxtries, 0, 0);
    if (rc) return rc;
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    return 0;
}

static int run_predict(gp_ctx *g, int include_noise);

// gp_fit followed by gp_predict on the resident candidates, as ONE pipelined pass (the BO loop always runs
// them back to back: GPyOpt/GPyOpt/core/bo.py:236-254 then acquisitions/base.py:33-39).  Results are those of
// the two separate calls; the candidate solve merely overlaps the factorisation's latency-bound phases.
int gp_fit_predict(gp_t *g, int maxtries, int include_noise, double *lml, double *logdet, double *jitter_used,
                   double *mean, double *var) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit_predict");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const int nt = (int)(g->Npad / GP_TILE);
    // the emulated candidate solve runs after the factorisation (its residue planes of L need the complete factor)
    const bool can_pipe = g->lookahead && nt > g->panel_tiles && round_up(g->M, GP_TILE) <= g->mc_max && !g->emulate_fp64;
    int rc;
    if (can_pipe) {
        if ((rc = fit_impl(g, maxtries, 1, include_noise))) return rc;
    } else {
        if ((rc = fit_impl(g, maxtries, 0, 0))) return rc;
        if ((rc = ensure_out(g))) return rc;
        if ((rc = run_predict(g, include_noise))) return rc;
    }
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * g->M * g->P, hipMemcpyDeviceToHost, g->s));
    if (var) HIPCHK(hipMemcpyAsync(var, g->dVar, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_get_alpha(gp_t *g, double *alpha) {
    if (!g || !alpha) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    std::vector<double> tmp((size_t)g->P * g->Npad);
    HIPCHK(hipMemcpy(tmp.data(), g->dAlpha, sizeof(double) * g->P * g->Npad, hipMemcpyDeviceToHost));
    for (long i = 0; i < g->N; ++i)
        for (int p = 0; p < g->P; ++p) alpha[i * g->P + p] = tmp[(size_t)p * g->Npad + i];
    return 0;
}

int gp_get_chol(gp_t *g, double *L) {
    if (!g || !L) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    const long N = g->N;
    HIPCHK(hipMemcpy2D(L, sizeof(double) * N, g->dA, sizeof(double) * g->Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    for (long i = 0; i < N; ++i)
        for (long j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    return 0;
}

int gp_kernel_matrix(gp_t *g, double *K) {
    if (!g || !K) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params first");
    HIPCHK(hipSetDevice(g->device));
    const long N = g->N;
    launch_kbuild(g->s, g->dA, g->Npad, g->dX, N, g->Npad, g->kp, 0.0, 1);
    HIPCHK(hipStreamSynchronize(g->s));
    HIPCHK(hipMemcpy2D(K, sizeof(double) * N, g->dA, sizeof(double) * g->Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    g->fitted = false;  // dA was overwritten
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

// ---- candidates / predict -----------------------------------------------------------------------
int gp_set_candidates(gp_t *g, const double *Xs, int64_t M) {
    if (!g || !Xs) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data) return fail(GP_ERR_STATE, "gp_set_data first");
    if (M < 1) return fail(GP_ERR_ARG, "M < 1");
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(hipStreamSynchronize(g->s));
    int rc;
    if ((rc = dev_realloc(&g->dXs, &g->capM, (long)M * g->D))) return rc;
    HIPCHK(hipMemcpy(g->dXs, Xs, sizeof(double) * M * g->D, hipMemcpyHostToDevice));
    g->M = M;
    g->predicted = false;
    return 0;
}

static int run_predict(gp_ctx *g, int include_noise) {
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, N = g->N, Npad = g->Npad;
    const int P = g->P;
    int rc;
    g->nphases = 0;
    const long mc_max = std::min(g->mc_max, round_up(M, GP_TILE));
    if ((rc = ensure_panel_inv(g))) return rc;
    if ((rc = dev_realloc(&g->dT, &g->capT, mc_max * Npad))) return rc;
    if ((rc = dev_realloc(&g->dT2, &g->capT2, mc_max * Npad))) return rc;
    for (long m0 = 0; m0 < M; m0 += mc_max) {
        const long mc = std::min(mc_max, M - m0);
        const long mcpad = round_up(mc, GP_TILE);
        int ph = phase_begin(g, "cross_k", 0.0, 8.0 * (double)(N + mc) * g->D + 8.0 * (double)N * mc);
        launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, g->N, Npad, g->kp);
        phase_end(g, ph);
        ph = phase_begin(g, g->emulate_fp64 ? "cand_solve_emulated" : "cand_solve", (double)N * N * mc, 0.0);
        if (g->emulate_fp64) {
            rc = solve_rows_rns(g, g->dT, g->dT2, (int)(mcpad / GP_TILE));
            if (rc == GP_ERR_RANGE) {   // non-finite candidates / factor: this chunk again in true fp64 (NaNs propagate as in the reference)
                ++g->emu_fallbacks;
                launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, g->N, Npad, g->kp);
                solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), 0);
            } else if (rc) {
                return rc;
            }
        } else {
            solve_rows(g, g->dT, g->dT2, (int)(mcpad / GP_TILE), 0);
        }
        phase_end(g, ph);
        ph = phase_begin(g, "reduce", 0.0, 8.0 * (double)N * mc);
        launch_predict_reduce(g->s, g->dT2, Npad, mc, N, g->dA + Npad * Npad, Npad, P, g->kp.variance,
                              include_noise ? g->noise : 0.0, g->dMean + m0 * P, g->dVar + m0);
        phase_end(g, ph);
    }
    g->predicted = true;
    g->predicted_noise = include_noise ? 1 : 0;
    return 0;
}

static int ensure_out(gp_ctx *g) {
    const long need = g->M * (long)std::max(1, g->P);
    if (g->dMean && g->dVar && g->dAcq && g->capOut >= need) return 0;
    for (double **b : {&g->dMean, &g->dVar, &g->dAcq}) {
        if (*b) hipFree(*b);
        *b = nullptr;
    }
    HIPCHK(hipMalloc((void **)&g->dMean, sizeof(double) * need));
    HIPCHK(hipMalloc((void **)&g->dVar, sizeof(double) * need));
    HIPCHK(hipMalloc((void **)&g->dAcq, sizeof(double) * need));
    g->capOut = need;
    return 0;
}

int gp_predict(gp_t *g, int include_noise, double *mean, double *var) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise))) return rc;
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * g->M * g->P, hipMemcpyDeviceToHost, g->s));
    if (var) HIPCHK(hipMemcpyAsync(var, g->dVar, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_fmin(gp_t *g, double *fmin) {
    if (!g || !fmin) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->P != 1) return fail(GP_ERR_ARG, "gp_fmin needs P == 1");
    HIPCHK(hipSetDevice(g->device));
    if (!g->fmin_valid) {
        if (g->fmin_direct)
            launch_train_mean(g->s, g->dX, g->N, g->kp, g->dAlpha, g->dMu);
        else
            launch_train_mean_identity(g->s, g->dY, g->dAlpha, g->noise + 1e-8 + g->jitter, g->N, g->dMu);
        launch_argbest(g->s, g->dMu, g->N, -1, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
        double v = 0.0;
        HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
        HIPCHK(hipStreamSynchronize(g->s));
        g->fmin = v;
        g->fmin_valid = true;
    }
    *fmin = g->fmin;
    return 0;
}

static int run_acq(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std) {
    if (g->P != 1) return fail(GP_ERR_ARG, "acquisitions need P == 1");
    if (type < GP_ACQ_EI || type > GP_ACQ_MPI) return fail(GP_ERR_ARG, "unknown acquisition %d", type);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if (!g->predicted || g->predicted_noise != 1)
        if ((rc = run_predict(g, 1))) return rc;  // GPModel.predict: with_noise=True (gpmodel.py:102)
    launch_acq(g->s, type, par, fmin, y_mean, y_std, g->dMean, g->dVar, g->M, g->dAcq);
    return 0;
}

int gp_acq(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, double *out) {
    if (!g || !out) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_acq_argbest(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int sense, int64_t *idx,
                   double *val) {
    if (!g || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
    double v = 0.0;
    long long i = 0;
    HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(&i, g->dRedI + 256, sizeof(long long), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    *val = v;
    *idx = (int64_t)i;
    return 0;
}

// ---- local penalisation (batch acquisition of run.py:1238-1257; GPyOpt/GPyOpt/acquisitions/LP.py) -----------
static int run_acq_lp(gp_ctx *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
                      const double *Xb, int nb, const double *r0, const double *s0) {
    if (nb < 0 || nb > 256) return fail(GP_ERR_ARG, "batch size out of range (0..256)");
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    // small batch arrays live behind the reduction scratch
    int rcb;
    if ((rcb = dev_realloc(&g->dLp, &g->capLp, (long)256 * (GP_MAX_D + 2)))) return rcb;
    double *dXb = g->dLp, *dr = g->dLp + 256 * GP_MAX_D, *ds = dr + 256;
    if (nb > 0) {
        HIPCHK(hipMemcpyAsync(dXb, Xb, sizeof(double) * nb * g->D, hipMemcpyHostToDevice, g->s));
        HIPCHK(hipMemcpyAsync(dr, r0, sizeof(double) * nb, hipMemcpyHostToDevice, g->s));
        HIPCHK(hipMemcpyAsync(ds, s0, sizeof(double) * nb, hipMemcpyHostToDevice, g->s));
    }
    launch_lp(g->s, g->dAcq, g->dXs, g->M, g->D, dXb, nb, dr, ds, transform, g->dAcq);
    return 0;
}

int gp_acq_lp(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
              const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out) {
    if (!g || !out || (nb > 0 && (!Xb || !r_x0 || !s_x0))) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_lp(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0))) return rc;
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_acq_lp_argbest(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int transform,
                      const double *Xb, int nb, const double *r_x0, const double *s_x0, int sense,
                      const int64_t *exclude, int nex, int64_t *idx, double *val) {
    if (!g || !idx || !val || (nb > 0 && (!Xb || !r_x0 || !s_x0)) || (nex > 0 && !exclude))
        return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (nex < 0 || nex > 256) return fail(GP_ERR_ARG, "too many excluded rows (<= 256)");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq_lp(g, type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0))) return rc;
    if (nex > 0) {  // rows already taken never win (run.py:1249-1252 masks them)
        for (int i = 0; i < nex; ++i)
            if (exclude[i] < 0 || exclude[i] >= g->M) return fail(GP_ERR_ARG, "excluded row out of range");
        HIPCHK(hipMemcpyAsync(g->dRedI + 300, exclude, sizeof(long long) * nex, hipMemcpyHostToDevice, g->s));
        launch_mask(g->s, g->dAcq, g->dRedI + 300, nex, sense > 0 ? -INFINITY : INFINITY);
    }
    launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
    double v = 0.0;
    long long i = 0;
    HIPCHK(hipMemcpyAsync(&v, g->dRedV + 256, sizeof(double), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(&i, g->dRedI + 256, sizeof(long long), hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    *val = v;
    *idx = (int64_t)i;
    return 0;
}

// ---- measurement ----------------------------------------------------------------------------------
int gp_last_phases(gp_t *g, int cap, const char **names, double *ms, double *flops, double *bytes) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s);
    int n = std::min(cap, g->nphases);
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        hipEventElapsedTime(&t, g->phases[i].e0, g->phases[i].e1);
        if (names) names[i] = g->phases[i].name;
        if (ms) ms[i] = t;
        if (flops) flops[i] = g->phases[i].flops;
        if (bytes) bytes[i] = g->phases[i].bytes;
    }
    return n;
}

int gp_profile(gp_t *g, int on) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    g->profiling = on != 0;
    g->gemm_ev_used = 0;
    g->gemm_tiles.clear();
    g->gemm_K.clear();
    g->gemm_launches = 0;
    g->gemm_flops = 0.0;
    g->gemm_flops_all = 0.0;
    g->rns_ev_used = 0;
    g->rns_ops = 0.0;
    return 0;
}

int gp_gemm_stats(gp_t *g, int64_t *launches, double *ms, double *flops) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g->gemm_ev_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g->gemm_events[i], g->gemm_events[i + 1]) == hipSuccess) tot += t;
    }
    if (launches) *launches = g->gemm_launches;
    if (ms) *ms = tot;
    if (flops) *flops = g->gemm_flops;
    return 0;
}

// The same for the residue GEMM (rns_gemm256_kernel; option "emulate_fp64"): launches, summed durations and int8
// operations (2 per multiply-add) since gp_profile(1).
int gp_rns_stats(gp_t *g, int64_t *launches, double *ms, double *ops) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    hipSetDevice(g->device);
    for (hipStream_t st : {g->s_panel, g->s_bulk, g->s_inv, g->s_pred, g->s})
        if (st) hipStreamSynchronize(st);
    double tot = 0.0;
    for (size_t i = 0; i + 1 < g->rns_ev_used; i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g->rns_events[i], g->rns_events[i + 1]) == hipSuccess) tot += t;
    }
    if (launches) *launches = (int64_t)(g->rns_ev_used / 2);
    if (ms) *ms = tot;
    if (ops) *ops = g->rns_ops;
    return 0;
}

// Wall time during which at least one of the profiled launches was running (the union of their [start, end] intervals,
// measured against the first profiled launch's start event).  With overlapping launches (gp_fit_predict) the SUM of the
// durations counts shared time twice; flops / busy is the kernel's throughput while it runs.
int gp_gemm_busy(gp_t *g, double *busy_ms) {
    if (!g || !busy_ms) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    std::vector<std::pair<double, double>> iv;
    for (size_t i = 0; i + 1 < g->gemm_ev_used; i += 2) {
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, g->gemm_events[0], g->gemm_events[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&b, g->gemm_events[0], g->gemm_events[i + 1]) != hipSuccess) continue;
        iv.emplace_back((double)a, (double)b);
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0.0, cur_a = 0.0, cur_b = -1.0;
    for (auto &p : iv) {
        if (cur_b < cur_a || p.first > cur_b) {
            if (cur_b >= cur_a) busy += cur_b - cur_a;
            cur_a = p.first;
            cur_b = p.second;
        } else if (p.second > cur_b) {
            cur_b = p.second;
        }
    }
    if (cur_b >= cur_a) busy += cur_b - cur_a;
    *busy_ms = busy;
    return 0;
}

int gp_gemm_trace(gp_t *g, int cap, int64_t *tiles, int *K, double *ms) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    hipSetDevice(g->device);
    hipStreamSynchronize(g->s_panel);
    if (g->s_bulk) hipStreamSynchronize(g->s_bulk);
    if (g->s_inv) hipStreamSynchronize(g->s_inv);
    if (g->s_pred) hipStreamSynchronize(g->s_pred);
    hipStreamSynchronize(g->s);
    int n = (int)std::min<size_t>((size_t)cap, g->gemm_tiles.size());
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        hipEventElapsedTime(&t, g->gemm_events[2 * i], g->gemm_events[2 * i + 1]);
        if (tiles) tiles[i] = g->gemm_tiles[i];
        if (K) K[i] = g->gemm_K[i];
        if (ms) ms[i] = t;
    }
    return n;
}

// ---- multi-GPU --------------------------------------------------------------------------------------
int gp_comm_unique_id(char *uid128) {
    if (!uid128) return fail(GP_ERR_ARG, "null uid");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(uid128, &id, 128);
    return 0;
}

int gp_comm_init(gp_t *g, const char *uid128, int rank, int nranks) {
    if (!g || !uid128) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(GP_ERR_ARG, "bad rank %d / %d", rank, nranks);
    HIPCHK(hipSetDevice(g->device));
    if (g->comm) {
        ncclCommDestroy(g->comm);
        g->comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, uid128, 128);
    NCCLCHK(ncclCommInitRank(&g->comm, nranks, id, rank));
    g->rank = rank;
    g->nranks = nranks;
    return 0;
}

// what the communicator itself reports (ncclCommCount / ncclCommUserRank): the bench line carries these
int gp_comm_info(gp_t *g, int *rank, int *nranks) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    int r = -1, n = -1;
    NCCLCHK(ncclCommUserRank(g->comm, &r));
    NCCLCHK(ncclCommCount(g->comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return 0;
}

int gp_comm_destroy(gp_t *g) {
    if (!g) return 0;
    if (g->comm) {
        hipSetDevice(g->device);
        ncclCommDestroy(g->comm);
        g->comm = nullptr;
    }
    g->rank = 0;
    g->nranks = 1;
    return 0;
}

int gp_comm_allgather_best(gp_t *g, double val, int64_t idx, double *vals, int64_t *idxs) {
    if (!g || !vals || !idxs) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    if (g->nranks > 128) return fail(GP_ERR_ARG, "nranks > 128");
    HIPCHK(hipSetDevice(g->device));
    // one 16-byte record per rank: {double val, int64 idx} moved as 2 x 8 bytes
    double *send = g->dRedV + 300;       // 2 doubles
    double *recv = g->dRedV + 304;       // 2 * nranks doubles (<= 208 here: nranks <= 100)
    if (2 * g->nranks > 200) return fail(GP_ERR_ARG, "nranks too large for the gather scratch");
    double rec[2];
    rec[0] = val;
    memcpy(&rec[1], &idx, 8);
    HIPCHK(hipMemcpyAsync(send, rec, 16, hipMemcpyHostToDevice, g->s));
    NCCLCHK(ncclAllGather(send, recv, 2, ncclDouble, g->comm, g->s));
    std::vector<double> out(2 * g->nranks);
    HIPCHK(hipMemcpyAsync(out.data(), recv, 16 * g->nranks, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    for (int r = 0; r < g->nranks; ++r) {
        vals[r] = out[2 * r];
        memcpy(&idxs[r], &out[2 * r + 1], 8);
    }
    return 0;
}

int gp_comm_bcast_fit(gp_t *g, int root) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "every rank needs data and params set");
    HIPCHK(hipSetDevice(g->device));
    if (root < 0 || root >= g->nranks) return fail(GP_ERR_ARG, "root %d out of range", root);
    if (g->rank == root && !g->fitted) return fail(GP_ERR_STATE, "the root rank must be fitted");
    const long Npad = g->Npad;
    // host-side fit state rides along as a small record: a receiver's gp_fmin uses the root's jitter
    // (y - (noise + 1e-8 + jitter) alpha) and reports the root's LML / log det
    double rec[4] = {g->jitter, g->lml, g->logdet, 0.0};
    double *dRec = g->dScal + 400;
    if (g->rank == root) HIPCHK(hipMemcpyAsync(dRec, rec, sizeof rec, hipMemcpyHostToDevice, g->s));
    NCCLCHK(ncclGroupStart());
    NCCLCHK(ncclBroadcast(g->dA, g->dA, (size_t)(Npad + GP_MAX_RHS) * Npad, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(g->dInvL, g->dInvL, (size_t)Npad * GP_TILE, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(g->dAlpha, g->dAlpha, (size_t)Npad * g->P, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(dRec, dRec, 4, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclGroupEnd());
    HIPCHK(hipMemcpyAsync(rec, dRec, sizeof rec, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    g->jitter = rec[0];
    g->lml = rec[1];
    g->logdet = rec[2];
    g->fitted = true;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

// ---- Ky^-1 (potri-equivalent): dtrtri + dlauum re-expressed on the NT GEMM -------------------------
// W = L^-T is the candidate solve applied to the identity (row c of W = (L^-1 e_c)^T); rows above
// the current panel are still zero, so the tile sets are trapezoids and the cost is N^3/3.
// Ky^-1 = W W^T with the contraction of tile row a starting at column a*128: another N^3/3.
// Reference: pdinv / dpotri (GPy/GPy/util/linalg.py:127-145,193-214), Posterior.woodbury_inv
// (posterior.py:176-196).
// Ky^-1 from dT2 = L^-T (block upper triangular) into dWi
static int wi_lauum(gp_ctx *g) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    hipStream_t s = g->s;
    int ph;
    ph = phase_begin(g, "potri_lauum", (double)g->N * g->N * g->N / 3.0, 0.0);
    if (g->lauum_panels) {
        // Ky^-1 = (L^-T)(L^-T)^T accumulated k-panel by k-panel: panel p (W tiles of k) adds to the tiles (i, c), c <= i,
        // with i below the panel's end.  Every tile of a launch then walks the SAME k range, so the workgroups of an
        // XCD share their operand panels in L2 like the trailing updates do; as one launch over k = i*128 .. N each
        // tile streams its own up-to-33 MB row panels at its own offset and the product runs at the fabric's pace
        // (47 TFLOP/s at N = 32768).  The accumulator holds -Ky^-1 (C -= A B^T is the kernel's update form).
        HIPCHK(hipMemsetAsync(g->dWi, 0, sizeof(double) * Npad * Npad, s));
        const int W = g->panel_tiles;
        for (int k0 = 0; k0 < nt; k0 += W) {
            const int k1 = std::min(k0 + W, nt);
            GemmOpt o;
            o.k_tri = 1;
            o.k_sub = k0;
            gemm(g, s, 1, g->dWi, Npad, g->dT2 + (long)k0 * GP_TILE, Npad, g->dT2 + (long)k0 * GP_TILE, Npad, 1,
                 (k1 - k0) * GP_TILE, TileSet{0, k1, 0, k1, 1}, o);
        }
        launch_symmetrize_scale(s, g->dWi, Npad, Npad, -1.0);
    } else {
        GemmOpt o;
        o.k_tri = 1;
        gemm(g, s, 0, g->dWi, Npad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, nt, 0, nt, 1}, o);
        launch_symmetrize(s, g->dWi, Npad, Npad);
    }
    phase_end(g, ph);
    return 0;
}

// Ky^-1 in residue form ("emulate_fp64"): W = L^-T by the emulated solve of the identity (trapezoid; the residues of every
// solved panel stay in planes of N columns), then Ky^-1 = W W^T as residue launches over groups of k panels -- group
// [k0, k1) adds to the blocks (i, c), c <= i, whose rows lie above tile k1 -- into a zeroed accumulator, one CRT
// reconstruction of the lower blocks at the end and the same symmetrisation as the fp64 path.  Bound (rns.hip): the rows
// of W have norm sqrt((Ky^-1)_ii) <= 1 / sqrt(noise + 1e-8 + jitter) = 2^(eS-1) at most, so both contractions stay below
// 2^102 in integer units.  Reference: dtrtri + dpotri, GPy/GPy/util/linalg.py:127-145,193-214.
static int wi_rns(gp_ctx *g) {
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const int W = g->invp_W;
    hipStream_t s = g->s;
    int rc;
    RnsGeom r;
    if ((rc = rns_prepare(g, g->jitter, &r))) return rc;
    const int nt256 = r.nt256;
    const double lam = g->noise + 1e-8 + g->jitter;
    if (!(lam > 0.0)) return GP_ERR_RANGE;
    const int eS = std::max(0, 1 + (int)std::ceil(std::log2(1.0 / std::sqrt(lam))));
    const long wpitch = Npad, wrows = (long)nt256 * 256, wplane = wrows * wpitch;
    if ((rc = byte_realloc(&g->dWr, &g->capWr, (long)GP_RNS_T * wplane))) return rc;
    HIPCHK(hipMemsetAsync(g->dWr, 0, (size_t)GP_RNS_T * wplane, s));
    int ph = phase_begin(g, "potri_solve_emulated", (double)g->N * g->N * g->N / 3.0, 0.0);
    launch_set_identity(s, g->dT, Npad, Npad);
    RnsSolveOpt o;
    o.trapezoid = true;
    o.eS = eS;
    o.Wr = g->dWr;
    o.wpitch = wpitch;
    o.wplane = wplane;
    if ((rc = solve_rows_rns(g, g->dT, g->dT2, nt, o))) return rc;
    phase_end(g, ph);
    ph = phase_begin(g, "potri_lauum_emulated", (double)g->N * g->N * g->N / 3.0, 0.0);
    HIPCHK(hipMemsetAsync(g->dRr, 0, (size_t)GP_RNS_T * nt256 * nt256 * 65536, s));
    const long PB = (long)W * GP_TILE;
    const int G = std::max(1, std::min(g->rns_group, (int)(GP_RNS_KMAX / PB)));
    for (int k0 = 0; k0 < nt; k0 += G * W) {
        const int k1 = std::min(k0 + G * W, nt);
        const int rb = (k1 + 1) / 2;   // row blocks that hold a non-zero row of these columns of W
        rns_gemm(g, s, g->dWr + (long)k0 * GP_TILE, wpitch, wplane, g->dWr + (long)k0 * GP_TILE, wpitch, wplane, g->dRr, nt256, nt256,
                 rb, 0, rb, (k1 - k0) * GP_TILE, 0, 1);
    }
    HIPCHK(hipMemsetAsync(g->dWi, 0, sizeof(double) * Npad * Npad, s));
    launch_rns_reconstruct256(s, g->dRr, nt256, nt256, nt256, 0, nt, Npad, g->dWi, Npad, -std::ldexp(1.0, 2 * eS), 1);
    launch_symmetrize(s, g->dWi, Npad, Npad);
    phase_end(g, ph);
    return 0;
}

static int ensure_wi(gp_ctx *g) {
    if (g->wi_valid) return 0;
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    int rc;
    if ((rc = ensure_panel_inv(g))) return rc;
    if ((rc = dev_realloc(&g->dT, &g->capT, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dT2, &g->capT2, Npad * Npad))) return rc;
    if ((rc = dev_realloc(&g->dWi, &g->capWi, Npad * Npad))) return rc;
    double *T = g->dT;
    hipStream_t s = g->s;
    if (g->emulate_fp64 && g->emulate_fit && g->invp_W % 2 == 0) {
        rc = wi_rns(g);
        if (rc == 0) {
            g->wi_valid = true;
            g->predicted = false;
            return 0;
        }
        if (rc != GP_ERR_RANGE) return rc;
        ++g->emu_fallbacks;   // non-finite factor: the true-fp64 path below returns what the reference would
        g->nphases = 0;
    }
    int ph = phase_begin(g, "potri_solve", (double)g->N * g->N * g->N / 3.0, 0.0);
    launch_set_identity(s, T, Npad, Npad);
    solve_rows(g, T, g->dT2, nt, 1);  // dT2 = L^-T (block upper triangular)
    phase_end(g, ph);
    if ((rc = wi_lauum(g))) return rc;
    g->wi_valid = true;
    g->predicted = false;  // dT was reused
    return 0;
}

int gp_get_woodbury_inv(gp_t *g, double *Wi) {
    if (!g || !Wi) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_wi(g))) return rc;
    HIPCHK(hipStreamSynchronize(g->s));
    HIPCHK(hipMemcpy2D(Wi, sizeof(double) * g->N, g->dWi, sizeof(double) * g->Npad, sizeof(double) * g->N, g->N,
                       hipMemcpyDeviceToHost));
    return 0;
}

static int lml_grad_impl(gp_ctx *g, double *dvariance, double *dlengthscale, double *dnoise, bool reset_phases) {
    if (!g || !dvariance || !dlengthscale || !dnoise) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->P > 16) return fail(GP_ERR_ARG, "gp_lml_grad supports P <= 16");
    if (g->kp.gower) return fail(GP_ERR_STATE, "hyper-gradients of the Gower kernel are not replicated (the fork mixes "
                                               "Gower K with Euclidean dK/dr, stationary.py:218-238)");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if (reset_phases) g->nphases = 0;
    if ((rc = ensure_wi(g))) return rc;
    const long Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE), D = g->D;
    const long ntile = (long)nt * (nt + 1) / 2;
    // per-tile partials live in dT (free after ensure_wi): ntile * NACC doubles << Npad^2
    double *partial = g->dT;
    int ph = phase_begin(g, "lml_grad", 0.0, 8.0 * (double)g->N * g->N / 2);
    std::vector<double> host((size_t)GP_GRAD_NACC * ((D + GP_GRAD_CH - 1) / GP_GRAD_CH));
    int pass = 0;
    for (int d0 = 0; d0 < D; d0 += GP_GRAD_CH, ++pass) {
        launch_lml_grad(g->s, g->dX, g->N, Npad, g->kp, g->ard, d0, g->dAlpha, g->P, g->dWi, Npad, partial,
                        g->dScal + 64 + pass * GP_GRAD_NACC);
        if (!g->ard) break;
    }
    phase_end(g, ph);
    const int npass = g->ard ? (D + GP_GRAD_CH - 1) / GP_GRAD_CH : 1;
    HIPCHK(hipMemcpyAsync(host.data(), g->dScal + 64, sizeof(double) * GP_GRAD_NACC * npass, hipMemcpyDeviceToHost,
                          g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    (void)ntile;
    *dvariance = host[0] / g->kp.variance;  // stationary.py:224
    *dnoise = host[1];                      // gaussian.py:78-79
    if (g->ard) {
        for (int d = 0; d < D; ++d)         // -sum tmp (dx_q)^2 / l_q^3, stationary.py:230-235,260-261
            dlengthscale[d] = -host[(d / GP_GRAD_CH) * GP_GRAD_NACC + 2 + (d % GP_GRAD_CH)] / g->kp.ls[d];
    } else {
        dlengthscale[0] = -host[2] / g->kp.ls[0];  // -sum(dL_dr * r) / l, stationary.py:237-238
    }
    return 0;
}

int gp_lml_grad(gp_t *g, double *dvariance, double *dlengthscale, double *dnoise) {
    return lml_grad_impl(g, dvariance, dlengthscale, dnoise, true);
}

// gp_fit + gp_lml_grad as ONE call (what every L-BFGS evaluation of the hyper-parameter loop asks for:
// Model.objective_function + objective_function_gradients, core/model.py:96-127).  The first stages of the solve for
// L^-T ride behind the factorisation's latency-bound tail, like the candidate stages of gp_fit_predict.
int gp_fit_grad(gp_t *g, int maxtries, double *lml, double *logdet, double *jitter_used, double *dvariance,
                double *dlengthscale, double *dnoise) {
    if (!g || !dvariance || !dlengthscale || !dnoise) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params before gp_fit_grad");
    if (g->P > 16) return fail(GP_ERR_ARG, "gp_lml_grad supports P <= 16");
    if (g->kp.gower) return fail(GP_ERR_STATE, "hyper-gradients of the Gower kernel are not replicated (the fork mixes "
                                               "Gower K with Euclidean dK/dr, stationary.py:218-238)");
    HIPCHK(hipSetDevice(g->device));
    const int nt = (int)(g->Npad / GP_TILE);
    // emulated: Ky^-1 in residue form after the factorisation (wi_rns) instead of fp64 stages pipelined behind it
    const bool emu_wi = g->emulate_fp64 && g->emulate_fit && g->panel_tiles % 2 == 0;
    const bool can_pipe = g->lookahead && nt > g->panel_tiles && !emu_wi;
    int rc;
    if ((rc = fit_impl(g, maxtries, can_pipe ? 2 : 0, 0))) return rc;
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter_used) *jitter_used = g->jitter;
    return lml_grad_impl(g, dvariance, dlengthscale, dnoise, false);
}

// ---- second candidate-sized buffer (beta = K(Xs,X) Ky^-1, or the full covariance) -----------------
static int ensure_grad_buffers(gp_ctx *g, long elemsBeta, long M) {
    int rc;
    if ((rc = dev_realloc(&g->dCov, &g->capCov, elemsBeta))) return rc;
    const long need = M * (long)g->D * std::max(1, g->P);
    if (!g->dDm || g->capD < need) {
        for (double **b : {&g->dDm, &g->dDv, &g->dDacq}) {
            if (*b) hipFree(*b);
            *b = nullptr;
        }
        HIPCHK(hipMalloc((void **)&g->dDm, sizeof(double) * need));
        HIPCHK(hipMalloc((void **)&g->dDv, sizeof(double) * need));
        HIPCHK(hipMalloc((void **)&g->dDacq, sizeof(double) * need));
        g->capD = need;
    }
    return 0;
}

// predictive gradients of all resident candidates into dDm [M, D, P] and dDv [M, D]
static int run_predict_grad(gp_ctx *g) {
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->kp.gower) return fail(GP_ERR_STATE, "predictive gradients of the Gower kernel are not replicated");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    int rc;
    if ((rc = ensure_wi(g))) return rc;
    const long M = g->M, N = g->N, Npad = g->Npad;
    const int nt = (int)(Npad / GP_TILE);
    const long mc_max = std::min(g->mc_max, round_up(M, GP_TILE));
    if ((rc = dev_realloc(&g->dT, &g->capT, std::max(g->capT, mc_max * Npad)))) return rc;
    if ((rc = ensure_grad_buffers(g, mc_max * Npad, M))) return rc;
    for (long m0 = 0; m0 < M; m0 += mc_max) {
        const long mc = std::min(mc_max, M - m0);
        const long mcpad = round_up(mc, GP_TILE);
        const int mt = (int)(mcpad / GP_TILE);
        launch_cross_k(g->s, g->dT, Npad, g->dXs + m0 * g->D, mc, mcpad, g->dX, N, Npad, g->kp);
        // beta = K(Xs, X) Ky^-1   (gp.py:451-452; Ky^-1 symmetric => rows of Wi serve as the B operand)
        gemm(g, g->s, 0, g->dCov, Npad, g->dT, Npad, g->dWi, Npad, 1, (int)Npad, TileSet{0, mt, 0, nt, 0});
        launch_predict_grad(g->s, g->dXs + m0 * g->D, mc, g->dX, N, g->kp, g->dAlpha, Npad, g->P, g->dCov, Npad,
                            g->dDm + m0 * g->D * g->P, g->dDv + m0 * g->D);
    }
    g->predicted = false;  // dT no longer holds the solved candidates
    return 0;
}

int gp_predict_grad(gp_t *g, double *dmdx, double *dvdx) {
    if (!g || !dmdx || !dvdx) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_predict_grad(g))) return rc;
    HIPCHK(hipMemcpyAsync(dmdx, g->dDm, sizeof(double) * g->M * g->D * g->P, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(dvdx, g->dDv, sizeof(double) * g->M * g->D, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

int gp_acq_grad(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, double *out, double *dout) {
    if (!g || !out || !dout) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (g->P != 1) return fail(GP_ERR_ARG, "acquisitions need P == 1");
    if (type < GP_ACQ_EI || type > GP_ACQ_MPI) return fail(GP_ERR_ARG, "unknown acquisition %d", type);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict_grad(g))) return rc;
    if ((rc = run_predict(g, 1))) return rc;
    launch_acq_grad(g->s, type, par, fmin, y_mean, y_std, g->dMean, g->dVar, g->dDm, g->dDv, g->M, g->D, g->dAcq,
                    g->dDacq);
    HIPCHK(hipMemcpyAsync(out, g->dAcq, sizeof(double) * g->M, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(dout, g->dDacq, sizeof(double) * g->M * g->D, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    return 0;
}

// full_cov = True branch of PosteriorExact._raw_predict (posterior.py:280-284)
int gp_predict_full_cov(gp_t *g, int include_noise, double *mean, double *cov) {
    if (!g || !cov) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, Npad = g->Npad, Mpad = round_up(M, GP_TILE);
    if (Mpad > g->mc_max) return fail(GP_ERR_ARG, "full covariance needs M <= mc_max (%ld)", g->mc_max);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise))) return rc;  // leaves S = K(Xs,X) L^-T in dT2 (single chunk)
    if ((rc = dev_realloc(&g->dCov, &g->capCov, std::max(g->capCov, Mpad * Mpad)))) return rc;
    const int mt = (int)(Mpad / GP_TILE);
    launch_kbuild(g->s, g->dCov, Mpad, g->dXs, M, Mpad, g->kp, 0.0, 1);  // K(Xs, Xs)
    gemm(g, g->s, 1, g->dCov, Mpad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, mt, 0, mt, 0});
    if (include_noise) launch_add_diag(g->s, g->dCov, Mpad, M, g->noise);  // gaussian.py:104-105
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * M * g->P, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    HIPCHK(hipMemcpy2D(cov, sizeof(double) * M, g->dCov, sizeof(double) * Mpad, sizeof(double) * M, M,
                       hipMemcpyDeviceToHost));
    return 0;
}


/* ---- fit state (host scalars of the last fit, also what gp_comm_bcast_fit delivers to the receivers) ---- */
int gp_get_fit_state(gp_t *g, double *lml, double *logdet, double *jitter) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter) *jitter = g->jitter;
    return 0;
}

// ---- top-k of the acquisition scores (anchor_points_generator.py:61: argsort(scores)[:num_anchor]) ---------------
// k rounds of the deterministic arg-best reduction, each followed by masking the winner on the device: ties resolve
// to the lowest index in every round, i.e. the order of a stable sort by (score, index).
int gp_acq_topk(gp_t *g, int type, double par, double fmin, double y_mean, double y_std, int sense, int k,
                int64_t *idx, double *val) {
    if (!g || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (k < 1 || k > GP_TOPK_MAX) return fail(GP_ERR_ARG, "k out of range (1..%d)", GP_TOPK_MAX);
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = run_acq(g, type, par, fmin, y_mean, y_std))) return rc;
    if ((rc = dev_realloc(&g->dComm, &g->capComm, 2L * GP_TOPK_MAX * (1 + 128)))) return rc;
    double *dv = g->dComm;
    long long *di = (long long *)(g->dComm + GP_TOPK_MAX);
    const int kk = (int)std::min<long>(k, g->M);
    for (int j = 0; j < kk; ++j) {
        launch_argbest(g->s, g->dAcq, g->M, sense, g->dRedV + 256, g->dRedI + 256, g->dRedV, g->dRedI);
        HIPCHK(hipMemcpyAsync(dv + j, g->dRedV + 256, 8, hipMemcpyDeviceToDevice, g->s));
        HIPCHK(hipMemcpyAsync(di + j, g->dRedI + 256, 8, hipMemcpyDeviceToDevice, g->s));
        launch_mask(g->s, g->dAcq, g->dRedI + 256, 1, sense > 0 ? -INFINITY : INFINITY);
    }
    std::vector<long long> hi(kk);
    HIPCHK(hipMemcpyAsync(val, dv, sizeof(double) * kk, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpyAsync(hi.data(), di, sizeof(long long) * kk, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    for (int j = 0; j < kk; ++j) idx[j] = (int64_t)hi[j];
    for (int j = kk; j < k; ++j) {  // fewer candidates than k: the tail is marked empty
        idx[j] = -1;
        val[j] = sense > 0 ? -INFINITY : INFINITY;
    }
    return 0;
}

int gp_comm_allgather_topk(gp_t *g, int k, const double *vals, const int64_t *idxs, double *all_vals,
                           int64_t *all_idxs) {
    if (!g || !vals || !idxs || !all_vals || !all_idxs) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    if (k < 1 || k > GP_TOPK_MAX) return fail(GP_ERR_ARG, "k out of range (1..%d)", GP_TOPK_MAX);
    if (g->nranks > 128) return fail(GP_ERR_ARG, "nranks > 128");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = dev_realloc(&g->dComm, &g->capComm, 2L * GP_TOPK_MAX * (1 + 128)))) return rc;
    // k records of {double val, int64 idx} per rank, moved as 2k x 8 bytes
    std::vector<double> rec(2 * (size_t)k);
    for (int j = 0; j < k; ++j) {
        rec[2 * j] = vals[j];
        memcpy(&rec[2 * j + 1], &idxs[j], 8);
    }
    double *send = g->dComm, *recv = g->dComm + 2 * GP_TOPK_MAX;
    HIPCHK(hipMemcpyAsync(send, rec.data(), 16 * (size_t)k, hipMemcpyHostToDevice, g->s));
    NCCLCHK(ncclAllGather(send, recv, 2 * (size_t)k, ncclDouble, g->comm, g->s));
    std::vector<double> out(2 * (size_t)k * g->nranks);
    HIPCHK(hipMemcpyAsync(out.data(), recv, 16 * (size_t)k * g->nranks, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    for (size_t r = 0; r < (size_t)k * g->nranks; ++r) {
        all_vals[r] = out[2 * r];
        memcpy(&all_idxs[r], &out[2 * r + 1], 8);
    }
    return 0;
}

// ---- dL_dK = 0.5 (alpha alpha^T - P Ky^-1)  (exact_gaussian_inference.py:70) -------------------------------
// What grad_dict['dL_dK'] carries into kern.update_gradients_full (gp.py:269) when the reference's own kernel classes
// consume it on the host.  gp_lml_grad forms the same matrix implicitly inside its fused reduction.
int gp_get_dl_dk(gp_t *g, double *dL_dK) {
    if (!g || !dL_dK) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    int rc;
    if ((rc = ensure_wi(g))) return rc;  // leaves dT free (Npad x Npad)
    const long N = g->N, Npad = g->Npad;
    launch_dldk(g->s, g->dT, Npad, g->dAlpha, Npad, g->P, g->dWi, Npad, N);
    HIPCHK(hipStreamSynchronize(g->s));
    HIPCHK(hipMemcpy2D(dL_dK, sizeof(double) * N, g->dT, sizeof(double) * Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    g->predicted = false;  // dT was reused
    return 0;
}

// ---- posterior samples of the latent function (GP.posterior_samples_f, gp.py:581-609) ------------------------
// dev[s, :] = C z_s with C C^T = cov(Xs) (+ noise I) the full posterior covariance (posterior.py:280-284) of the resident
// candidates and z_s the caller's standard normals: the M x M Cholesky runs on the device with the same tile kernels as
// the fit, under GPy's jitter ladder (jitchol, linalg.py:56-81).  mean[M,P] is returned beside the deviations; a sample
// of output d is mean[:, d] + dev[s, :].  (The reference draws through numpy's multivariate_normal, whose SVD factor
// differs from C by an orthogonal matrix: same distribution, different draws for the same generator state.)
int gp_posterior_samples(gp_t *g, int include_noise, const double *Z, int S, int maxtries, double *mean, double *dev,
                         double *jitter_used) {
    if (!g || !Z || !dev) return fail(GP_ERR_ARG, "null argument");
    if (g->dead) return fail(GP_ERR_STATE, "the library was shut down (gp_shutdown): destroy this context and create a new one");
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (g->M < 1) return fail(GP_ERR_STATE, "gp_set_candidates first");
    if (S < 1) return fail(GP_ERR_ARG, "S < 1");
    HIPCHK(hipSetDevice(g->device));
    const long M = g->M, Npad = g->Npad, Mpad = round_up(M, GP_TILE), Spad = round_up(S, GP_TILE);
    if (Mpad > g->mc_max) return fail(GP_ERR_ARG, "posterior samples need M <= mc_max (%ld)", g->mc_max);
    int rc;
    if ((rc = ensure_out(g))) return rc;
    if ((rc = run_predict(g, include_noise))) return rc;  // S_c = K(Xs,X) L^-T in dT2 (single chunk), mean in dMean
    // dCov: [cov Mpad x Mpad][Z^T Spad x Mpad][dev Spad x Mpad]; the inverted diagonal tiles go to dT (free now)
    if ((rc = dev_realloc(&g->dCov, &g->capCov, std::max(g->capCov, Mpad * Mpad + 2 * Spad * Mpad)))) return rc;
    double *C = g->dCov, *Zd = g->dCov + Mpad * Mpad, *Dv = Zd + Spad * Mpad;
    double *invL = g->dT;
    const int mt = (int)(Mpad / GP_TILE), st = (int)(Spad / GP_TILE);
    HIPCHK(hipMemsetAsync(Zd, 0, sizeof(double) * Spad * Mpad, g->s));
    HIPCHK(hipMemcpy2DAsync(Zd, sizeof(double) * Mpad, Z, sizeof(double) * M, sizeof(double) * M, S,
                            hipMemcpyHostToDevice, g->s));
    const double diag0 = (g->kp.gower ? std::pow(g->kp.variance, g->D) : g->kp.variance) + (include_noise ? g->noise : 0.0);
    double jitter = 0.0;
    int tries = 0, info = 0;
    for (;;) {
        launch_kbuild(g->s, C, Mpad, g->dXs, M, Mpad, g->kp, 0.0, 1);  // K(Xs, Xs), identity on the padding rows
        gemm(g, g->s, 1, C, Mpad, g->dT2, Npad, g->dT2, Npad, 1, (int)Npad, TileSet{0, mt, 0, mt, 0});
        if (include_noise) launch_add_diag(g->s, C, Mpad, M, g->noise);
        if (jitter != 0.0) launch_add_diag(g->s, C, Mpad, M, jitter);
        HIPCHK(hipMemsetAsync(g->dInfo, 0, sizeof(int) * 4, g->s));
        factor_buf(g, C, Mpad, mt, mt, invL, g->dInfo);
        HIPCHK(hipMemcpyAsync(&info, g->dInfo, sizeof(int), hipMemcpyDeviceToHost, g->s));
        unsigned sync_words[2] = {0, 0};
        if (g->tail_tiles > 0) HIPCHK(hipMemcpyAsync(sync_words, g->dSync, sizeof sync_words, hipMemcpyDeviceToHost, g->s));
        HIPCHK(hipStreamSynchronize(g->s));
        if (sync_words[1] != 0)
            return fail(GP_ERR_HIP, "cooperative tail kernel: grid barrier timed out (a workgroup was not resident)");
        if (g->emulate_fp64 && info == 0) {
            int bad = 0;
            HIPCHK(hipMemcpy(&bad, g->dInfo + 2, sizeof(int), hipMemcpyDeviceToHost));
            if (bad) return fail(GP_ERR_STATE, "emulate_fp64: an entry of L left the fixed-point range");
        }
        if (info == 0) break;
        // jitchol: mean(diag) * 1e-6 * 10^k (linalg.py:62-75); the posterior covariance's diagonal is bounded by diag0
        if (!(diag0 > 0.0)) return fail(GP_ERR_NOT_PD_DIAG, "not pd: non-positive diagonal elements");
        jitter = tries == 0 ? diag0 * 1e-6 : jitter * 10.0;
        if (++tries > maxtries || !std::isfinite(jitter)) {
            g_err = "not positive definite, even with jitter.";
            return info > 0 ? info : 1;
        }
    }
    launch_zero_upper_diag(g->s, C, Mpad, mt);
    // dev[s, m] = sum_{k <= m} z[s, k] C[m, k]: B = the factor's rows, contraction ends at the diagonal tile
    GemmOpt o;
    o.k_end_tri = 1;
    gemm(g, g->s, 0, Dv, Mpad, Zd, Mpad, C, Mpad, 1, (int)Mpad, TileSet{0, st, 0, mt, 0}, o);
    if (mean) HIPCHK(hipMemcpyAsync(mean, g->dMean, sizeof(double) * M * g->P, hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipMemcpy2DAsync(dev, sizeof(double) * M, Dv, sizeof(double) * Mpad, sizeof(double) * M, S,
                            hipMemcpyDeviceToHost, g->s));
    HIPCHK(hipStreamSynchronize(g->s));
    if (jitter_used) *jitter_used = jitter;
    g->predicted = false;  // dT was used as workspace
    return 0;
}

}  // extern "C"


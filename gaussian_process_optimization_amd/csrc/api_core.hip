// Lifecycle of a context (gp_t), the per-device stream set, options, data / parameter upload, the accounted GEMM
// launcher and the small getters.  See include/gphip.h for the contract (reference file:line per entry point).
#include "api_internal.h"

thread_local std::string g_err;

// first failed launch / stream operation of the calling thread since the last report (gphip_internal.h)
static thread_local bool g_note_set = false;
static thread_local std::string g_note_msg;

// Every error return of the library goes through here.  A launch failure noted earlier in the SAME call (GP_LAUNCH / GP_NOTE)
// that this return would otherwise leave behind -- the entry point is leaving before its GP_SYNC -- is folded into the message
// and cleared: it must not surface as the failure of the thread's next, unrelated call.
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (g_note_set) {
        g_err += " [after: " + g_note_msg + "]";
        g_note_set = false;
    }
    return code;
}

void gp_note_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e == hipSuccess || g_note_set) return;
    char buf[512];
    snprintf(buf, sizeof buf, "%s -> %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_note_msg = buf;
    g_note_set = true;
}
int gp_pending_error() {
    if (!g_note_set) return 0;
    g_note_set = false;
    return fail(GP_ERR_HIP, "%s", g_note_msg.c_str());
}
// A note that survived its call after all (an entry point that returned success without synchronising) is dropped when the next
// entry point starts: GP_DEAD_CHECK, the first thing every gp_* function does with its context.
void gp_clear_stale_note() { g_note_set = false; }

static std::mutex g_ds_mu;
static std::map<int, DevStreams> g_ds;
static std::set<gp_ctx *> g_live;  // contexts created and not yet destroyed
static bool g_atexit_registered = false;

// Ordered shutdown (exported as gp_shutdown, and registered with atexit() at the first stream creation so that it runs
// BEFORE the HIP runtime's and a profiler's own exit handlers, which were registered earlier): quiesce every device the
// library touched, destroy the events recorded on the shared streams, then the streams.  Without it the five
// process-lifetime queues per device -- two of them created with hipExtStreamCreateWithCUMask -- were still alive when
// the runtime's static destructors ran; under rocprofv3 the runtime's queue teardown then called into the already
// finalised tool and the process died with SIGSEGV inside __cxa_finalize (round 1: every profiled run after the
// per-device stream set was introduced; plain runs exited 0).  See DESIGN.md, "Lifecycle".
void shutdown_all() {
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

int make_bulk_stream(int device, int reserve, hipStream_t *out) {
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

int get_streams(int device, int reserve, DevStreams *out) {
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

// ---- phase timing -----------------------------------------------------------------------------
int phase_begin(gp_ctx *g, const char *name, double flops, double bytes) {
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
    GP_NOTE(hipEventRecord(p.e0, g->s));
    return g->nphases++;
}
void phase_end(gp_ctx *g, int id) {
    if (id >= 0) GP_NOTE(hipEventRecord(g->phases[id].e1, g->s));
}

// ---- GEMM wrapper with accounting ---------------------------------------------------------------
void gemm(gp_ctx *g, hipStream_t s, int mode, double *C, long ldc, const double *A, long lda, const double *B, long ldb, int b_mul, int K, TileSet ts, const GemmOpt &o) {
    const long n = tileset_count(ts) * o.batch;
    if (n <= 0 || K <= 0) return;
    GemmOpt oo = o;
    if (n >= 1024 && !oo.stagger) oo.stagger = g->stagger;
    if (n >= 1024 && g->waves8) oo.waves8 = 1;  // 4 waves/SIMD: +2 % on the long launches (measured)
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
    // gp_profile(2) brackets the OTHER fp64 symbol instead, gemm_nt_kernel<1, 64, 2, false, 64> (C -= A B^T as 64 x 64 work
    // units: the chain's in-panel / look-ahead updates and the short candidate updates) -- in a pass of its own, for the reason above
    const bool small_sym = mode == 1 && oo.small && !oo.pair && !oo.rows64;
    const bool timed = g->profiling && (g->profile_class == 1 ? small_sym :
                       (n >= g->profile_min_tiles &&
                        (g->profile_min_tiles < 1024 || (mode == 1 && oo.waves8 && !oo.small))));  // tracing tools lower the threshold
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
        GP_NOTE(hipEventRecord(e0, s));
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
        GP_NOTE(hipEventRecord(e1, s));
        g->gemm_launches++;
        g->gemm_flops += 2.0 * GP_TILE * GP_TILE * (double)K * (double)n * ((o.k_tri || o.k_end_tri) ? 0.5 : 1.0);
    }
    g->gemm_flops_all += 2.0 * GP_TILE * GP_TILE * (double)K * (double)n * ((o.k_tri || o.k_end_tri) ? 0.5 : 1.0);
}

// ---- memory helpers -----------------------------------------------------------------------------
int dev_realloc(double **p, long *cap, long need) {
    if (need <= *cap && *p) return 0;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)p, (size_t)need * sizeof(double));
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hipMalloc(%ld doubles) -> %s", need, hipGetErrorString(e));
    *cap = need;
    return 0;
}

extern "C" const char *gp_last_error(void) { return g_err.c_str(); }
extern "C" const char *gp_version(void) { return "gphip 0.1 (gfx950, fp64 MFMA)"; }

extern "C" int gp_device_count(int *count) {
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

extern "C" int gp_device_info(int device, char *name, int cap, int *cus, int64_t *hbm_bytes) {
    hipDeviceProp_t pr;
    HIPCHK(hipGetDeviceProperties(&pr, device));
    if (name && cap > 0) {
        snprintf(name, cap, "%s (%s)", pr.name, pr.gcnArchName);
    }
    if (cus) *cus = pr.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)pr.totalGlobalMem;
    return 0;
}

extern "C" int gp_create(gp_t **out, int device) {
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
    hipError_t e = hipMalloc((void **)&g->dInfo, sizeof(int) * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dScal, sizeof(double) * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dRedV, sizeof(double) * 512);
    if (e == hipSuccess) e = hipMalloc((void **)&g->dRedI, sizeof(long long) * 1024);
    if (e != hipSuccess) {
        if (g->dInfo) hipFree(g->dInfo);
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
    if (const char *ev = getenv("GPHIP_INNER_TILES")) g->inner_tiles = atoi(ev) == 2 ? 2 : 1;
    if (const char *ev = getenv("GPHIP_INNER_MIN_ROWS")) g->inner_min_rows = std::max(0, atoi(ev));   // A/B of the in-panel step across unmodified tools
    if (const char *ev = getenv("GPHIP_OWN_KEEP_PER_ROW")) g->own_keep_per_row = std::max(0, atoi(ev));
    if (const char *ev = getenv("GPHIP_OWN_KEEP_BASE")) g->own_keep_base = std::max(0, atoi(ev));   // (a wide owned range forced on an unmodified test suite)
    *out = g;
    return 0;
}

extern "C" int gp_shutdown(void) {
    shutdown_all();
    return 0;
}

void destroy_ctx_events(gp_ctx *g) {
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
        for (hipEvent_t e : v)
            if (e) hipEventDestroy(e);
        v.clear();
    }
}

extern "C" int gp_destroy(gp_t *g) {
    if (!g) return 0;
    {
        std::lock_guard<std::mutex> lk(g_ds_mu);
        if (!g_live.erase(g)) return 0;  // not a live context (double destroy)
    }
    hipSetDevice(g->device);
    hipDeviceSynchronize();
    if (g->comm) ncclCommDestroy(g->comm);
    double *ptrs[] = {g->dX, g->dY, g->dA, g->dInvL, g->dAlpha, g->dW, g->dMu, g->dScal, g->dRedV,
                      g->dXs, g->dT, g->dMean, g->dVar, g->dAcq, g->dWi, g->dT2, g->dDm, g->dDv, g->dDacq, g->dCov, g->dInvP, g->dInvPw, g->dLp, g->dComm,
                      g->dX2, g->dK2, g->dLi, g->dRows};
    for (double *p : ptrs)
        if (p) hipFree(p);
    if (g->dRowsCounter) hipFree(g->dRowsCounter);
    if (g->hRowsOut) hipHostFree(g->hRowsOut);
    if (g->dInfo) hipFree(g->dInfo);
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

extern "C" int gp_set_option(gp_t *g, const char *name, int64_t value) {
    if (!g || !name) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!strcmp(name, "panel_tiles")) {
        if (value < 1 || value > 64) return fail(GP_ERR_ARG, "panel_tiles out of range");
        g->panel_tiles = (int)value;
    } else if (!strcmp(name, "inner_tiles")) {
        if (value != 1 && value != 2) return fail(GP_ERR_ARG, "inner_tiles must be 1 or 2");
        g->inner_tiles = (int)value;
    } else if (!strcmp(name, "inner_min_rows")) {
        if (value < 0) return fail(GP_ERR_ARG, "inner_min_rows < 0");
        g->inner_min_rows = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "own_keep_per_row")) {
        if (value < 0) return fail(GP_ERR_ARG, "own_keep_per_row < 0");
        g->own_keep_per_row = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "own_keep_pipe_pct")) {
        if (value < 0 || value > 400) return fail(GP_ERR_ARG, "own_keep_pipe_pct out of range");
        g->own_keep_pipe_pct = (int)value;
    } else if (!strcmp(name, "own_keep_base")) {
        if (value < 0) return fail(GP_ERR_ARG, "own_keep_base < 0");
        g->own_keep_base = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "lookahead")) {
        g->lookahead = value ? 1 : 0;
    } else if (!strcmp(name, "lookahead_min_tiles")) {
        if (value < 0) return fail(GP_ERR_ARG, "lookahead_min_tiles < 0");
        g->lookahead_min_tiles = (int)std::min<int64_t>(value, 1 << 20);
    } else if (!strcmp(name, "pipe_start_pct")) {
        if (value < -1 || value > 100) return fail(GP_ERR_ARG, "pipe_start_pct out of range");
        g->pipe_start_pct = (int)value;   // -1: automatic
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
        g->li_valid = false;
        g->rows_calls_since_fit = 0;
        g->w_in_t2 = false;
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
        if (g->supertile != (int)value) {   // the cached tile orders belong to the old edge
            HIPCHK(hipSetDevice(g->device));
            HIPCHK(hipDeviceSynchronize());
            for (auto &kv : g->tile_lists) hipFree(kv.second);
            g->tile_lists.clear();
        }
        g->supertile = (int)value;
    } else if (!strcmp(name, "reserve_cus")) {
        if (value != g->bulk_reserved)
            return fail(GP_ERR_ARG, "reserve_cus is fixed when the device's streams are created (%d); set GPHIP_RESERVE_CUS "
                                    "before the first gp_create", g->bulk_reserved);
    } else if (!strcmp(name, "trsm_rows64")) {
        g->trsm_rows64 = (int)value;
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
    } else if (!strcmp(name, "debug_potrf_lds")) {
        if (value < 0 || value > (1 << 20)) return fail(GP_ERR_ARG, "debug_potrf_lds out of range");
        potrf_set_debug_lds((int)value);   // process-wide test hook: forces refused diagonal-tile launches (tests/test_gpu_round4.py)
    } else if (!strcmp(name, "debug_rows_skew")) {
        // test hook: shifts the arrival base of the one-location path's next pass, so that no workgroup finishes it
        g->rows_counter_base += (unsigned int)value;
    } else if (!strcmp(name, "rows_build")) {
        if (value < -1 || value > 1) return fail(GP_ERR_ARG, "rows_build out of range (-1: by rule, 0: never, 1: at the first call)");
        g->rows_build = (int)value;
    } else if (!strcmp(name, "rows_nt")) {
        if (value < -1 || value > 1) return fail(GP_ERR_ARG, "rows_nt out of range (-1: automatic, 0, 1)");
        g->rows_nt = (int)value;
    } else if (!strcmp(name, "small_m")) {
        if (value < 0 || value > 8) return fail(GP_ERR_ARG, "small_m out of range (0..8)");
        g->small_m = value;
        g->predicted = false;
    } else if (!strcmp(name, "mc_max")) {
        if (value < GP_TILE) return fail(GP_ERR_ARG, "mc_max < 128");
        g->mc_max = round_up(value, GP_TILE);
    } else
        return fail(GP_ERR_ARG, "unknown option %s", name);
    return 0;
}

extern "C" int gp_synchronize(gp_t *g) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    HIPCHK(hipSetDevice(g->device));
    GP_SYNC(g->s_panel);
    if (g->s_bulk) GP_SYNC(g->s_bulk);
    if (g->s_inv) GP_SYNC(g->s_inv);
    if (g->s_pred) GP_SYNC(g->s_pred);
    GP_SYNC(g->s);
    return 0;
}

extern "C" int gp_set_data(gp_t *g, const double *X, const double *Y, int64_t N, int D, int P) {
    if (!g || !X || !Y) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (N < 1 || D < 1 || D > GP_MAX_D || P < 1 || P > GP_MAX_RHS)
        return fail(GP_ERR_ARG, "bad shape N=%ld D=%d P=%d (D <= %d, P <= %d)", (long)N, D, P, GP_MAX_D, GP_MAX_RHS);
    HIPCHK(hipSetDevice(g->device));
    GP_SYNC(g->s);
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
        // the diagonal-tile kernel writes the lower block triangle of each inverted tile only (potrf.hip): the blocks above
        // it are zero from here on
        HIPCHK(hipMemsetAsync(g->dInvL, 0, sizeof(double) * capN * GP_TILE, g->s));
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
    GP_SYNC(g->s);
    g->have_data = true;
    g->fitted = false;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    g->kp.D = D;
    return 0;
}

// The fork's Gower kernel option (GPy/GPy/kern/src/stationary.py:61-65,116-135; lengthscales = variable ranges,
// GPyOpt/GPyOpt/core/task/space.py:351-362).  Only K is Gower: Kdiag stays `variance` and the gradient
// formulas stay Euclidean in the fork.  Both are reproduced for the predictive gradients (gp_predict_grad, gp_acq_grad,
// gp_acq_lp_grad: Gower K(Xs, X) inside Euclidean gradients_X, see run_predict_grad) and for the hyper-parameter gradients
// (gp_lml_grad: Gower K in the variance gradient, Euclidean dK/dr in the lengthscale gradients, stationary.py:218-238).
extern "C" int gp_set_gower(gp_t *g, int enable, const int *is_discrete, const double *range) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
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
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

extern "C" int gp_set_params(gp_t *g, int kernel, int ard, double variance, const double *lengthscale, double noise) {
    if (!g || !lengthscale) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
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
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

int byte_realloc(signed char **p, long *cap, long need) {
    if (need <= *cap && *p) return 0;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void **)p, (size_t)need);
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hipMalloc(%ld bytes) -> %s", need, hipGetErrorString(e));
    *cap = need;
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
int ensure_bulk_stream(gp_ctx *g) {
    DevStreams d;
    int rc = get_streams(g->device, g->reserve_cus, &d);
    if (rc) return rc;
    g->s_bulk = d.bulk;
    g->bulk_reserved = d.reserved;
    return 0;
}

hipEvent_t la_event(gp_ctx *g, int kind, size_t i) {
    std::vector<hipEvent_t> &v = g->la_events[kind];
    while (v.size() <= i) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            // recorded, not swallowed: the entry point that ordered its streams through this event fails (la_events_ok)
            g->ev_error = true;
            e = nullptr;
        }
        v.push_back(e);
    }
    return v[i];
}
// Every stream-ordering event the last multi-stream section asked for exists (a failed hipEventCreateWithFlags would
// otherwise silently drop a dependency between two streams).
int la_events_ok(gp_ctx *g) {
    if (!g->ev_error) return 0;
    g->ev_error = false;
    for (auto &v : g->la_events) {   // drop the holes so that a later call retries the creation
        while (!v.empty() && v.back() == nullptr) v.pop_back();
        for (hipEvent_t e : v)
            if (!e) return fail(GP_ERR_HIP, "hipEventCreateWithFlags failed (stream-ordering event of the factorisation)");
    }
    return fail(GP_ERR_HIP, "hipEventCreateWithFlags failed (stream-ordering event of the factorisation)");
}

extern "C" int gp_get_alpha(gp_t *g, double *alpha) {
    if (!g || !alpha) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    std::vector<double> tmp((size_t)g->P * g->Npad);
    HIPCHK(hipMemcpy(tmp.data(), g->dAlpha, sizeof(double) * g->P * g->Npad, hipMemcpyDeviceToHost));
    for (long i = 0; i < g->N; ++i)
        for (int p = 0; p < g->P; ++p) alpha[i * g->P + p] = tmp[(size_t)p * g->Npad + i];
    return 0;
}

extern "C" int gp_get_chol(gp_t *g, double *L) {
    if (!g || !L) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    HIPCHK(hipSetDevice(g->device));
    const long N = g->N;
    HIPCHK(hipMemcpy2D(L, sizeof(double) * N, g->dA, sizeof(double) * g->Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    for (long i = 0; i < N; ++i)
        for (long j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    return 0;
}

extern "C" int gp_kernel_matrix(gp_t *g, double *K) {
    if (!g || !K) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params first");
    HIPCHK(hipSetDevice(g->device));
    const long N = g->N;
    launch_kbuild(g->s, g->dA, g->Npad, g->dX, N, g->Npad, g->kp, 0.0, 1);
    GP_SYNC(g->s);
    HIPCHK(hipMemcpy2D(K, sizeof(double) * N, g->dA, sizeof(double) * g->Npad, sizeof(double) * N, N,
                       hipMemcpyDeviceToHost));
    g->fitted = false;  // dA was overwritten
    g->wi_valid = false;
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
    return 0;
}

// kern.K(X, X2), stationary.py:107-140 with X2 given: the training inputs play the candidate role of cross_k_kernel
// (rows of the output), X2 the training role (columns), so K[N, M2] comes out row-major as the reference returns it.
extern "C" int gp_cross_kernel_matrix(gp_t *g, const double *X2, int64_t M2, double *K) {
    if (!g || !X2 || !K) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "set data and params first");
    if (M2 < 1) return fail(GP_ERR_ARG, "M2 < 1");
    HIPCHK(hipSetDevice(g->device));
    const long N = g->N, Npad = g->Npad, M2pad = round_up(M2, GP_TILE);
    int rc;
    if ((rc = dev_realloc(&g->dX2, &g->capX2, (long)M2 * g->D))) return rc;
    if ((rc = dev_realloc(&g->dK2, &g->capK2, Npad * M2pad))) return rc;
    HIPCHK(hipMemcpyAsync(g->dX2, X2, sizeof(double) * M2 * g->D, hipMemcpyHostToDevice, g->s));
    launch_cross_k(g->s, g->dK2, M2pad, g->dX, N, Npad, g->dX2, M2, M2pad, g->kp);
    GP_SYNC(g->s);
    HIPCHK(hipMemcpy2D(K, sizeof(double) * M2, g->dK2, sizeof(double) * M2pad, sizeof(double) * M2, N, hipMemcpyDeviceToHost));
    return 0;
}

/* ---- fit state (host scalars of the last fit, also what gp_comm_bcast_fit delivers to the receivers) ---- */
extern "C" int gp_get_fit_state(gp_t *g, double *lml, double *logdet, double *jitter) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->fitted) return fail(GP_ERR_STATE, "gp_fit first");
    if (lml) *lml = g->lml;
    if (logdet) *logdet = g->logdet;
    if (jitter) *jitter = g->jitter;
    return 0;
}

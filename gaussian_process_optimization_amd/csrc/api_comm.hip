// The one collective of the path: RCCL all-gather of per-rank (best value, global row) pairs; broadcast of a fit.
// Reference call patterns: run.py:1240-1241, GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59-61.
#include "api_internal.h"

// ---- multi-GPU --------------------------------------------------------------------------------------
extern "C" int gp_comm_unique_id(char *uid128) {
    if (!uid128) return fail(GP_ERR_ARG, "null uid");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(uid128, &id, 128);
    return 0;
}

// the version the librccl this process resolved reports about itself (the bench line carries it beside the mapped path)
extern "C" int gp_comm_version(int *version) {
    if (!version) return fail(GP_ERR_ARG, "null argument");
    NCCLCHK(ncclGetVersion(version));
    return 0;
}

extern "C" int gp_comm_init(gp_t *g, const char *uid128, int rank, int nranks) {
    if (!g || !uid128) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
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
extern "C" int gp_comm_info(gp_t *g, int *rank, int *nranks) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    int r = -1, n = -1;
    NCCLCHK(ncclCommUserRank(g->comm, &r));
    NCCLCHK(ncclCommCount(g->comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return 0;
}

extern "C" int gp_comm_destroy(gp_t *g) {
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

extern "C" int gp_comm_allgather_best(gp_t *g, double val, int64_t idx, double *vals, int64_t *idxs) {
    if (!g || !vals || !idxs) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
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
    GP_SYNC(g->s);
    for (int r = 0; r < g->nranks; ++r) {
        vals[r] = out[2 * r];
        memcpy(&idxs[r], &out[2 * r + 1], 8);
    }
    return 0;
}

// The fit's host scalars ride along with the factor as one small record; a receiving rank takes them over and drops
// everything derived from its previous factor.
#define GP_FIT_RECORD_LEN 4
static void pack_fit_record(const gp_ctx *g, double *rec) {
    rec[0] = g->jitter;
    rec[1] = g->lml;
    rec[2] = g->logdet;
    rec[3] = 0.0;
}
static void apply_fit_record(gp_ctx *g, const double *rec) {
    g->jitter = rec[0];   // a receiver's gp_fmin is y - (noise + 1e-8 + jitter) alpha with the ROOT's jitter
    g->lml = rec[1];
    g->logdet = rec[2];
    g->fitted = true;
    g->fmin_valid = false;
    g->wi_valid = false;
    g->li_valid = false;
    g->rows_calls_since_fit = 0;
    g->w_in_t2 = false;
    g->invp_valid = false;
    g->lr_valid = false;
    g->predicted = false;
}

extern "C" int gp_comm_bcast_fit(gp_t *g, int root) {
    if (!g) return fail(GP_ERR_ARG, "null gp");
    GP_DEAD_CHECK(g);
    if (!g->comm) return fail(GP_ERR_STATE, "gp_comm_init first");
    if (!g->have_data || !g->have_params) return fail(GP_ERR_STATE, "every rank needs data and params set");
    HIPCHK(hipSetDevice(g->device));
    if (root < 0 || root >= g->nranks) return fail(GP_ERR_ARG, "root %d out of range", root);
    if (g->rank == root && !g->fitted) return fail(GP_ERR_STATE, "the root rank must be fitted");
    const long Npad = g->Npad;
    // host-side fit state rides along as a small record: a receiver's gp_fmin uses the root's jitter
    // (y - (noise + 1e-8 + jitter) alpha) and reports the root's LML / log det
    double rec[GP_FIT_RECORD_LEN];
    pack_fit_record(g, rec);
    double *dRec = g->dScal + 400;
    if (g->rank == root) HIPCHK(hipMemcpyAsync(dRec, rec, sizeof rec, hipMemcpyHostToDevice, g->s));
    NCCLCHK(ncclGroupStart());
    NCCLCHK(ncclBroadcast(g->dA, g->dA, (size_t)(Npad + GP_MAX_RHS) * Npad, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(g->dInvL, g->dInvL, (size_t)Npad * GP_TILE, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(g->dAlpha, g->dAlpha, (size_t)Npad * g->P, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclBroadcast(dRec, dRec, GP_FIT_RECORD_LEN, ncclDouble, root, g->comm, g->s));
    NCCLCHK(ncclGroupEnd());
    HIPCHK(hipMemcpyAsync(rec, dRec, sizeof rec, hipMemcpyDeviceToHost, g->s));
    GP_SYNC(g->s);
    apply_fit_record(g, rec);
    return 0;
}

// Host-only check of the two functions above (no device, no communicator: RCCL with two ranks cannot run on a one-GPU
// lease, and the receiver-side assignment must not depend on having been the root).  A scratch context is put into the
// state of a receiving rank that holds stale results of an earlier fit -- everything valid, other scalars -- the root's
// record is packed from `root_state` = {jitter, lml, logdet} and applied; state_out = the receiver's {jitter, lml, logdet},
// flags_out = {fitted, fmin_valid, wi_valid, invp_valid, lr_valid, predicted}.
extern "C" int gp_comm_selftest_fit_record(const double *root_state, double *state_out, int *flags_out) {
    if (!root_state || !state_out || !flags_out) return fail(GP_ERR_ARG, "null argument");
    gp_ctx root, recv;
    root.jitter = root_state[0];
    root.lml = root_state[1];
    root.logdet = root_state[2];
    recv.jitter = -1.0;
    recv.lml = recv.logdet = 12345.0;
    recv.fitted = false;
    recv.fmin_valid = recv.wi_valid = recv.invp_valid = recv.lr_valid = recv.predicted = true;
    double rec[GP_FIT_RECORD_LEN];
    pack_fit_record(&root, rec);
    apply_fit_record(&recv, rec);
    state_out[0] = recv.jitter;
    state_out[1] = recv.lml;
    state_out[2] = recv.logdet;
    const bool f[6] = {recv.fitted, recv.fmin_valid, recv.wi_valid, recv.invp_valid, recv.lr_valid, recv.predicted};
    for (int i = 0; i < 6; ++i) flags_out[i] = f[i] ? 1 : 0;
    return 0;
}

extern "C" int gp_comm_allgather_topk(gp_t *g, int k, const double *vals, const int64_t *idxs, double *all_vals,
                           int64_t *all_idxs) {
    if (!g || !vals || !idxs || !all_vals || !all_idxs) return fail(GP_ERR_ARG, "null argument");
    GP_DEAD_CHECK(g);
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
    GP_SYNC(g->s);
    for (size_t r = 0; r < (size_t)k * g->nranks; ++r) {
        all_vals[r] = out[2 * r];
        memcpy(&all_idxs[r], &out[2 * r + 1], 8);
    }
    return 0;
}

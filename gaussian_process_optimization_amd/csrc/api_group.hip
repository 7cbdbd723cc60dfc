// gp_group_*: the candidate table of ONE caller thread scored on several devices of the node.
//
// The reference's Bayesian-optimisation loop is one Python process (GPyOpt/GPyOpt/core/bo.py:73-168, run.py:1207-1258): it
// calls acquisition_function(table) and takes argmax / argsort()[:5] (run.py:1240-1241,
// GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59-61).  A group gives that caller every GPU without changing its
// process model: one context (gp_t) per entry of devices[], the model replicated on each (the fit does not shard, SURVEY.md
// 8e: every member factors its own replica, concurrently), the table cut into contiguous row blocks, one per member, and
// the per-block winners exchanged and merged with NumPy's lowest-index tie rule.  Inside, one worker thread per DISTINCT
// device drives that device's members through the same entry points a rank of the one-process-per-GPU layout calls
// (gp_fit, gp_acq_argbest / gp_acq_topk), so both layouts run the same scoring code.
//   * devices all different: the members get the communicators of ncclCommInitAll and the winners travel by ONE grouped RCCL
//     all-gather over xGMI, enqueued for every member by the calling thread (group_allgather below); every member ends up
//     with all pairs (member 0's copy is merged).  NOT yet run on more than one device: no multi-GPU box has been available to
//     any round of this build (include/gphip.h says so too); the host-merge form is what the tests exercise;
//   * a device listed twice (a one-GPU box rehearsing the N > 1 logic): no communicator can hold one device twice, the pairs
//     are merged on the host instead -- said by gp_group_info.
#include "api_internal.h"

#include <thread>

struct gp_group {
    std::vector<gp_ctx *> m;             // members, in the order of devices[]
    std::vector<int> dev;
    std::vector<std::vector<int>> lanes; // member indices per distinct device: one worker thread each
    bool rccl = false;                   // members hold ncclCommInitAll communicators
    std::string rccl_note;               // why not, when not
    long M = 0;                          // rows of the resident table
    std::vector<long> lo, hi;            // member i scores rows [lo[i], hi[i])
};

// first M % n members take one extra row (sharded.shard_bounds)
static void block_of(long M, int i, int n, long *lo, long *hi) {
    const long base = M / n, rem = M % n;
    *lo = i * base + std::min<long>(i, rem);
    *hi = *lo + base + (i < rem ? 1 : 0);
}

// fn(member index) on every member: the members of one device in order on one thread, the devices side by side.  The first
// failure (lowest member index) is the group's; its message is carried over from the worker's thread-local slot.
template <class F>
static int for_members(gp_group *grp, F fn) {
    const size_t n = grp->m.size();
    std::vector<int> rcs(n, 0);
    std::vector<std::string> msgs(n);
    auto lane = [&](const std::vector<int> &ids) {
        for (int i : ids) {
            rcs[i] = fn(i);
            if (rcs[i]) {
                msgs[i] = gp_last_error();
                break;   // later members of this device are not started on a failed device
            }
        }
    };
    if (grp->lanes.size() == 1) {
        lane(grp->lanes[0]);
    } else {
        std::vector<std::thread> th;
        for (const auto &ids : grp->lanes) th.emplace_back(lane, std::cref(ids));
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < n; ++i)
        if (rcs[i]) {
            if (rcs[i] > 0) {   // a leading minor of the replica is not positive definite: the reference's message, the same code
                g_err = msgs[i];
                return rcs[i];
            }
            return fail(rcs[i], "member %zu (device %d): %s", i, grp->dev[i], msgs[i].c_str());
        }
    return 0;
}

extern "C" int gp_group_create(gp_group_t **out, int ndev, const int *devices) {
    if (!out || !devices) return fail(GP_ERR_ARG, "null argument");
    if (ndev < 1 || ndev > 128) return fail(GP_ERR_ARG, "ndev out of range (1..128)");
    gp_group *grp = new gp_group();
    bool distinct = true;
    for (int i = 0; i < ndev; ++i) {
        gp_t *g = nullptr;
        int rc = gp_create(&g, devices[i]);
        if (rc) {
            const std::string msg = gp_last_error();
            for (gp_ctx *p : grp->m) gp_destroy(p);
            delete grp;
            return fail(rc, "gp_group_create: member %d (device %d): %s", i, devices[i], msg.c_str());
        }
        grp->m.push_back(g);
        grp->dev.push_back(devices[i]);
        size_t l = 0;
        for (; l < grp->lanes.size(); ++l)
            if (grp->dev[grp->lanes[l][0]] == devices[i]) break;
        if (l == grp->lanes.size()) grp->lanes.emplace_back();
        else distinct = false;
        grp->lanes[l].push_back(i);
    }
    if (distinct) {
        std::vector<ncclComm_t> comms(ndev, nullptr);
        ncclResult_t r = ncclCommInitAll(comms.data(), ndev, devices);
        if (r == ncclSuccess) {
            for (int i = 0; i < ndev; ++i) {
                grp->m[i]->comm = comms[i];
                grp->m[i]->rank = i;
                grp->m[i]->nranks = ndev;
            }
            grp->rccl = true;
        } else {
            grp->rccl_note = std::string("ncclCommInitAll -> ") + ncclGetErrorString(r) + ": host merge";
        }
    } else {
        grp->rccl_note = "a device is listed more than once: host merge";
    }
    *out = grp;
    return 0;
}

extern "C" int gp_group_destroy(gp_group_t *grp) {
    if (!grp) return 0;
    for (gp_ctx *g : grp->m) gp_destroy(g);   // (destroys the member's communicator too)
    delete grp;
    return 0;
}

extern "C" int gp_group_info(gp_group_t *grp, int *ndev, int *uses_rccl, char *note, int cap) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    if (ndev) *ndev = (int)grp->m.size();
    if (uses_rccl) *uses_rccl = grp->rccl ? 1 : 0;
    if (note && cap > 0) snprintf(note, cap, "%s", grp->rccl ? "RCCL all-gather (ncclCommInitAll)" : grp->rccl_note.c_str());
    return 0;
}

extern "C" int gp_group_member(gp_group_t *grp, int i, gp_t **member) {
    if (!grp || !member) return fail(GP_ERR_ARG, "null argument");
    if (i < 0 || i >= (int)grp->m.size()) return fail(GP_ERR_ARG, "member %d out of range", i);
    *member = grp->m[i];
    return 0;
}

extern "C" int gp_group_set_option(gp_group_t *grp, const char *name, int64_t value) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    for (gp_ctx *g : grp->m) {
        int rc = gp_set_option(g, name, value);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gp_group_set_data(gp_group_t *grp, const double *X, const double *Y, int64_t N, int D, int P) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    return for_members(grp, [&](int i) { return gp_set_data(grp->m[i], X, Y, N, D, P); });
}

extern "C" int gp_group_set_params(gp_group_t *grp, int kernel, int ard, double variance, const double *lengthscale, double noise) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    for (gp_ctx *g : grp->m) {
        int rc = gp_set_params(g, kernel, ard, variance, lengthscale, noise);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int gp_group_set_gower(gp_group_t *grp, int enable, const int *is_discrete, const double *range) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    for (gp_ctx *g : grp->m) {
        int rc = gp_set_gower(g, enable, is_discrete, range);
        if (rc) return rc;
    }
    return 0;
}

// Every member factors its own replica, the devices side by side ("replicas only": no traffic, the wall time of one fit).
// The scalars returned are member 0's; replicas of one model on one kind of device run the same instructions, and a member
// whose LML differs from member 0's in any bit fails the call (a replica that diverged would score its block against
// another posterior).
extern "C" int gp_group_fit(gp_group_t *grp, int maxtries, double *lml, double *logdet, double *jitter_used) {
    if (!grp) return fail(GP_ERR_ARG, "null group");
    const size_t n = grp->m.size();
    std::vector<double> l(n), d(n), j(n);
    int rc = for_members(grp, [&](int i) { return gp_fit(grp->m[i], maxtries, &l[i], &d[i], &j[i]); });
    if (rc) return rc;
    for (size_t i = 1; i < n; ++i)
        if (memcmp(&l[i], &l[0], sizeof(double)) || memcmp(&j[i], &j[0], sizeof(double)))
            return fail(GP_ERR_STATE, "replica %zu (device %d) disagrees with replica 0: LML %.17g vs %.17g, jitter %g vs %g", i,
                        grp->dev[i], l[i], l[0], j[i], j[0]);
    if (lml) *lml = l[0];
    if (logdet) *logdet = d[0];
    if (jitter_used) *jitter_used = j[0];
    return 0;
}

extern "C" int gp_group_fmin(gp_group_t *grp, double *fmin) {
    if (!grp || !fmin) return fail(GP_ERR_ARG, "null argument");
    return gp_fmin(grp->m[0], fmin);
}

// The WHOLE table; member i keeps rows [lo_i, hi_i) resident (contiguous blocks, the first M % ndev one row longer).
extern "C" int gp_group_set_candidates(gp_group_t *grp, const double *Xs, int64_t M) {
    if (!grp || !Xs) return fail(GP_ERR_ARG, "null argument");
    if (M < 1) return fail(GP_ERR_ARG, "M < 1");
    const int n = (int)grp->m.size();
    grp->lo.assign(n, 0);
    grp->hi.assign(n, 0);
    for (int i = 0; i < n; ++i) block_of(M, i, n, &grp->lo[i], &grp->hi[i]);
    grp->M = 0;
    int rc = for_members(grp, [&](int i) {
        if (grp->hi[i] == grp->lo[i]) return 0;   // fewer rows than members: this member sits the round out
        return gp_set_candidates(grp->m[i], Xs + grp->lo[i] * grp->m[i]->D, grp->hi[i] - grp->lo[i]);
    });
    if (rc) return rc;
    grp->M = M;
    return 0;
}

// lowest global row among the best values (np.argmax / np.argmin on the unsharded vector); pairs with idx < 0 are empty
static int merge_best(const std::vector<double> &v, const std::vector<int64_t> &ix, int sense, int64_t *idx, double *val) {
    bool have = false;
    double bv = 0.0;
    int64_t bi = -1;
    for (size_t r = 0; r < v.size(); ++r) {
        if (ix[r] < 0) continue;
        const bool better = !have || (sense > 0 ? v[r] > bv : v[r] < bv) || (v[r] == bv && ix[r] < bi);
        if (better) {
            have = true;
            bv = v[r];
            bi = ix[r];
        }
    }
    if (!have) return fail(GP_ERR_STATE, "no member produced a candidate");
    *idx = bi;
    *val = bv;
    return 0;
}

static int exchange_and_merge_best(gp_group *grp, std::vector<double> &v, std::vector<int64_t> &ix, int sense, int64_t *idx,
                                   double *val);

extern "C" int gp_group_acq_argbest(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int sense,
                         int64_t *idx, double *val) {
    if (!grp || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    if (grp->M < 1) return fail(GP_ERR_STATE, "gp_group_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    const int n = (int)grp->m.size();
    std::vector<double> v(n, sense > 0 ? -INFINITY : INFINITY);
    std::vector<int64_t> ix(n, -1);
    int rc = for_members(grp, [&](int i) {
        if (grp->hi[i] == grp->lo[i]) return 0;
        int64_t li = -1;
        int r = gp_acq_argbest(grp->m[i], type, par, fmin, y_mean, y_std, sense, &li, &v[i]);
        if (r == 0) ix[i] = grp->lo[i] + li;
        return r;
    });
    if (rc) return rc;
    return exchange_and_merge_best(grp, v, ix, sense, idx, val);
}

// The group's one collective, issued for ALL members from the calling thread inside ncclGroupStart / ncclGroupEnd: k records
// {double value, int64 global row} per member, every member receives all n k.  Every member is validated and has its record on
// the device BEFORE the first ncclAllGather is enqueued, so no member can be left waiting in a collective that a failed peer never
// joined (the earlier form -- one worker thread per device each calling gp_comm_allgather_* -- had that hang path).  If the
// grouped enqueue itself fails, the communicators are aborted and the group falls back to the host merge for good.
static int group_allgather(gp_group *grp, int k, const std::vector<double> &v, const std::vector<int64_t> &ix,
                           std::vector<double> &gv, std::vector<int64_t> &gi) {
    const int n = (int)grp->m.size();
    const size_t nd = 2 * (size_t)k;                       // doubles per member
    for (int i = 0; i < n; ++i) {
        gp_ctx *g = grp->m[i];
        if (g->dead) return fail(GP_ERR_STATE, "member %d: the library was shut down (gp_shutdown)", i);
        if (!g->comm) return fail(GP_ERR_STATE, "member %d holds no communicator", i);
    }
    std::vector<std::vector<double>> rec(n, std::vector<double>(nd)), out(n, std::vector<double>(nd * n));
    for (int i = 0; i < n; ++i) {
        gp_ctx *g = grp->m[i];
        HIPCHK(hipSetDevice(g->device));
        int rc;
        if ((rc = dev_realloc(&g->dComm, &g->capComm, 2L * GP_TOPK_MAX * (1 + 128)))) return rc;
        for (int j = 0; j < k; ++j) {
            rec[i][2 * j] = v[(size_t)i * k + j];
            memcpy(&rec[i][2 * j + 1], &ix[(size_t)i * k + j], 8);
        }
        HIPCHK(hipMemcpyAsync(g->dComm, rec[i].data(), sizeof(double) * nd, hipMemcpyHostToDevice, g->s));
    }
    ncclResult_t bad = ncclSuccess;
    ncclGroupStart();
    for (int i = 0; i < n; ++i) {
        gp_ctx *g = grp->m[i];
        ncclResult_t r = ncclAllGather(g->dComm, g->dComm + 2 * GP_TOPK_MAX, nd, ncclDouble, g->comm, g->s);
        if (r != ncclSuccess && bad == ncclSuccess) bad = r;
    }
    ncclResult_t rend = ncclGroupEnd();
    if (bad == ncclSuccess) bad = rend;
    if (bad != ncclSuccess) {
        for (gp_ctx *g : grp->m) {   // nothing may stay half-enqueued: abort every communicator, host merge from now on
            if (g->comm) ncclCommAbort(g->comm);
            g->comm = nullptr;
            g->nranks = 1;
        }
        grp->rccl = false;
        grp->rccl_note = std::string("grouped ncclAllGather -> ") + ncclGetErrorString(bad) + ": communicators aborted, host merge";
        return fail(GP_ERR_RCCL, "%s", grp->rccl_note.c_str());
    }
    int first = 0;
    for (int i = 0; i < n; ++i) {   // every member is drained even when an earlier one reported an error
        gp_ctx *g = grp->m[i];
        hipError_t e = hipSetDevice(g->device);
        if (e == hipSuccess)
            e = hipMemcpyAsync(out[i].data(), g->dComm + 2 * GP_TOPK_MAX, sizeof(double) * nd * n, hipMemcpyDeviceToHost, g->s);
        hipError_t es = hipStreamSynchronize(g->s);
        if (e == hipSuccess) e = es;
        if (e != hipSuccess && !first) first = fail(GP_ERR_HIP, "member %d: gather -> %s", i, hipGetErrorString(e));
    }
    if (first) return first;
    gv.resize((size_t)n * n * k);
    gi.resize((size_t)n * n * k);
    for (int i = 0; i < n; ++i)
        for (size_t r = 0; r < (size_t)n * k; ++r) {
            gv[(size_t)i * n * k + r] = out[i][2 * r];
            memcpy(&gi[(size_t)i * n * k + r], &out[i][2 * r + 1], 8);
        }
    return 0;
}

// Shared tail of the arg-best entry points: every member has its (value, global row) pair -- exchange (RCCL when the members hold
// communicators) and merge.
static int exchange_and_merge_best(gp_group *grp, std::vector<double> &v, std::vector<int64_t> &ix, int sense, int64_t *idx,
                                   double *val) {
    const int n = (int)grp->m.size();
    if (grp->rccl) {
        std::vector<double> gv;
        std::vector<int64_t> gi;
        // (the collective starts only once EVERY member has its pair, and is enqueued for all of them by this one thread)
        int rc = group_allgather(grp, 1, v, ix, gv, gi);
        if (rc) return rc;
        // every member holds all pairs now; member 0's copy is merged (the others must equal it)
        for (int i = 1; i < n; ++i)
            if (memcmp(&gv[(size_t)i * n], &gv[0], sizeof(double) * n) || memcmp(&gi[(size_t)i * n], &gi[0], sizeof(int64_t) * n))
                return fail(GP_ERR_RCCL, "members disagree on the gathered pairs");
        v.assign(gv.begin(), gv.begin() + n);
        ix.assign(gi.begin(), gi.begin() + n);
    }
    return merge_best(v, ix, sense, idx, val);
}

// The local-penalisation acquisition over the whole table (run.py:1238-1257: penalised scores of the candidate table, arg-max,
// rows already taken masked): every member scores its block with gp_acq_lp_argbest, the excluded GLOBAL rows handed to the member
// whose block holds them.
extern "C" int gp_group_acq_lp_argbest(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int transform,
                            const double *Xb, int nb, const double *r_x0, const double *s_x0, int sense, const int64_t *exclude,
                            int nex, int64_t *idx, double *val) {
    if (!grp || !idx || !val || (nex > 0 && !exclude)) return fail(GP_ERR_ARG, "null argument");
    if (grp->M < 1) return fail(GP_ERR_STATE, "gp_group_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (nex < 0 || nex > 256) return fail(GP_ERR_ARG, "too many excluded rows (<= 256)");
    for (int e = 0; e < nex; ++e)
        if (exclude[e] < 0 || exclude[e] >= grp->M) return fail(GP_ERR_ARG, "excluded row out of range");
    const int n = (int)grp->m.size();
    std::vector<double> v(n, sense > 0 ? -INFINITY : INFINITY);
    std::vector<int64_t> ix(n, -1);
    int rc = for_members(grp, [&](int i) {
        if (grp->hi[i] == grp->lo[i]) return 0;
        std::vector<int64_t> mine;
        for (int e = 0; e < nex; ++e)
            if (exclude[e] >= grp->lo[i] && exclude[e] < grp->hi[i]) mine.push_back(exclude[e] - grp->lo[i]);
        if ((long)mine.size() == grp->hi[i] - grp->lo[i]) return 0;   // every row of this block is taken already
        int64_t li = -1;
        int r = gp_acq_lp_argbest(grp->m[i], type, par, fmin, y_mean, y_std, transform, Xb, nb, r_x0, s_x0, sense, mine.data(),
                                  (int)mine.size(), &li, &v[i]);
        if (r == 0) ix[i] = grp->lo[i] + li;
        return r;
    });
    if (rc) return rc;
    return exchange_and_merge_best(grp, v, ix, sense, idx, val);
}

// ---- the merges as host-only entry points (no device, no group): what every layout applies to the gathered pairs -------------
extern "C" int gp_merge_best(int n, const double *vals, const int64_t *idxs, int sense, int64_t *idx, double *val) {
    if (n < 1 || !vals || !idxs || !idx || !val) return fail(GP_ERR_ARG, "bad argument");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    return merge_best(std::vector<double>(vals, vals + n), std::vector<int64_t>(idxs, idxs + n), sense, idx, val);
}

static void merge_topk(const std::vector<double> &v, const std::vector<int64_t> &ix, int sense, int k, int64_t *idx, double *val);

extern "C" int gp_merge_topk(int n, const double *vals, const int64_t *idxs, int sense, int k, int64_t *idx, double *val) {
    if (n < 1 || k < 1 || !vals || !idxs || !idx || !val) return fail(GP_ERR_ARG, "bad argument");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    merge_topk(std::vector<double>(vals, vals + n), std::vector<int64_t>(idxs, idxs + n), sense, k, idx, val);
    return 0;
}

extern "C" int gp_group_acq_topk(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int sense, int k,
                      int64_t *idx, double *val) {
    if (!grp || !idx || !val) return fail(GP_ERR_ARG, "null argument");
    if (grp->M < 1) return fail(GP_ERR_STATE, "gp_group_set_candidates first");
    if (sense != 1 && sense != -1) return fail(GP_ERR_ARG, "sense must be +1 or -1");
    if (k < 1 || k > GP_TOPK_MAX) return fail(GP_ERR_ARG, "k out of range (1..%d)", GP_TOPK_MAX);
    const int n = (int)grp->m.size();
    const double empty = sense > 0 ? -INFINITY : INFINITY;
    std::vector<double> v((size_t)n * k, empty);
    std::vector<int64_t> ix((size_t)n * k, -1);
    std::vector<double> gv;
    std::vector<int64_t> gi;
    int rc = for_members(grp, [&](int i) {
        if (grp->hi[i] == grp->lo[i]) return 0;
        int64_t *ii = &ix[(size_t)i * k];
        int r = gp_acq_topk(grp->m[i], type, par, fmin, y_mean, y_std, sense, k, ii, &v[(size_t)i * k]);
        if (r) return r;
        for (int j = 0; j < k; ++j)
            if (ii[j] >= 0) ii[j] += grp->lo[i];
        return 0;
    });
    if (rc) return rc;
    if (grp->rccl) {
        if ((rc = group_allgather(grp, k, v, ix, gv, gi))) return rc;
        for (int i = 1; i < n; ++i)
            if (memcmp(&gv[(size_t)i * n * k], &gv[0], sizeof(double) * n * k) ||
                memcmp(&gi[(size_t)i * n * k], &gi[0], sizeof(int64_t) * n * k))
                return fail(GP_ERR_RCCL, "members disagree on the gathered pairs");
        v.assign(gv.begin(), gv.begin() + (size_t)n * k);
        ix.assign(gi.begin(), gi.begin() + (size_t)n * k);
    }
    merge_topk(v, ix, sense, k, idx, val);
    return 0;
}

// k rounds of the lowest-index arg-best over the gathered pairs: a stable sort by (value, global row); pairs with idx < 0 are
// empty slots, a tail that cannot be filled is idx = -1
static void merge_topk(const std::vector<double> &v, const std::vector<int64_t> &ix, int sense, int k, int64_t *idx, double *val) {
    const double empty = sense > 0 ? -INFINITY : INFINITY;
    std::vector<char> used(v.size(), 0);
    for (int j = 0; j < k; ++j) {
        long best = -1;
        for (size_t r = 0; r < v.size(); ++r) {
            if (used[r] || ix[r] < 0) continue;
            if (best < 0 || (sense > 0 ? v[r] > v[best] : v[r] < v[best]) || (v[r] == v[best] && ix[r] < ix[best])) best = (long)r;
        }
        if (best < 0) {
            idx[j] = -1;
            val[j] = empty;
        } else {
            used[best] = 1;
            idx[j] = ix[best];
            val[j] = v[best];
        }
    }
}

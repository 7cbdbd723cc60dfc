/*
 * gphip.h -- C ABI of libgphip.so: the MI355X-native (gfx950) exact-GP hot path.
 *
 * Drop-in boundary for the GPy / GPyOpt path named in BASELINE.json:north_star.
 * Each entry point cites the reference interface it replaces (paths relative
 * to the reference tree, file:line).  Plain pointers and sizes only; all
 * matrices are float64, C-contiguous (row-major), owned by the caller.  The
 * library copies host->device, owns all device memory inside the opaque
 * gp_t, writes results into caller-allocated buffers and keeps no host
 * pointer after returning.  One gp_t per device; calls are synchronous; a
 * gp_t is not thread-safe, distinct gp_t may be used from distinct threads
 * (gp_group_* below does exactly that for a single-threaded caller).
 *
 * Return codes: 0 = OK; k > 0 = leading minor k of Ky is not positive
 * definite even after the reference's jitter ladder (the host raises
 * numpy.linalg.LinAlgError with the reference's messages, GPy/GPy/util/
 * linalg.py:62-75); < 0 = bad argument / HIP / RCCL error (message via
 * gp_last_error()).  The library never aborts the process.
 */
#ifndef GPHIP_H
#define GPHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gp_ctx gp_t;

/* kernel families: GPy/GPy/kern/src/rbf.py:12-57, stationary.py:546-579 */
#define GP_KERNEL_RBF 0
#define GP_KERNEL_MATERN52 1

/* acquisitions: GPyOpt/GPyOpt/acquisitions/{EI,LCB,MPI}.py */
#define GP_ACQ_EI 0
#define GP_ACQ_LCB 1
#define GP_ACQ_MPI 2

/* error codes (< 0) */
#define GP_ERR_ARG (-1)
#define GP_ERR_HIP (-2)
#define GP_ERR_STATE (-3)
#define GP_ERR_RCCL (-4)
#define GP_ERR_NOT_PD_DIAG (-5) /* "not pd: non-positive diagonal elements", linalg.py:63-64 */

#define GP_TOPK_MAX 64 /* largest k of gp_acq_topk / gp_comm_allgather_topk */

/* ---- library / device -------------------------------------------------- */
const char *gp_last_error(void);
const char *gp_version(void);
int gp_device_count(int *count);
/* fills name (<= cap bytes), compute units, HBM bytes */
int gp_device_info(int device, char *name, int cap, int *cus, int64_t *hbm_bytes);

/* ---- lifetime ---------------------------------------------------------- */
/* Replaces constructing GPy.models.GPRegression / GP (GPy/GPy/models/gp_regression.py:29-36,
 * GPy/GPy/core/gp.py:38-110): an empty model bound to one device. */
int gp_create(gp_t **out, int device);
int gp_destroy(gp_t *gp);
/* Ordered library shutdown: waits for every device the library used, destroys the events of the live contexts and then
 * the per-device stream set (which otherwise lives for the whole process: hardware queues keep their command-processor
 * pipe only while they are never re-created).  Registered with atexit() at the first gp_create, so a process that
 * simply exits -- also under rocprofv3 -- tears the queues down before the HIP runtime's own exit handlers run.
 * Contexts still alive afterwards only accept gp_destroy. */
int gp_shutdown(void);

/* GP.set_XY (GPy/GPy/core/gp.py:202-238): X[N,D], Y[N,P] row-major, copied H2D. 1 <= P <= 128. */
int gp_set_data(gp_t *gp, const double *X, const double *Y, int64_t N, int D, int P);

/* Kernel + Gaussian-noise hyper-parameters in natural space
 * (Stationary.__init__ stationary.py:61-82; Gaussian variance likelihoods/gaussian.py:43).
 * lengthscale has 1 entry (ard = 0) or D entries (ard = 1). */
int gp_set_params(gp_t *gp, int kernel, int ard, double variance, const double *lengthscale, double noise);

/* The fork's mixed-variable "Gower" covariance (GPy/GPy/kern/src/stationary.py:116-135 of the reference tree):
 * K = prod_d K_of_r(r_d), r_d = |x_d - x'_d| / range[d] on continuous dimensions and (x_d != x'_d) on discrete
 * ones (is_discrete[d] != 0).  enable = 0 restores the Euclidean kernel.  Kdiag remains `variance`, as in the fork.
 * Predictive gradients of a Gower model (gp_predict_grad, gp_acq_grad, gp_acq_lp_grad) are the fork's: Euclidean
 * Stationary.gradients_X on the kernel's own lengthscale (stationary.py:336-364, _inv_dist :251-258) with the Gower
 * K(Xs, X) inside dv/dx (gp.py:451-452) -- what run.py:1206-1225,1244 runs L-BFGS and estimate_L on.  gp_lml_grad and
 * gp_fit_grad likewise return what the fork's update_gradients_full computes (stationary.py:218-238): the Gower K as the
 * weight of the variance gradient, Euclidean dK/dr on the kernel's own lengthscale in the lengthscale gradients.  Those
 * are NOT derivatives of the Gower model's LML (K does not depend on that lengthscale at all, and depends on the variance
 * as variance^D); the host layer derives the true gradient from them (D x the variance entry, 0 for the lengthscale) unless
 * told to follow the fork (GPRegression.gower_gradients). */
int gp_set_gower(gp_t *gp, int enable, const int *is_discrete, const double *range);

/* ---- fit ---------------------------------------------------------------
 * ExactGaussianInference.inference (GPy/GPy/inference/latent_function_inference/
 * exact_gaussian_inference.py:37-74) minus the gradient terms:
 *   K = kern.K(X) (stationary.py:107-193); Ky = K + (noise + 1e-8) I (:55-56);
 *   L = jitchol(Ky, maxtries) with the reference's jitter ladder (linalg.py:56-81);
 *   logdet = 2 sum log L_ii (linalg.py:208); alpha = Ky^-1 Y (dpotrs, :60);
 *   lml = 0.5 (-N P log 2pi - P logdet - sum(alpha * Y)) (:62).
 * Outputs: *lml, *logdet, *jitter_used (0 when the first dpotrf succeeds). */
int gp_fit(gp_t *gp, int maxtries, double *lml, double *logdet, double *jitter_used);

/* lml / logdet / jitter of the last fit (or of the root's fit after gp_comm_bcast_fit). */
int gp_get_fit_state(gp_t *gp, double *lml, double *logdet, double *jitter);
/* Posterior.woodbury_vector (posterior.py:198-214): alpha[N,P]. */
int gp_get_alpha(gp_t *gp, double *alpha);
/* Posterior.woodbury_chol: L[N,N] row-major, lower triangle (upper written as 0). */
int gp_get_chol(gp_t *gp, double *L);
/* Posterior.woodbury_inv (posterior.py:176-196): Ky^-1 [N,N], symmetric.  Computed lazily (potri). */
int gp_get_woodbury_inv(gp_t *gp, double *Wi);
/* kern.K(X) without noise (stationary.py:107-140), for parity tests of the K-build kernel: K[N,N]. */
int gp_kernel_matrix(gp_t *gp, double *K);
/* kern.K(X, X2) -- the cross covariance of the kernel contract (Kern.K, GPy/GPy/kern/src/kern.py:119;
 * Stationary.K with X2 given, stationary.py:107-140: _unscaled_dist's X2 branch :168-173 has no diagonal fix, the Gower
 * branch :116-135 likewise takes X2).  X = the training inputs of gp_set_data, X2[M2, D] row-major (caller-owned);
 * K[N, M2] row-major.  Uses the current kernel parameters; leaves the fit and the resident candidates untouched. */
int gp_cross_kernel_matrix(gp_t *gp, const double *X2, int64_t M2, double *K);

/* GP.parameters_changed gradient push-down (gp.py:268-269):
 *   dL_dK = 0.5 (alpha alpha^T - P Ky^-1) (exact_gaussian_inference.py:70);
 *   dnoise = sum diag(dL_dK) (gaussian.py:78-79);
 *   dvariance, dlengthscale[1 or D] = Stationary.update_gradients_full (stationary.py:218-238).
 * Natural-space gradients of the LML; the Logexp chain rule stays on the host. Requires gp_fit. */
int gp_lml_grad(gp_t *gp, double *dvariance, double *dlengthscale, double *dnoise);

/* grad_dict['dL_dK'] of ExactGaussianInference.inference (exact_gaussian_inference.py:70,74):
 *   dL_dK[N,N] = 0.5 (alpha alpha^T - P Ky^-1), the matrix GP.parameters_changed hands to
 *   kern.update_gradients_full (gp.py:269) -- for reference-side kernels that reduce it on the host.
 *   (gp_lml_grad reduces the same matrix on the device without materialising it.)  Requires gp_fit. */
int gp_get_dl_dk(gp_t *gp, double *dL_dK);

/* gp_fit + gp_lml_grad as ONE call -- what every L-BFGS evaluation of the hyper-parameter loop asks for
 * (Model.objective_function + objective_function_gradients, GPy/GPy/core/model.py:96-127; GPyOpt
 * models/gpmodel.py:88-93).  The first "pipe_stages_grad" stages of the solve for L^-T (dpotri, linalg.py:127-145)
 * ride behind the factorisation's latency-bound tail.  Same results as the two calls in sequence. */
int gp_fit_grad(gp_t *gp, int maxtries, double *lml, double *logdet, double *jitter_used, double *dvariance,
                double *dlengthscale, double *dnoise);

/* ---- predict -----------------------------------------------------------
 * Candidates Xs[M,D] are made resident once; the calls below then run on them. */
int gp_set_candidates(gp_t *gp, const double *Xs, int64_t M);

/* PosteriorExact._raw_predict (posterior.py:273-302, full_cov = False) followed by
 * Gaussian.predictive_values (likelihoods/gaussian.py:102-110) when include_noise != 0:
 *   mean[M,P] = K(Xs,X) alpha;  var[M] = variance - sum_rows (L^-1 K(X,Xs))^2 (+ noise).  No clipping. */
int gp_predict(gp_t *gp, int include_noise, double *mean, double *var);

/* gp_fit + gp_predict on the resident candidates as ONE call.  The first "pipe_stages" (3 at N = 16384) panel stages of
 * the candidate solve ride behind the factorisation once "pipe_start_pct" % (default: 32 up to 24 panels, 40 beyond) of its panels are done --
 * from there the factorisation's latency chain leaves CUs idle -- and the rest run after it.  Bitwise the results of
 * the two calls in sequence; 5 % faster at N=16384, M=10^4 (BO.suggest_next_locations always runs the two back to
 * back: GPyOpt/GPyOpt/core/bo.py:236-254 then acquisitions/base.py:33-39). */
int gp_fit_predict(gp_t *gp, int maxtries, int include_noise, double *lml, double *logdet, double *jitter_used,
                   double *mean, double *var);

/* full_cov = True branch (posterior.py:280-284): cov[M,M] = K(Xs) - tmp^T tmp (+ noise I). */
int gp_predict_full_cov(gp_t *gp, int include_noise, double *mean, double *cov);

/* GP.posterior_samples_f (gp.py:581-609) on the resident candidates: the full posterior covariance
 * (posterior.py:280-284, + noise I when include_noise) is factored on the device, C C^T = cov, under GPy's jitter
 * ladder (jitchol, linalg.py:56-81; maxtries as there), and applied to the caller's standard normals Z[S,M]:
 *   dev[S,M] = (C z_s)^T;  a draw of output d is mean[:,d] + dev[s,:].  mean[M,P] may be NULL.
 * M <= "mc_max".  Return codes as gp_fit (k > 0: not positive definite even with jitter). */
int gp_posterior_samples(gp_t *gp, int include_noise, const double *Z, int S, int maxtries, double *mean, double *dev,
                         double *jitter_used);

/* GP.predictive_gradients (gp.py:407-454): dmdx[M,D,P], dvdx[M,D].  dvdx = NULL: the mean's gradients alone
 * (gradients_X(alpha^T, X*, X), gp.py:433-438) -- what estimate_L maximises over 500 + N points
 * (GPyOpt/GPyOpt/core/evaluators/batch_local_penalization.py:55-64); they need neither Ky^-1 nor K(X*, X) Ky^-1. */
int gp_predict_grad(gp_t *gp, double *dmdx, double *dvdx);

/* GPModel.get_fmin (GPyOpt/GPyOpt/models/gpmodel.py:125-129): min over the training
 * inputs of the posterior mean.  Cached per fit (the reference recomputes it per call). */
int gp_fmin(gp_t *gp, double *fmin);

/* AcquisitionBase.acquisition_function on the resident candidates
 * (GPyOpt/GPyOpt/acquisitions/base.py:33-39 with constant cost and no constraints):
 * GPModel.predict (gpmodel.py:95-112: var clipped at 1e-10, s = sqrt(v), with noise),
 * get_quantiles (util/general.py:113-129), EI.py:32-40 / LCB.py:31-37 / MPI.py:32-40.
 * par = jitter (EI, MPI) or exploration weight (LCB).  Requires P == 1.
 * y_mean / y_std undo a host-side Standardize normalizer (1 / 0 when none).
 * out[M] holds the NEGATED acquisition, as the reference returns it. */
int gp_acq(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std, double *out);

/* Same scores reduced on the device: sense = +1 -> argmax of out, -1 -> argmin of out
 * (run.py:1240-1241 takes argmax; anchor_points_generator.py:61 takes the smallest);
 * ties resolve to the lowest index (NumPy argmax/argmin). idx is relative to this gp's
 * candidate set. */
int gp_acq_argbest(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std,
                   int sense, int64_t *idx, double *val);

/* The k best scores in order (AnchorPointsGenerator.get, GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59-61:
 * argsort(scores)[:num_anchor]): idx[k], val[k]; equal scores come out lowest index first; when M < k the tail is
 * idx = -1.  1 <= k <= GP_TOPK_MAX. */
int gp_acq_topk(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std, int sense, int k,
                int64_t *idx, double *val);

/* acquisition_function_withGradients (base.py:42-50; EI.py:42-51, LCB.py:39-46, MPI.py:42-51):
 * out[M] negated value, dout[M,D] negated gradient. */
int gp_acq_grad(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std,
                double *out, double *dout);

/* Local-penalisation batch acquisition on the resident candidates (the evaluator run.py:1219,1238-1257 uses):
 * AcquisitionLP._penalized_acquisition (GPyOpt/GPyOpt/acquisitions/LP.py:70-89):
 *   out = -T(acq(x)) - sum_k log Phi((|x - Xb_k| - r_x0[k]) / s_x0[k]),  T = log(. + 1e-50) (transform 0) or
 *   log(softplus(.)) (transform 1); acq is the base EI / LCB / MPI value; r_x0, s_x0 from
 *   _hammer_function_precompute (LP.py:49-62), computed by the host from gp_predict at the batch points.
 * nb = 0 gives the un-penalised log acquisition.  out[M] as the reference returns it (to be minimised). */
int gp_acq_lp(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std, int transform,
              const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out);
/* AcquisitionLP.acquisition_function_withGradients (LP.py:112-140) on the resident candidates: out[M] as gp_acq_lp,
 * dout[M,D] = scale * d(-acq)/dx - sum_k pen_k, with scale = 1 / acq (transform 0) or 1 / (softplus(acq) (1 + exp(-acq)))
 * (transform 1) and pen_k = exp(-z^2/2) / (s_x0[k] sqrt(2 pi) Phi(z) |x - Xb_k|), z = (|x - Xb_k| - r_x0[k]) / s_x0[k],
 * taken as 0 where Phi(z) < 1e-50.  The reference's _d_hammer_function (LP.py:91-103) sums that SCALAR over the batch and
 * subtracts it from every input dimension (the direction (x - Xb_k) / |x - Xb_k| is missing there); reproduced as is so that
 * an L-BFGS run over this function follows the reference's. */
int gp_acq_lp_grad(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std, int transform,
                   const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out, double *dout);
/* arg-best of the same vector, skipping the rows listed in exclude (run.py:1249-1252). */
int gp_acq_lp_argbest(gp_t *gp, int type, double par, double fmin, double y_mean, double y_std, int transform,
                      const double *Xb, int nb, const double *r_x0, const double *s_x0, int sense,
                      const int64_t *exclude, int nex, int64_t *idx, double *val);

/* ---- a handful of locations per call: the acquisition optimiser's inner loop --------------------------------
 * scipy's L-BFGS-B evaluates the acquisition ONE location per call, hundreds of times between two fits
 * (GPyOpt/GPyOpt/optimization/optimizer.py:36-61 -> acquisitions/base.py:33-50, LP.py:105-140 -> models/gpmodel.py:95-142
 * -> GPy/GPy/core/gp.py:297-354,407-454).  These two entry points are gp_set_candidates followed by the batched calls
 * named below, as ONE call taking the locations Xs[M, D] by value; for M <= option "small_m" (8) locations of a
 * single-output model with min(M, 4) D <= 128 they run as three launches (two for a value-only call) per pass of up to
 * four locations over the explicit inverse factor L^-1 (dtrtri, linalg.py:217-227; built once per fit -- at the first
 * call for N <= 4096, above that after the first N / 768 calls, which go through substitutions against L: option
 * "rows_build" -- at half the work of Ky^-1) with no copy commands (csrc/onerow.hip), otherwise through the batched calls
 * themselves.  Same results either way to rounding
 * (tests/test_gpu_rows.py: 1e-9 on well-conditioned models, the north-star 1e-6 against the oracle everywhere); the
 * resident candidate block of gp_set_candidates is unspecified afterwards.
 *
 * gp_predict_rows = gp_predict(include_noise, mean[M], var[M]) and, when dmdx / dvdx are given,
 *                   gp_predict_grad(dmdx[M, D], dvdx[M, D]).  mean / var may be NULL.  P = 1 layouts.  dmdx alone (dvdx,
 *                   mean, var NULL) = gp_predict_grad(dmdx, NULL): the mean's gradient, one pass over the training points
 *                   with no inverse factor behind it -- estimate_L's inner call (batch_local_penalization.py:55-67).
 * gp_acq_rows     = gp_acq / gp_acq_grad (lp = 0) or gp_acq_lp / gp_acq_lp_grad (lp = 1, with transform, Xb, nb, r_x0,
 *                   s_x0 as there): out[M], and dout[M, D] when given. */
int gp_predict_rows(gp_t *gp, const double *Xs, int64_t M, int include_noise, double *mean, double *var, double *dmdx,
                    double *dvdx);
int gp_acq_rows(gp_t *gp, const double *Xs, int64_t M, int type, double par, double fmin, double y_mean, double y_std,
                int lp, int transform, const double *Xb, int nb, const double *r_x0, const double *s_x0, double *out,
                double *dout);
/* how many of those calls took the fused path / the batched calls since gp_create (route checks of the tests) */
int gp_rows_stats(gp_t *gp, int64_t *fused, int64_t *fallback);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) ---------------------
 * The candidate table shards across ranks; every rank holds a replica of the
 * fitted model.  uid is ncclUniqueId (128 bytes) produced on rank 0 and carried
 * to the other ranks by the launcher (any side channel). */
int gp_comm_unique_id(char *uid128);
int gp_comm_init(gp_t *gp, const char *uid128, int rank, int nranks);
int gp_comm_destroy(gp_t *gp);
/* rank / size as the RCCL communicator reports them (ncclCommUserRank, ncclCommCount) */
int gp_comm_info(gp_t *gp, int *rank, int *nranks);
/* all-gather of one (val, idx) pair per rank: vals[nranks], idxs[nranks]. */
int gp_comm_allgather_best(gp_t *gp, double val, int64_t idx, double *vals, int64_t *idxs);
/* ncclGetVersion of the librccl the process resolved (e.g. 22606): recorded by bench.py beside the mapped library path */
int gp_comm_version(int *version);
/* the top-k variant ("gather 8 x k pairs"): every rank contributes its k best (val, global idx) pairs
 * (idx < 0 marks an empty slot); all_vals / all_idxs [nranks * k], rank-major. */
int gp_comm_allgather_topk(gp_t *gp, int k, const double *vals, const int64_t *idxs, double *all_vals,
                           int64_t *all_idxs);
/* broadcast of a fitted model from root to all ranks: L, alpha, z, inverse tiles and the host-side scalars of the fit
 * (jitter, LML, log det), so that gp_fmin and gp_get_fit_state agree on every rank. */
int gp_comm_bcast_fit(gp_t *gp, int root);
/* Host-only self-test of gp_comm_bcast_fit's receiver side (no device, no communicator): packs the root's record from
 * root_state = {jitter, lml, logdet}, applies it to a scratch context standing in for a receiving rank with stale
 * results, and reports the receiver's {jitter, lml, logdet} and {fitted, fmin_valid, wi_valid, invp_valid, lr_valid,
 * predicted}.  Exists because RCCL with 2 ranks cannot run on a 1-GPU box (tests/test_host_logic.py). */
int gp_comm_selftest_fit_record(const double *root_state, double *state_out, int *flags_out);

/* ---- single-process multi-GPU: a device group ----------------------------------------------------------------------
 * The reference's BO loop is ONE Python process (GPyOpt/GPyOpt/core/bo.py:73-168, run.py:1207-1258) that scores a candidate
 * table and takes argmax / argsort()[:5] (run.py:1240-1241, GPyOpt/GPyOpt/optimization/anchor_points_generator.py:59-61).
 * A group lets that one caller thread use several GPUs of the node: one gp_t per entry of devices[] (the model replicated:
 * the fit does not shard, every member factors its replica, the devices side by side), the table cut into contiguous row
 * blocks -- member i of n takes rows [i M/n .., first M % n members one row longer] -- and the per-block winners merged with
 * NumPy's lowest-index tie rule.  With all devices different the members hold the communicators of ncclCommInitAll and
 * the winners travel by ONE grouped RCCL all-gather over xGMI, enqueued for every member by the calling thread inside
 * ncclGroupStart / ncclGroupEnd after every member has been validated (no member can wait on a peer that never joined; a
 * failed enqueue aborts the communicators and the group merges on the host from then on); with a device listed more than
 * once (a one-GPU box rehearsing the logic) the pairs are merged on the host, which gp_group_info reports.
 * STATUS: the RCCL route has not run on more than one device in any record of this repository -- no multi-GPU box was
 * available to the builder; what the tests execute is the host-merge route (devices = {0, 0, ...}) and the
 * single-member communicator.  Calls are synchronous; one group per caller thread. */
typedef struct gp_group gp_group_t;
int gp_group_create(gp_group_t **out, int ndev, const int *devices);
int gp_group_destroy(gp_group_t *grp);
/* ndev, whether the winners travel by RCCL, and a short note (<= cap bytes) saying which exchange is in use and why */
int gp_group_info(gp_group_t *grp, int *ndev, int *uses_rccl, char *note, int cap);
/* member i's context (owned by the group), e.g. for gp_get_alpha / gp_last_phases on one replica */
int gp_group_member(gp_group_t *grp, int i, gp_t **member);
/* gp_set_data / gp_set_params / gp_set_gower / gp_set_option applied to every member */
int gp_group_set_data(gp_group_t *grp, const double *X, const double *Y, int64_t N, int D, int P);
int gp_group_set_params(gp_group_t *grp, int kernel, int ard, double variance, const double *lengthscale, double noise);
int gp_group_set_gower(gp_group_t *grp, int enable, const int *is_discrete, const double *range);
int gp_group_set_option(gp_group_t *grp, const char *name, int64_t value);
/* gp_fit on every member at once; the scalars are member 0's, and a replica whose LML or jitter differs from member 0's in
 * any bit fails the call (GP_ERR_STATE) */
int gp_group_fit(gp_group_t *grp, int maxtries, double *lml, double *logdet, double *jitter_used);
int gp_group_fmin(gp_group_t *grp, double *fmin);
/* the WHOLE candidate table Xs[M, D]; each member keeps its block resident */
int gp_group_set_candidates(gp_group_t *grp, const double *Xs, int64_t M);
/* gp_acq_argbest / gp_acq_topk over the whole table: idx are rows of the table passed to gp_group_set_candidates */
int gp_group_acq_argbest(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int sense,
                         int64_t *idx, double *val);
/* gp_acq_lp_argbest over the whole table (the local-penalisation batch loop of run.py:1238-1257): exclude[] holds rows of the
 * table passed to gp_group_set_candidates */
int gp_group_acq_lp_argbest(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int transform,
                            const double *Xb, int nb, const double *r_x0, const double *s_x0, int sense,
                            const int64_t *exclude, int nex, int64_t *idx, double *val);
int gp_group_acq_topk(gp_group_t *grp, int type, double par, double fmin, double y_mean, double y_std, int sense, int k,
                      int64_t *idx, double *val);

/* The merge every layout applies to gathered (value, global row) pairs, as host-only helpers (no device needed): the best pair /
 * the k best in order, equal values lowest row first (np.argmax / np.argmin / a stable argsort on the unsharded vector); pairs with
 * idx < 0 are empty slots; a top-k tail that cannot be filled is idx = -1. */
int gp_merge_best(int n, const double *vals, const int64_t *idxs, int sense, int64_t *idx, double *val);
int gp_merge_topk(int n, const double *vals, const int64_t *idxs, int sense, int k, int64_t *idx, double *val);

/* ---- measurement ---------------------------------------------------------
 * Phase timings of the last gp_fit / gp_predict measured with HIP events on the
 * library's stream; names[i] is a static string, ms[i] milliseconds, flops[i] the
 * algorithmic flop count of the phase (0 for bandwidth phases), bytes[i] its
 * algorithmic bytes.  Returns the number of phases written (<= cap). */
int gp_last_phases(gp_t *gp, int cap, const char **names, double *ms, double *flops, double *bytes);
/* Dominant-kernel accounting (the fp64 MFMA GEMM): launches, summed device time (ms, HIP events around the launches
 * when profiling is on), algorithmic flops.  With the default threshold the events bracket exactly the launches of one
 * kernel symbol, gemm_nt_kernel<1, 128, 4, false, 128> (C -= A B^T, >= 1400 output tiles), so that the average agrees with that
 * symbol's row of a rocprofv3 --stats summary; "profile_min_tiles" < 1024 brackets every launch above it instead
 * (tracing tools).  Reset by gp_profile(gp, 1).  gp_profile(gp, 2) brackets the launches of the second fp64 symbol instead,
 * gemm_nt_kernel<1, 64, 2, false, 64> (the same update as 64 x 64 work units: the factorisation chain's in-panel and
 * look-ahead updates, short candidate updates); events around those ~100-us launches stall the chain, so this is for a
 * pass outside any timed region. */
int gp_profile(gp_t *gp, int on);
int gp_gemm_stats(gp_t *gp, int64_t *launches, double *ms, double *flops);
/* the same for the residue GEMM of "emulate_fp64" (rns_gemm256_kernel, every launch): launches, summed device time, int8
 * operations (2 per multiply-add of the blocks a launch computes) */
int gp_rns_stats(gp_t *gp, int64_t *launches, double *ms, double *ops);
/* wall time (ms) during which at least one profiled launch was running: the union of their intervals.  Launches of
 * gp_fit_predict overlap, so the SUM of durations above counts shared time twice; flops / busy is the kernel's
 * throughput while it runs. */
int gp_gemm_busy(gp_t *gp, double *busy_ms);
/* per-launch record of the profiled GEMM launches: output tiles, K (negative: triangular contraction), ms */
int gp_gemm_trace(gp_t *gp, int cap, int64_t *tiles, int *K, double *ms);
int gp_synchronize(gp_t *gp);
/* Tunables (none changes a result beyond rounding; tests/test_gpu_random_shapes.py sweeps the blocking ones):
 *   "panel_tiles"        outer panel width of the factorisation and of the inverted panels, in 128-tiles (default 6)
 *   "lookahead"          0/1: one panel of look-ahead on separate streams (default 1)
 *   "lookahead_min_tiles"  gp_fit alone: matrices of at most this many 128-tiles (default 40, N <= 5120) take the same
 *                        factorisation on ONE stream (bitwise the same factor; the look-ahead's per-panel events cost more there)
 *   "mc_max"             candidate rows per chunk (default 16384)
 *   "small_below", "chain_small_below"   launches with fewer 128-tiles run as 64x64 work units (1400 / 400 on the chain)
 *   "waves8", "stagger", "trsm_rows64", "supertile"   GEMM launch shape
 *   "pipe_stages", "pipe_start_pct"   gp_fit_predict: how many candidate stages ride behind the factorisation (0 = automatic:
 *                        14 % of the panels, 3 at N = 16384) and
 *                        after which share of its panels they are released (-1 = automatic: 32 up to 24 panels, 40 beyond)
 *   "pipe_stages_grad", "pipe_start_pct_grad"   the same for gp_fit_grad (0 = automatic: 36 % of the panels; 40)
 *   "lauum_panels"       Ky^-1 product accumulated per k-panel (default 1)
 *   "pair_tri"           triangular-K products: column tiles paired for equal contraction length (default 2)
 *   "fmin_direct"        gp_fmin through the N^2 product K(X,X) alpha instead of y - d alpha (default 0)
 *   "emulate_fp64"       0/1 (default 0; environment GPHIP_EMULATE_FP64 sets the default of new contexts): the candidate
 *                        solve's updates T[:, > J] -= S_J L[> J, J]^T (dtrtrs, posterior.py:294; 95 % of gp_predict's flops)
 *                        and, with "emulate_fit" (default 1), the factorisation's trailing update (dsyrk/dgemm inside dpotrf,
 *                        linalg.py:58) and Ky^-1 (dtrtri + dpotri, linalg.py:127-145) run on the int8 matrix cores in residue form (csrc/rns.hip): operands as 52-bit
 *                        fixed point, 14 moduli, exact int32 accumulation, CRT reconstruction once per column.  Same results
 *                        to ~1e-12 (only the operands are rounded, to one fp64 ulp of the largest entry); diagonal tiles,
 *                        panel solves and all reductions stay true fp64; gp_fit_predict then runs fit and predict one after
 *                        the other.  Non-finite data cannot be put into fixed point: the call then repeats the step in
 *                        true fp64 (NaNs propagate as in the reference).  "rns_group" / "rns_group_fit" (default 8, 1..16): panels per residue launch of the
 *                        candidate solve / of the trailing update; "rns_interleave" (default 1, process-wide): the eight
 *                        XCDs work on one modulus at a time.  Results do not depend on these three.
 *   "inner_tiles"        tile columns per step of the in-panel factorisation: 1 (default) = diagonal tile, panel solve, K = 128 update;
 *                        2 = a 256 x 256 diagonal block per launch (potrf_pair_kernel), both tile columns of the rows below per launch
 *                        (trsm2_kernel), one K = 256 update -- half the dependent launches, the same wall time;
 *                        "inner_min_rows": two columns only while at least this many row tiles lie below (default 0)
 *   "own_keep_per_row"   look-ahead factorisation: of a trailing update with n row tiles below the look-ahead panel the masked bulk stream keeps
 *   "own_keep_base"      own_keep_base + own_keep_per_row * n * panel_tiles / 6 tiles (what lasts as long as the chain is busy with that panel; defaults 200 and
 *                        36); the rest, the far tile columns, is updated on the chain stream -- every CU -- in the window in which the chain
 *                        would wait for the bulk stream.  Same bits as without; own_keep_per_row = 0 switches it off.
 *                        "own_keep_pipe_pct" (default 0): the kept share in per cent of the rule's once the pipelined candidate stages of
 *                        gp_fit_predict / gp_fit_grad share the bulk stream's CUs
 *   "small_m"            up to this many candidates (default 8, 0 = never) are solved as matrix-vector work bound by one read of L
 *                        (csrc/smallm.hip: the acquisition optimiser's one-row calls) instead of through the 128-row tile path;
 *                        gp_predict_full_cov / gp_posterior_samples always take the tile path
 *   "debug_potrf_lds"    test hook, process-wide: extra dynamic LDS requested with every diagonal-tile launch (a refused launch beyond
 *                        ~9 KB: what the launch checks are tested with)
 *   "profile_min_tiles"  see gp_profile
 * The number of CUs kept free of the trailing update for the look-ahead chain is fixed per process
 * (environment GPHIP_RESERVE_CUS, default 32; "reserve_cus" only checks the value). */
int gp_set_option(gp_t *gp, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* GPHIP_H */

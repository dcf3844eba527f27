/*
 * gpz_hip.h — C ABI of libgpz_hip.so: the MI355X (gfx950) implementation of GPz's
 * marginal-likelihood objective/gradient path.
 *
 * Every entry point replaces one reference interface (paths relative to the
 * OxfordML/GPz tree); INTEGRATION.md shows the MEX / ctypes binding for each.
 *
 *   gpz_ctx_create      the closure  f = @(params) GPz(params,model,X,Y,Psi,omega,training,validation)
 *                       GPz/train.m:40, GPz/init.m:89  (data captured once; row selection of
 *                       GPz/getPHI.m:14-22 is done here, once, instead of on every call)
 *   gpz_eval            [nlogML,grad] = GPz(theta,...)      GPz/GPz.m:1  (nargout<=2 mode, :89-261)
 *                       + globals trainRMSE/trainLL/validRMSE/validLL   GPz/GPz.m:3-7,236-259
 *   gpz_solve           [~,~,w,iSigma_w] = GPz(theta,...)   GPz/GPz.m:84-87 (nargout>2 mode)
 *   gpz_phi             [PHI,Gamma,lnBeta_i,N] = getPHI(X,Psi,theta,model,[]) GPz/getPHI.m:1
 *   gpz_predict_full    predictFull(X,theta,w,iSigma_w,model) GPz/predictDiag.m:58-74, predictCov.m:53-69
 *   gpz_predict_noisy   predictNoisy(X,Psi,...)               GPz/predictDiag.m:75-125, predictCov.m:70-132
 *   gpz_predict_missing predictMissing / predictNoisyMissing  GPz/predictDiag.m:127-297, predictCov.m:134-337
 *   gpz_prior           prior = getPrior(X,Psi,theta,model,set) GPz/getPrior.m:1
 *   gpz_inv_logdet      [Xi,logdet] = inv_logdet(X)         GPz/inv_logdet.m:1
 *   gpz_dxy             D = Dxy(X,Y)                        GPz/Dxy.m:1
 *   gpz_nan_groups      the NaN-pattern grouping loop       GPz/getPHI.m:43-54 (== GPz.m:118-129)
 *   gpz_mgpu_*          the same closure / calls on all GPUs of the node behind one synchronous call
 *                       (minFunc_2012/minFunc/minFunc.m:314 calls funObj once and waits)
 *
 * Limits: none on d, k or the number of NaN patterns for the evaluation, getPHI, predictFull, predictNoisy, getPrior and the
 * NaN grouping (the reference is generic, getPHI.m:60-110, GPz.m:133-213).  d <= 20 and k <= 8 run the instantiated,
 * register-resident kernels; wider inputs / more outputs take runtime-d kernels with the row data in LDS (k_wide.hip) and, for
 * GC/VC with missing values, a workspace-backed form of the general path (DESIGN.md section 7 gives the cost; when a row tile
 * does not fit the 160 KB of LDS - d beyond ~100 for the diagonal kinds with input noise and missing values, ~300 without - the
 * PHI build reads the row data where it uses it, and GC/VC beyond d = 142 factor Gamma_j in a device workspace).  GC/VC with
 * input noise in fp64 runs register-resident up to d = 10 and as a block elimination in f64 MFMA accumulators for
 * 10 < d <= 64 (four pairs per wave up to d = 48), the workspace form beyond.
 * gpz_predict_missing for GC/VC runs register / MFMA kernels up to d = 32 and scratch-resident kernels with 64-wide temporaries for
 * 32 < d <= 64 (correct and slow: 32 KB of scratch per thread and temporary) and, for ANY wider input, the same kernels with their
 * temporaries in a device workspace (k_pmiss_covg.hip: 3 d^2 + 2 d doubles per thread, at most 2 GB per launch; 9 rows of one
 * pattern at d = 100: 0.8 s with m = 4, 2.9 s with m = 16).  The diagonal kinds run tuned to d = 144 (pattern mask by value, d KB of
 * LDS per 64 basis pairs) and LDS-free beyond (pattern as device flags; d = 260, m = 8, 30 rows: 83 ms).  No width is refused.
 * Rows: PHI and T (2 n mp doubles) stay on the device while they fit; beyond that the evaluation streams them in row tiles (PHI built
 * twice per evaluation, +10 % at c4's shape; plain route: no Psi, no missing values on GC/VC) - gpz_ctx_route reports it, gpz_get_phi is
 * refused there.
 * dtype = f32 with d > 20 takes
 * the fp64 kernels (the fp32 pair kernels hold a d <= 20 triangle in registers); gpz_ctx_route says which route a context runs.
 *
 * Conventions (MATLAB's, so a MEX shim is pure marshalling):
 *   - all matrices are column-major double; masks are 1 byte per row (MATLAB logical);
 *     a NULL pointer plays MATLAB's [].
 *   - theta / grad use the reference packing [P(:);Gamma(:);lnAlpha(:);b(:);v(:);lnTau(:)]
 *     (GPz/GPz.m:227-231, GPz/init.m:87-97).
 *   - host pointers are never retained after a call returns; outputs are caller-allocated.
 *   - return value: 0 ok; <0 error (text via gpz_last_error()).  Numerical breakdown
 *     (non-positive-definite or non-finite SIGMA) is NOT an error: f and g come back NaN so the
 *     caller's line search backs off (minFunc/WolfeLineSearch.m:53-70, isLegal.m); this is a
 *     documented deviation — the reference would raise from svd().
 *   - no torch types, no C++ in the signatures.  The library owns its device memory
 *     (hipMalloc) and runs on the HIP stream given at context creation (NULL = the null stream).
 */
#ifndef GPZ_HIP_H
#define GPZ_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPZ_OK               0
#define GPZ_ERR_ARG         -1   /* bad argument / unsupported combination */
#define GPZ_ERR_HIP         -2   /* HIP runtime failure */
#define GPZ_ERR_ALLOC       -3
#define GPZ_ERR_COMM        -4   /* the all-reduce hook failed */
#define GPZ_ERR_UNSUPPORTED -5   /* valid in the reference, not built yet (see DESIGN.md scope table) */

#define GPZ_VERSION 3   /* 3: no limits on d, k, NaN patterns; gpz_release_cached_memory; psi_kind 3 */

typedef struct gpz_ctx gpz_ctx;

/* model.{d,k,m,method,heteroscedastic} (GPz/init.m:16-20) + placement. */
typedef struct gpz_desc {
    int32_t d;                /* input dimension */
    int32_t m;                /* number of basis functions */
    int32_t k;                /* number of outputs */
    char    method[4];        /* "GL","VL","GD","VD","GC","VC" (NUL padded) */
    int32_t heteroscedastic;  /* model.heteroscedastic */
    int32_t device;           /* HIP device ordinal */
    void   *stream;           /* hipStream_t to run on; NULL = null stream */
    int32_t rank;             /* this shard's rank (0 when unsharded) */
    int32_t world;            /* number of row shards (1 when unsharded) */
    int32_t dtype;            /* GPZ_F64 (0, default) or GPZ_F32: precision flag of the path.  f32 applies to
                               * GC/VC with input noise and no missing values (BASELINE config 5): fp32 per-(sample, basis)
                               * factorisations, and the two MFMA contractions on fp32-rounded operands (PHI'W PHI with fp64
                               * master sums, PHI*inv(SIGMA) with fp32 accumulation).  Delta, ln PHI, PHI, every sum over
                               * rows, the m x m stage, theta, f and g stay fp64; every other configuration ignores the
                               * flag and runs the fp64 path bit for bit.  Gates: 1e-4 on f, 1e-3 on g (of max|g|) */
    int32_t omega_cols;       /* columns of omega: 0 or 1 = n_tot x 1, one weight per row for every output; k = n_tot x k,
                               * per-output weights (GPz.m:48 omega(training,:); getOmega.m:19 returns (1+Y).^-2, n x k for a
                               * k-column Y).  The two RMSE statistics read omega(training) = the FIRST column (GPz.m:236,258) */
    int32_t reserved[2];
} gpz_desc;
#define GPZ_F64 0
#define GPZ_F32 1

/* In-place SUM all-reduce over ranks of `count` doubles at device pointer `buf`, ordered on
 * `stream`.  Return 0 on success.  The Python host wires this to torch.distributed (RCCL);
 * a single-process caller leaves it unset. */
typedef int (*gpz_allreduce_fn)(void *user, void *buf, size_t count, void *stream);

/* Build the evaluation context.  X is n_tot x d, Y n_tot x k, omega n_tot x 1 or n_tot x k (desc->omega_cols; NULL = ones),
 * training/validation n_tot x 1 logical (NULL = all rows / no validation), all HOST pointers,
 * column-major.  psi_kind: 0 none; 1 = n_tot x d (after fixPsi, diag kinds); 2 = d x d x n_tot cube (GC/VC);
 * 3 = n_tot x d per-dimension variances for GC/VC, meaning the diagonal cubes fixPsi.m:27-31 builds from them
 * (Psi(:,:,i) = diag(S(i,:))): the library expands them, so the caller never materialises d x d x n (6.4 GB at
 * n = 2e6, d = 20).  Results are identical to passing that cube with psi_kind 2.
 * When world>1 the arrays hold only this rank's row shard. */
int gpz_ctx_create(const gpz_desc *desc, int64_t n_tot,
                   const double *X, const double *Y,
                   const double *Psi, int32_t psi_kind,
                   const double *omega,
                   const uint8_t *training, const uint8_t *validation,
                   gpz_ctx **out);

/* gpz_ctx_create for a row shard of GC/VC data with missing values.  The NaN-pattern groups of getPHI.m:43-54 must be
 * the same on every rank (the second all-reduce carries one record block per pattern): `patterns` is the table of the
 * WHOLE data set, n_patterns x d bytes row-major, 1 = missing (isnan), in first-occurrence order; a rank may hold no
 * row of some pattern.  NULL / 0 behaves like gpz_ctx_create. */
int gpz_ctx_create_sharded(const gpz_desc *desc, int64_t n_tot, const double *X, const double *Y,
                           const double *Psi, int32_t psi_kind, const double *omega,
                           const uint8_t *training, const uint8_t *validation,
                           const uint8_t *patterns, int32_t n_patterns, gpz_ctx **out);
void gpz_ctx_destroy(gpz_ctx *ctx);
int  gpz_ctx_set_allreduce(gpz_ctx *ctx, gpz_allreduce_fn fn, void *user);

/* numel(theta) for this context's model / for a model description (init.m:65-97; -1: bad description). */
int64_t gpz_theta_len(const gpz_ctx *ctx);
int64_t gpz_theta_len_of(const gpz_desc *desc);
/* rows selected by the training / validation mask on this rank. */
int64_t gpz_n_train(const gpz_ctx *ctx);
int64_t gpz_n_valid(const gpz_ctx *ctx);

/* [f,g] = GPz(theta,...).  stats[0..3] = trainRMSE, trainLL, validRMSE, validLL (the last two
 * untouched when there is no validation mask).  diag (optional, may be NULL) receives
 * diag[0] = Cholesky info (0 ok, j>0: pivot j not positive), diag[1] = global n. */
int gpz_eval(gpz_ctx *ctx, const double *theta, double *f, double *g, double stats[4], double diag[2]);

/* [~,~,w,iSigma_w] = GPz(theta,...): w is m x k, iSigma_w m x m x k; nlogML_partial (optional)
 * is the un-normalised 1 x k vector of GPz.m:81-82. */
int gpz_solve(gpz_ctx *ctx, const double *theta, double *w, double *iSigma_w, double *nlogML_partial);

/* gpz_eval for a device-resident caller: theta_dev and g_dev are device pointers on the context's device; only f and
 * the statistics cross PCIe.  Used with the L-BFGS memory below to keep the optimiser's vectors on the GPU. */
int gpz_eval_dev(gpz_ctx *ctx, const double *theta_dev, double *f, double *g_dev, double stats[4], double diag[2]);

/* Copy PHI (n_train x m, column-major) of the last gpz_eval/gpz_solve back to the host
 * (5th output of GPz.m:1). */
int gpz_get_phi(gpz_ctx *ctx, double *PHI);

/* Which branch of inv_logdet.m:7-12 an evaluation takes.  mode 0 (default): the Cholesky inverse, and the
 * rank-truncating SVD pseudo-inverse whenever SIGMA is close enough to singular that the reference might drop
 * singular values (decided on the device from ||SIGMA||_F and ||inv||_F, or a failed pivot); 1: always the
 * SVD route; -1: never (a failed pivot then gives NaN f/g).
 * gpz_ctx_last_pinv: out[0] = 1 if the last gpz_eval/gpz_solve took the SVD route, out[1] = rank kept
 * (minimum over outputs), out[2] = largest singular value, out[3] = Jacobi sweeps. */
int gpz_ctx_set_pinv_mode(gpz_ctx *ctx, int mode);
int gpz_ctx_last_pinv(const gpz_ctx *ctx, double out[4]);

/* Per-stage GPU time (HIP events on the context's stream).  enable = 1: events around every stage (the evaluation then runs as
 * eager launches); enable = 2: around the dominant stages only (phi_build, syrk, tgemm, moments), the evaluation still replayed as
 * hipGraph segments with the events between them; 0: off.  gpz_ctx_timings copies up to `cap` accumulated stage times in ms and the
 * call counts, returns the number of stages; names are static strings. */
int gpz_ctx_enable_timing(gpz_ctx *ctx, int enable);   /* 3: level 2 plus events around every other segment ("rest") and around the
                                                          * all-reduce hooks ("exchange"): wall time of a call minus the sum of all stage times is then what the device spent BETWEEN
                                                          * segments - the launch-to-launch gaps of a replayed evaluation */
int gpz_ctx_timings(gpz_ctx *ctx, const char **names, double *ms, int64_t *calls, int cap);
int gpz_ctx_reset_timings(gpz_ctx *ctx);
/* Which kernel family this context's rows run on (e.g. dtype = GPZ_F32 with missing values or d > 20 takes the fp64 pair kernels:
 * said here instead of silently), which MFMA operand type the contractions use, and the state of the captured evaluation graph
 * ("replayed" / "eager" / "disabled").  Writes a NUL-terminated description of at most cap bytes; returns its full length. */
int gpz_ctx_route(const gpz_ctx *ctx, char *buf, int cap);

/* [PHI,~,lnBeta_i,N] = getPHI(X,Psi,theta,model,[]) on ns rows: PHI ns x m, lnBeta_i ns x k, N ns x m
 * (column-major, host; any output may be NULL).  Psi / psi_kind as in gpz_ctx_create; X may contain NaN. */
int gpz_phi(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns,
            const double *Psi, int32_t psi_kind, double *PHI, double *lnBeta_i, double *N);

/* predictFull: mu = PHI*w (muY NOT added, as in predictDiag.m:65), nu, beta_i; PHI optional. */
int gpz_predict_full(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                     const double *Xs, int64_t ns,
                     double *mu, double *nu, double *beta_i, double *PHI);

/* predictNoisy (inputs with noise Psi, no missing values): also returns gamma; mu without muY. */
int gpz_predict_noisy(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                      const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                      double *mu, double *nu, double *beta_i, double *gamma, double *PHI);

/* predictMissing / predictNoisyMissing (GPz/predictDiag.m:127-297, GPz/predictCov.m:134-337) for ONE group of
 * rows that share a NaN pattern (predict.m:45-69 forms the groups; the pattern is read from the first row,
 * predictDiag.m:3).  priors: 1 x m mixture weights (model.best.priors, train.m:59,74).  Psi: n x d (diagonal
 * kinds) / d x d x n (GC, VC) or NULL. */
int gpz_predict_missing(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                        const double *priors, const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                        double *mu, double *nu, double *beta_i, double *gamma, double *PHI);

/* ---- device-resident L-BFGS memory: minFunc's lbfgsAdd.m / lbfgsProd.m (mex/lbfgsAddC.c, mex/lbfgsProdC.c) ----
 * S and Y (p x corrections) live on the device; all vector arguments are device pointers.
 * gpz_lbfgs_add:        y = g - g_old, s = t*d; skipped (added = 0) when y's <= 1e-10        (lbfgsAdd.m:2-4)
 * gpz_lbfgs_direction:  d = -H*g with Hdiag = y's/y'y of the newest pair                      (lbfgsProd.m, minFunc.m:553-578)
 *                       (called with the g of the gpz_lbfgs_add just before - minFunc's order - it reuses the products [S Y]'g of that pass)
 * gpz_vec_stats:        out = [g.d, max|g|, sum|g|, max|d|] (NaN-propagating), the scalars of minFunc's tests
 * gpz_vec_axpy:         out = x + t*d */
typedef struct gpz_lbfgs gpz_lbfgs;
int gpz_lbfgs_create(int64_t p, int32_t corrections, int32_t device, void *stream, gpz_lbfgs **out);
void gpz_lbfgs_destroy(gpz_lbfgs *h);
int gpz_lbfgs_add(gpz_lbfgs *h, const double *g_dev, const double *g_old_dev, double t, const double *d_dev, int32_t *added);
int gpz_lbfgs_direction(gpz_lbfgs *h, const double *g_dev, double *d_dev);
const char *gpz_lbfgs_last_error(void);
int gpz_vec_stats(const double *g_dev, const double *d_dev, int64_t p, int32_t device, void *stream, double out[4]);
int gpz_vec_axpy(double *out_dev, const double *x_dev, double t, const double *d_dev, int64_t p, int32_t device, void *stream);

/* prior = getPrior(X,Psi,theta,model,[]): mixture weights (1 x m) of the normalised basis densities;
 * iterations (optional) receives the number of fixed-point iterations used (<= 100). */
int gpz_prior(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns,
              const double *Psi, int32_t psi_kind, double *prior, int32_t *iterations);

/* [Xi,logdet] = inv_logdet(X) for a symmetric m x m matrix (GPz/inv_logdet.m:1-15).  A comfortably
 * positive-definite X goes through the Cholesky inverse; a numerically singular or indefinite one
 * through a Jacobi SVD with the reference's truncation (singular values <= m*eps(max s) dropped from
 * both Xi and logdet).  info (optional): number of singular values dropped (0 = none), -1 = X is not
 * finite (Xi, logdet are NaN then; MATLAB's svd raises). */
int gpz_inv_logdet(const double *A, int32_t m, int32_t device, double *Xi, double *logdet, int32_t *info);

/* D = | |x|^2 + |y|^2 - 2 x y' |, X nx x d, Y ny x d, D nx x ny. */
int gpz_dxy(const double *X, int64_t nx, const double *Y, int64_t ny, int32_t d, int32_t device, double *D);

/* Group id per row = rank (by first occurrence) of the row's NaN pattern; returns the number of
 * groups in *n_groups.  Bit-exact with the greedy loop of getPHI.m:43-54. */
int gpz_nan_groups(const double *X, int64_t n, int32_t d, int32_t device, int32_t *group_id, int32_t *n_groups);

/* ---- all GPUs of the node behind ONE synchronous call (SURVEY.md 8b "threading", 8e) -------------------------------
 * The reference calls [f,g] = funObj(x) from one MATLAB process and blocks on it (minFunc/minFunc.m:314,
 * WolfeLineSearch.m:34,114,194; closure at GPz/train.m:40).  gpz_mgpu_create takes the same arguments as gpz_ctx_create
 * (the whole, unsharded data set), splits the training-selected rows and the validation rows into contiguous balanced
 * blocks, one per device, and keeps one context per device; gpz_mgpu_eval / gpz_mgpu_solve then have the semantics of
 * gpz_eval / gpz_solve.  Inside, one host thread per device drives that device's stream and the two all-reduces of an
 * evaluation are RCCL calls (ncclCommInitAll communicators) on the library's device buffers: only m x m and
 * m x (d^2+d) partials cross xGMI.  desc->device / stream / rank / world are ignored.
 *   n_gpus   <= 0: every device of the node (hipGetDeviceCount)
 *   devices  NULL: 0 .. n_gpus-1 (loopback: all 0)
 *   reducer  GPZ_REDUCER_RCCL, or GPZ_REDUCER_LOOPBACK: an in-library rank-ordered reducer for shards that share one
 *            device (RCCL refuses duplicate devices) - the way the sharded path is tested on single-GPU machines. */
#define GPZ_REDUCER_RCCL     0
#define GPZ_REDUCER_LOOPBACK 1
typedef struct gpz_mgpu gpz_mgpu;
int gpz_mgpu_create(const gpz_desc *desc, int32_t n_gpus, const int32_t *devices, int32_t reducer, int64_t n_tot,
                    const double *X, const double *Y, const double *Psi, int32_t psi_kind, const double *omega,
                    const uint8_t *training, const uint8_t *validation, gpz_mgpu **out);
void gpz_mgpu_destroy(gpz_mgpu *h);
int gpz_mgpu_eval(gpz_mgpu *h, const double *theta, double *f, double *g, double stats[4], double diag[2]);
int gpz_mgpu_solve(gpz_mgpu *h, const double *theta, double *w, double *iSigma_w, double *nlogML_partial);
int32_t gpz_mgpu_size(const gpz_mgpu *h);
/* Failure of one rank inside a call (HIP error, allocation failure, a failed exchange): with GPZ_REDUCER_RCCL the other ranks would
 * wait inside ncclAllReduce for ever, so the failing rank aborts every communicator of the handle (ncclCommAbort); the call returns
 * that rank's error and the handle is DEAD: gpz_mgpu_alive returns 0 and every later gpz_mgpu_eval / _solve returns GPZ_ERR_COMM until
 * the handle is destroyed and re-created.  (The loopback reducer releases its barrier instead and stays usable.)
 * gpz_mgpu_debug_fail_at: test hook - the next call fails on `rank` at its exchange point `exchange` (1 or 2), once. */
int32_t gpz_mgpu_alive(const gpz_mgpu *h);
int gpz_mgpu_debug_fail_at(gpz_mgpu *h, int32_t rank, int32_t exchange);
int64_t gpz_mgpu_theta_len(const gpz_mgpu *h);
/* the context of one rank, for gpz_n_train / gpz_ctx_enable_timing / gpz_ctx_timings / gpz_ctx_set_pinv_mode (apply
 * settings to every rank); owned by the handle - never destroy it. */
gpz_ctx *gpz_mgpu_ctx(gpz_mgpu *h, int32_t rank);
int gpz_device_count(void);
/* Prediction of ONE NaN-pattern group (predict.m:60-69) over several GPUs: the rows are independent, so contiguous row blocks go
 * to the devices, each through the single-device entry its content selects — gpz_predict_full / _noisy / _missing, the choice
 * predictDiag.m:39-55 makes from X (pattern of the first row) and Psi.  Arguments as in those entries (priors and gamma may be
 * NULL when the group has neither missing values nor input noise: gamma is then not written / written as 0); results equal the
 * single-device call row for row.  n_gpus <= 0: every device; devices NULL: 0 .. n_gpus-1 (cyclic when n_gpus exceeds the node:
 * several blocks per device, which is how single-GPU machines exercise the path). */
int gpz_mgpu_predict(const gpz_desc *desc, int32_t n_gpus, const int32_t *devices, const double *theta, const double *w,
                     const double *iSigma_w, const double *priors, const double *Xs, int64_t ns, const double *Psi,
                     int32_t psi_kind, double *mu, double *nu, double *beta_i, double *gamma, double *PHI);

/* ---- one rank per PROCESS (torchrun / mpirun launchers): RCCL inside the library instead of a caller-supplied hook.
 * Rank 0 calls gpz_rccl_unique_id and ships the 128 bytes to the other ranks by any out-of-band means; every rank then
 * calls gpz_ctx_init_rccl on its sharded context (collective: returns when all `world` ranks have called).  The
 * communicator is destroyed with the context.  gpz_rccl_origin: which RCCL the library bound to (dlopen at first use:
 * one the process already carries, else librccl.so.1 from the library path). */
#define GPZ_RCCL_ID_BYTES 128
int gpz_rccl_unique_id(void *id128);
int gpz_ctx_init_rccl(gpz_ctx *ctx, const void *id128, int32_t rank, int32_t world, int32_t device);
const char *gpz_rccl_origin(void);
/* What the communicator behind a context's (a rank's) all-reduce reports about itself, so that a multi-GPU run can prove N ranks on N
 * devices from its own output: info[0] = ncclCommCount, info[1] = ncclCommUserRank, info[2] = ncclCommCuDevice (-1 each when there is
 * no in-library RCCL communicator: single rank, loopback reducer, a caller-supplied hook), info[3] = the HIP device ordinal of the
 * context; bus_id (optional, cap bytes) = that device's PCI bus id.  No reference counterpart. */
int gpz_ctx_comm_info(const gpz_ctx *ctx, int32_t info[4], char *bus_id, int32_t cap);
int gpz_mgpu_comm_info(const gpz_mgpu *h, int32_t rank, int32_t info[4], char *bus_id, int32_t cap);

/* Text of the calling thread's last failure - or of a NEWER failure on another thread (work that failed on one of the library's
 * worker threads); "" when nothing has failed.  The return code of the call is the authority, this is its text.  Valid until the
 * thread's next library call. */
const char *gpz_last_error(void);
int gpz_version(void);

/* Device buffers released by contexts and stand-alone calls are kept (up to 4 GiB per device) for the next call instead of
 * being returned to the runtime one hipFree at a time, and gpz_predict_missing (GC/VC) keeps the last model's covariance and
 * basis-pair tables on the device for the next NaN-pattern group; this hands all of that back.  No reference counterpart. */
void gpz_release_cached_memory(void);

/* Test hook: the kth device allocation from now on (kth >= 1; 0 disarms) reports out-of-memory once, so the release-and-retry
 * path of the allocator can be exercised without exhausting 288 GB.  Process-wide.  No reference counterpart. */
void gpz_debug_fail_alloc(int64_t kth);

#ifdef __cplusplus
}
#endif
#endif /* GPZ_HIP_H */

// gpz_mex.cpp — MEX gateway between MATLAB and libgpz_hip.so (pure marshalling, no arithmetic).
//
// Build on a machine that has MATLAB (not available in the build image; tests/test_mex_gateway.py compiles this file
// against the declarations-only header tests/stubs/mex.h and runs it against a stand-in MEX runtime on the GPU box):
//     mex -R2017b gpz_mex.cpp -I../include -L../gpz_amd/lib -lgpz_hip
// Same gateway convention as the reference's own MEX files (minFunc_2012/minFunc/mex/lbfgsProdC.c:7); input types are
// checked and reported with mexErrMsg* like lbfgsProdC.c:24-25; outputs come from mxCreate* (lbfgsProdC.c:43); prhs is
// never written (the reference's lbfgsAddC.c:30-33 does write into its inputs, knowingly).
//
//   [f, g, stats] = gpz_mex('eval',  theta, model, X, Y, Psi, omega, training, validation)   GPz.m nargout<=2
//   [w, iSigma_w, part] = gpz_mex('solve', theta, model, X, Y, Psi, omega, training, validation)   GPz.m:84-87
//        The closure's data (train.m:40) lives on the GPUs between calls.  Every call hands the closure's arguments
//        over again — MATLAB passes shared-data copies, so this costs nothing — and the gateway compares them with
//        what the live context was built from: model fields, array sizes, data pointers, and the CONTENT of the arrays:
//        Y, omega, training and validation are hashed completely on every call (MATLAB edits an unshared variable in
//        place and keeps its pointer — `training(bad) = false` between two train() calls must not evaluate on stale
//        device data; 18 MB at n = 1e6, ~2 ms).  X and Psi (the big ones: 80 MB at config 4, 6.4 GB as cubes at config 5)
//        are compared on EVERY call by pointer, size and 256 strided samples, and by a complete 64-bit content hash
//        (computed when the context is built, on all host cores) that is re-checked every K-th call: an in-place edit
//        of X or Psi is caught at the latest K-1 evaluations later, rebuilds the context and counts in 'verify_info'.
//        K = model.verify_every when given (1 = every call: exact); otherwise K is chosen from measured times so that
//        the hashing costs at most 2 % of the evaluations it guards (K = ceil(hash seconds / (0.02 x evaluation
//        seconds)), 1 .. 64; c4: 80 MB hash ~2 ms on 16 cores against 55 ms => K = 2).  Anything different rebuilds
//        the context.  omega is n x 1 or n x model.k (GPz.m:48 `omega(training,:)`, getOmega.m:19).  model.n_gpus (optional) = number of GPUs, default all of the node; the rows are sharded across them
//        and reduced with RCCL inside the library (gpz_mgpu_*); model.reducer = 'loopback' (optional) puts model.n_gpus
//        shards on ONE device with the library's own reducer (single-GPU hosts, tests).
//        model.dtype = 'f32' (optional) selects the fp32 per-pair factorisations of GC/VC with input noise.
//   PHI = gpz_mex('phi')                                               5th output of GPz.m:1 (after eval / solve)
//   [PHI, lnBeta_i, N] = gpz_mex('getphi', model, theta, X, Psi)       getPHI.m:1 (rows already selected)
//   [mu,nu,beta_i,gamma,PHI] = gpz_mex('predict', model, theta, w, iSigma_w, priors, X, Psi)
//                                                  one NaN-pattern group of predict.m:60-69: predictFull / predictNoisy /
//                                                  predictMissing / predictNoisyMissing by what X and Psi contain
//   prior = gpz_mex('prior', model, theta, X, Psi)                     getPrior.m:1
//   [Xi, logdet] = gpz_mex('inv_logdet', A)                            inv_logdet.m:1
//   D = gpz_mex('dxy', X, Y)                                           Dxy.m:1
//   gpz_mex('pinv_mode', mode)                                         branch of inv_logdet.m:7-12 (0 auto, 1 always, -1 never)
//   n = gpz_mex('gpus')                                                GPUs the live context runs on (0: none)
//   n = gpz_mex('builds')                                              how many times a device context was (re)built so far
//   v = gpz_mex('verify_info')                                         [K, seconds per complete hash of X and Psi, seconds of the last
//                                                                       evaluation, rebuilds caused by an in-place edit of X / Psi]
//   gpz_mex('reset')                                                   drop the device context (clear global / new data)
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>
#include <math.h>
#include <unistd.h>
#include <pthread.h>
#include "mex.h"
#include "gpz_hip.h"

#define NARR 6                        /* X, Y, Psi, omega, training, validation */
#define NSAMP 256

typedef struct {
    int32_t d, m, k, hetero, dtype, n_gpus, reducer;
    char method[4];
    const void *ptr[NARR];
    size_t bytes[NARR];
    mwSize rows[NARR];
    uint64_t sum[NARR];
} closure_key;

static gpz_mgpu *g_mg = NULL;
static closure_key g_key;
static int g_m = 0, g_k = 0, g_pinv = 0, g_locked = 0, g_builds = 0;
/* the exact identity of X and Psi: complete content hashes taken at build time, re-checked every g_every-th call */
static uint64_t g_full[2];
static int g_every = 1, g_since = 0, g_stale = 0;
static double g_hash_s = 0.0, g_eval_s = 0.0;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void cleanup(void) {
    if (g_mg) { gpz_mgpu_destroy(g_mg); g_mg = NULL; }
    g_eval_s = 0.0;   /* times of another closure say nothing about the next */
}
static const double *opt(const mxArray *a) { return (a && !mxIsEmpty(a)) ? mxGetPr(a) : NULL; }
static const uint8_t *optmask(const mxArray *a) {
    if (!a || mxIsEmpty(a)) return NULL;
    if (!mxIsLogical(a)) mexErrMsgIdAndTxt("gpz:type", "training / validation must be logical");
    return (const uint8_t *)mxGetLogicals(a);
}
static void need_double(const mxArray *a, const char *what, int allow_empty) {
    if (!a || (mxIsEmpty(a) && allow_empty)) return;
    if (!mxIsDouble(a) || mxIsComplex(a) || mxIsEmpty(a)) mexErrMsgIdAndTxt("gpz:type", "%s must be a real double array", what);
}
static double field(const mxArray *s, const char *name) {
    const mxArray *f = mxIsStruct(s) ? mxGetField(s, 0, name) : NULL;
    if (!f) mexErrMsgIdAndTxt("gpz:model", "model.%s missing", name);
    return mxGetScalar(f);
}

static int32_t reducer_of(const mxArray *model) {
    const mxArray *r = mxIsStruct(model) ? mxGetField(model, 0, "reducer") : NULL;
    char buf[16] = "";
    if (!r || mxIsEmpty(r)) return GPZ_REDUCER_RCCL;
    if (mxGetString(r, buf, sizeof buf)) mexErrMsgIdAndTxt("gpz:model", "model.reducer must be 'rccl' or 'loopback'");
    if (!strcmp(buf, "loopback")) return GPZ_REDUCER_LOOPBACK;
    if (!strcmp(buf, "rccl")) return GPZ_REDUCER_RCCL;
    mexErrMsgIdAndTxt("gpz:model", "model.reducer must be 'rccl' or 'loopback'");
    return GPZ_REDUCER_RCCL;
}
static gpz_desc desc_of(const mxArray *model, int32_t *n_gpus) {
    gpz_desc d;
    memset(&d, 0, sizeof d);
    d.d = (int32_t)field(model, "d"); d.m = (int32_t)field(model, "m"); d.k = (int32_t)field(model, "k");
    d.heteroscedastic = (int32_t)field(model, "heteroscedastic");
    const mxArray *me = mxGetField(model, 0, "method");
    if (!me || mxGetString(me, d.method, sizeof d.method)) mexErrMsgIdAndTxt("gpz:model", "model.method missing");
    d.world = 1;
    const mxArray *dt = mxGetField(model, 0, "dtype");
    char buf[8] = "";
    if (dt && !mxIsEmpty(dt)) {
        if (mxGetString(dt, buf, sizeof buf) || (strcmp(buf, "f32") && strcmp(buf, "f64")))
            mexErrMsgIdAndTxt("gpz:model", "model.dtype must be 'f64' or 'f32'");
        if (!strcmp(buf, "f32")) d.dtype = GPZ_F32;
    }
    if (gpz_theta_len_of(&d) < 0) mexErrMsgIdAndTxt("gpz:model", "model.d, m, k must be >= 1 and model.method one of GL VL GD VD GC VC");
    if (n_gpus) {
        const mxArray *ng = mxGetField(model, 0, "n_gpus");
        *n_gpus = (ng && !mxIsEmpty(ng)) ? (int32_t)mxGetScalar(ng) : 0;       /* 0: every GPU of the node */
    }
    return d;
}
/* fixPsi.m layouts: [] / n x d / d x d x n; an n x d array handed to GC/VC is the per-dimension variances fixPsi.m:27-31 would
 * turn into diagonal cubes - psi_kind 3, expanded by the library */
static int psi_kind_of(const mxArray *Psi, const gpz_desc *d) {
    if (!Psi || mxIsEmpty(Psi)) return 0;
    if (mxGetNumberOfDimensions(Psi) == 3) return 2;
    return d->method[1] == 'C' ? 3 : 1;
}
static int has_nan(const mxArray *X) {
    const double *x = mxGetPr(X);
    for (size_t i = 0, n = mxGetNumberOfElements(X); i < n; ++i)
        if (mxIsNaN(x[i])) return 1;
    return 0;
}
#define CHECK(call, id) do { if (call) mexErrMsgIdAndTxt(id, "%s", gpz_last_error()); } while (0)

/* 64-bit content hash, four independent lanes of 8-byte words (memory-bound: ~2 ms for the 18 MB of n-sized arrays at n = 1e6). */
static uint64_t hash_bytes(const unsigned char *p, size_t nbytes) {
    uint64_t h[4] = {1469598103934665603ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
    const size_t nw = nbytes / 8;
    size_t i = 0;
    for (; i + 4 <= nw; i += 4)
        for (int l = 0; l < 4; ++l) {
            uint64_t w;
            memcpy(&w, p + (i + l) * 8, 8);
            h[l] = (h[l] ^ w) * 0x100000001B3ull;
            h[l] ^= h[l] >> 29;
        }
    for (; i < nw; ++i) {
        uint64_t w;
        memcpy(&w, p + i * 8, 8);
        h[0] = (h[0] ^ w) * 0x100000001B3ull;
        h[0] ^= h[0] >> 29;
    }
    for (size_t b = nw * 8; b < nbytes; ++b) h[1] = (h[1] ^ p[b]) * 0x100000001B3ull;
    return (h[0] * 31 + h[1]) * 31 + (h[2] * 31 + h[3]) + nbytes;
}

/* The complete hash of a big array on all host cores: 4 MB chunks dealt to threads, the chunk hashes combined in chunk order (the
 * result does not depend on the number of threads). */
#define HCHUNK ((size_t)4 << 20)
typedef struct { const unsigned char *p; size_t nbytes, nchunk; uint64_t *out; int t, nt; } hash_job;
static void *hash_worker(void *arg) {
    hash_job *j = (hash_job *)arg;
    for (size_t c = (size_t)j->t; c < j->nchunk; c += (size_t)j->nt) {
        const size_t lo = c * HCHUNK, len = j->nbytes - lo < HCHUNK ? j->nbytes - lo : HCHUNK;
        j->out[c] = hash_bytes(j->p + lo, len);
    }
    return NULL;
}
static uint64_t hash_bytes_all_cores(const unsigned char *p, size_t nbytes) {
    if (nbytes <= HCHUNK) return hash_bytes(p, nbytes);
    const size_t nchunk = (nbytes + HCHUNK - 1) / HCHUNK;
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = nc < 1 ? 1 : (nc > 16 ? 16 : (int)nc);
    if ((size_t)nt > nchunk) nt = (int)nchunk;
    uint64_t *out = (uint64_t *)calloc(nchunk, sizeof(uint64_t));
    if (!out) mexErrMsgIdAndTxt("gpz:alloc", "out of memory");
    pthread_t th[16];
    hash_job job[16];
    int started = 0;
    for (int t = 0; t < nt; ++t) {
        job[t].p = p; job[t].nbytes = nbytes; job[t].nchunk = nchunk; job[t].out = out; job[t].t = t; job[t].nt = nt;
        if (t == nt - 1 || pthread_create(&th[t], NULL, hash_worker, &job[t])) {   /* the last share (and any a thread could not be had for) here */
            for (int u = t; u < nt; ++u) { job[u] = job[t]; job[u].t = u; hash_worker(&job[u]); }
            break;
        }
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    const uint64_t h = hash_bytes((const unsigned char *)out, nchunk * sizeof(uint64_t)) + nbytes;
    free(out);
    return h;
}
static void full_hashes(const mxArray *X, const mxArray *Psi, uint64_t out[2]) {
    const mxArray *a[2] = {X, Psi};
    for (int q = 0; q < 2; ++q)
        out[q] = (a[q] && !mxIsEmpty(a[q]))
                     ? hash_bytes_all_cores((const unsigned char *)mxGetData(a[q]), mxGetNumberOfElements(a[q]) * mxGetElementSize(a[q])) : 0;
}
/* K of the header comment: model.verify_every, or from the measured hash and evaluation times (2 % budget) */
static int verify_every_of(const mxArray *model) {
    const mxArray *v = mxIsStruct(model) ? mxGetField(model, 0, "verify_every") : NULL;
    if (v && !mxIsEmpty(v)) {
        const double k = mxGetScalar(v);
        if (!(k >= 1.0) || k != floor(k)) mexErrMsgIdAndTxt("gpz:model", "model.verify_every must be an integer >= 1");
        return k > 1e6 ? 1000000 : (int)k;
    }
    if (g_eval_s <= 0.0) return 1;                 /* nothing timed yet: check */
    const double k = ceil(g_hash_s / (0.02 * g_eval_s));
    return k < 1.0 ? 1 : (k > 64.0 ? 64 : (int)k);
}

/* What the live context was built from.  Bitwise: NaN payloads compare like any other bits.  q: 0 X, 1 Y, 2 Psi, 3 omega,
 * 4 training, 5 validation — X and Psi are sampled, the n-sized arrays are hashed completely (an in-place edit of ONE mask
 * element must rebuild the context). */
static void key_of(const mxArray *model, const mxArray *const *arr, closure_key *key) {
    memset(key, 0, sizeof *key);
    gpz_desc d = desc_of(model, &key->n_gpus);
    key->reducer = reducer_of(model);
    key->d = d.d; key->m = d.m; key->k = d.k; key->hetero = d.heteroscedastic; key->dtype = d.dtype;
    memcpy(key->method, d.method, sizeof key->method);
    for (int q = 0; q < NARR; ++q) {
        const mxArray *a = arr[q];
        if (!a || mxIsEmpty(a)) continue;
        const size_t n = mxGetNumberOfElements(a), es = mxGetElementSize(a);
        const unsigned char *p = (const unsigned char *)mxGetData(a);
        key->ptr[q] = p;
        key->bytes[q] = n * es;
        key->rows[q] = mxGetM(a);
        if (q != 0 && q != 2) { key->sum[q] = hash_bytes(p, n * es); continue; }
        const size_t step = n > NSAMP ? n / NSAMP : 1;
        uint64_t h = 1469598103934665603ull;
        for (size_t e = 0; e < n; e += step)
            for (size_t b = 0; b < es; ++b) { h ^= p[e * es + b]; h *= 1099511628211ull; }
        for (size_t b = 0; b < es; ++b) { h ^= p[(n - 1) * es + b]; h *= 1099511628211ull; }
        key->sum[q] = h;
    }
}

/* (Re)build the device context when the closure's arguments differ from the live one's.  args: model, X, Y, Psi, omega,
 * training, validation  (GPz.m:1's own argument order after theta). */
static void need_rows(const gpz_desc *d, const mxArray *X, const mxArray *Psi);
static void ensure_context(const mxArray *const *args) {
    const mxArray *model = args[0];
    closure_key key;
    key_of(model, args + 1, &key);
    // a handle a failed rank has left dead (GPZ_ERR_COMM with the RCCL reducer) is rebuilt like a changed closure: a MATLAB caller has no
    // other way to get rid of it than gpz_mex('reset')
    const mxArray *X = args[1], *Y = args[2], *Psi = args[3], *om = args[4];
    uint64_t full[2];
    int have_full = 0;
    if (g_mg && gpz_mgpu_alive(g_mg) && !memcmp(&key, &g_key, sizeof key)) {
        g_every = verify_every_of(model);
        if (++g_since < g_every) return;
        g_since = 0;
        const double t0 = now_s();
        full_hashes(X, Psi, full);
        g_hash_s = now_s() - t0;
        have_full = 1;
        if (full[0] == g_full[0] && full[1] == g_full[1]) return;
        ++g_stale;                                 /* X or Psi was edited in place where the samples do not look */
    }
    cleanup();
    need_double(X, "X", 0); need_double(Y, "Y", 0); need_double(Psi, "Psi", 1); need_double(om, "omega", 1);
    int32_t n_gpus = 0;
    gpz_desc d = desc_of(model, &n_gpus);
    need_rows(&d, X, Psi);   /* the library reads n*d (or d*d*n) doubles of Psi: a smaller array must not reach it */
    const mwSize n = mxGetM(X);
    if (mxGetN(X) != (mwSize)d.d || mxGetM(Y) != n || mxGetN(Y) != (mwSize)d.k)
        mexErrMsgIdAndTxt("gpz:size", "X must be n x model.d and Y n x model.k");
    for (int q = 5; q <= 6; ++q)
        if (args[q] && !mxIsEmpty(args[q]) && mxGetNumberOfElements(args[q]) < n)
            mexErrMsgIdAndTxt("gpz:size", "training / validation must have one entry per row of X");
    if (om && !mxIsEmpty(om)) {   /* omega(training,:) of GPz.m:48: n x 1, or n x k (getOmega.m:19 on a k-column Y) */
        if (mxGetM(om) == n && mxGetN(om) == (mwSize)d.k && d.k > 1) d.omega_cols = d.k;
        else if (mxGetNumberOfElements(om) != n) mexErrMsgIdAndTxt("gpz:size", "omega must be n x 1 or n x model.k");
    }
    if (gpz_mgpu_create(&d, n_gpus, NULL, reducer_of(model), (int64_t)n, mxGetPr(X), mxGetPr(Y), opt(Psi), psi_kind_of(Psi, &d),
                        opt(om), optmask(args[5]), optmask(args[6]), &g_mg))
        mexErrMsgIdAndTxt("gpz:create", "%s", gpz_last_error());
    g_key = key;
    ++g_builds;
    if (!have_full) { const double t0 = now_s(); full_hashes(X, Psi, full); g_hash_s = now_s() - t0; }
    g_full[0] = full[0]; g_full[1] = full[1];
    g_since = 0;
    g_every = verify_every_of(model);
    g_m = d.m; g_k = d.k;
    if (g_pinv)
        for (int32_t r = 0; r < gpz_mgpu_size(g_mg); ++r) (void)gpz_ctx_set_pinv_mode(gpz_mgpu_ctx(g_mg, r), g_pinv);
    if (!g_locked) { mexLock(); mexAtExit(cleanup); g_locked = 1; }
}

static const double *theta_of(const mxArray *th) {
    const mwSize p = (mwSize)gpz_mgpu_theta_len(g_mg);
    if (!mxIsDouble(th) || mxIsComplex(th) || mxGetNumberOfElements(th) != p)
        mexErrMsgIdAndTxt("gpz:theta", "theta must be a real double vector of %d elements", (int)p);
    return mxGetPr(th);
}

/* Stand-alone entries: theta / w / iSigma_w / priors / X / Psi must have the sizes the model implies — a short theta would be an
 * out-of-bounds host read inside the library. */
static void need_numel(const mxArray *a, size_t want, const char *what) {
    if (!a || !mxIsDouble(a) || mxIsComplex(a) || mxGetNumberOfElements(a) != want)
        mexErrMsgIdAndTxt("gpz:size", "%s must be a real double array of %d elements", what, (int)want);
}
static void need_theta(const gpz_desc *d, const mxArray *th) {
    const int64_t p = gpz_theta_len_of(d);
    if (p < 0) mexErrMsgIdAndTxt("gpz:model", "model.d / m / k / method do not describe a GPz model");
    if (!th || !mxIsDouble(th) || mxIsComplex(th) || mxGetNumberOfElements(th) != (size_t)p)
        mexErrMsgIdAndTxt("gpz:theta", "theta must be a real double vector of %d elements", (int)p);
}
static void need_rows(const gpz_desc *d, const mxArray *X, const mxArray *Psi) {
    need_double(X, "X", 0); need_double(Psi, "Psi", 1);
    if (mxGetN(X) != (mwSize)d->d) mexErrMsgIdAndTxt("gpz:size", "X must be n x model.d");
    const int kind = psi_kind_of(Psi, d);
    const size_t ns = mxGetM(X), dd = (size_t)d->d;
    if ((kind == 1 || kind == 3) && (mxGetM(Psi) != ns || mxGetN(Psi) != dd)) mexErrMsgIdAndTxt("gpz:size", "Psi must be n x model.d (fixPsi.m:42-53)");
    if (kind == 2) {
        const mwSize *dm = mxGetDimensions(Psi);
        if (dm[0] != dd || dm[1] != dd || dm[2] != ns) mexErrMsgIdAndTxt("gpz:size", "Psi must be d x d x n (fixPsi.m:22-38)");
    }
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[16];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("gpz:usage", "first argument: command");
    if (!strcmp(cmd, "reset")) { cleanup(); gpz_release_cached_memory(); return; }   /* also the buffers the library keeps between calls */
    if (!strcmp(cmd, "builds")) { plhs[0] = mxCreateDoubleScalar((double)g_builds); return; }
    if (!strcmp(cmd, "verify_info")) {
        plhs[0] = mxCreateDoubleMatrix(1, 4, mxREAL);
        double *o = mxGetPr(plhs[0]);
        o[0] = (double)g_every; o[1] = g_hash_s; o[2] = g_eval_s; o[3] = (double)g_stale;
        return;
    }
    if (!strcmp(cmd, "gpus")) { plhs[0] = mxCreateDoubleScalar(g_mg ? (double)gpz_mgpu_size(g_mg) : 0.0); return; }
    if (!strcmp(cmd, "comm")) {   /* what each rank's communicator reports about itself: [ncclCommCount ncclCommUserRank ncclCommCuDevice hipDevice] per row */
        const int n = g_mg ? (int)gpz_mgpu_size(g_mg) : 0;
        plhs[0] = mxCreateDoubleMatrix((mwSize)n, 4, mxREAL);
        double *o = mxGetPr(plhs[0]);
        for (int r = 0; r < n; ++r) {
            int32_t info[4];
            if (gpz_mgpu_comm_info(g_mg, r, info, NULL, 0)) mexErrMsgIdAndTxt("gpz:lib", "%s", gpz_last_error());
            for (int q = 0; q < 4; ++q) o[r + (size_t)q * n] = (double)info[q];
        }
        return;
    }
    /* ---- stand-alone entries (no context) ---- */
    if (!strcmp(cmd, "getphi")) {
        if (nrhs != 5) mexErrMsgIdAndTxt("gpz:usage", "getphi needs model,theta,X,Psi");
        gpz_desc d = desc_of(prhs[1], NULL);
        need_theta(&d, prhs[2]); need_rows(&d, prhs[3], prhs[4]);
        const mwSize ns = mxGetM(prhs[3]);
        plhs[0] = mxCreateDoubleMatrix(ns, d.m, mxREAL);
        mxArray *lb = mxCreateDoubleMatrix(ns, d.k, mxREAL), *N = nlhs > 2 ? mxCreateDoubleMatrix(ns, d.m, mxREAL) : NULL;
        CHECK(gpz_phi(&d, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)ns, opt(prhs[4]), psi_kind_of(prhs[4], &d),
                      mxGetPr(plhs[0]), mxGetPr(lb), N ? mxGetPr(N) : NULL), "gpz:getphi");
        if (nlhs > 1) plhs[1] = lb; else mxDestroyArray(lb);
        if (nlhs > 2) plhs[2] = N;
        return;
    }
    if (!strcmp(cmd, "predict")) {
        if (nrhs != 8) mexErrMsgIdAndTxt("gpz:usage", "predict needs model,theta,w,iSigma_w,priors,X,Psi");
        int32_t n_gpus = 0;
        gpz_desc d = desc_of(prhs[1], &n_gpus);
        const mxArray *X = prhs[6], *Psi = prhs[7];
        need_theta(&d, prhs[2]); need_rows(&d, X, Psi);
        need_numel(prhs[3], (size_t)d.m * d.k, "w");
        need_numel(prhs[4], (size_t)d.m * d.m * d.k, "iSigma_w");
        if (prhs[5] && !mxIsEmpty(prhs[5])) need_numel(prhs[5], (size_t)d.m, "priors");
        const mwSize ns = mxGetM(X);
        mxArray *o[5];
        for (int q = 0; q < 4; ++q) o[q] = mxCreateDoubleMatrix(ns, d.k, mxREAL);   /* mu nu beta_i gamma (gamma = 0 for predictFull) */
        o[4] = mxCreateDoubleMatrix(ns, d.m, mxREAL);
        const double *th = mxGetPr(prhs[2]), *w = mxGetPr(prhs[3]), *iS = mxGetPr(prhs[4]);
        /* the rows of a group are independent: contiguous blocks over model.n_gpus GPUs (default all), each through the entry
         * its content selects (predictFull / predictNoisy / predictMissing / predictNoisyMissing, predictDiag.m:39-55) */
        CHECK(gpz_mgpu_predict(&d, n_gpus, NULL, th, w, iS, opt(prhs[5]), mxGetPr(X), (int64_t)ns, opt(Psi), psi_kind_of(Psi, &d),
                               mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]), mxGetPr(o[3]), mxGetPr(o[4])), "gpz:predict");
        for (int q = 0; q < 5; ++q)
            if (q < nlhs || q == 0) plhs[q] = o[q]; else mxDestroyArray(o[q]);
        return;
    }
    if (!strcmp(cmd, "prior")) {
        if (nrhs != 5) mexErrMsgIdAndTxt("gpz:usage", "prior needs model,theta,X,Psi");
        gpz_desc d = desc_of(prhs[1], NULL);
        need_theta(&d, prhs[2]); need_rows(&d, prhs[3], prhs[4]);
        plhs[0] = mxCreateDoubleMatrix(1, d.m, mxREAL);
        CHECK(gpz_prior(&d, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)mxGetM(prhs[3]), opt(prhs[4]), psi_kind_of(prhs[4], &d),
                        mxGetPr(plhs[0]), NULL), "gpz:prior");
        return;
    }
    if (!strcmp(cmd, "inv_logdet")) {
        if (nrhs != 2) mexErrMsgIdAndTxt("gpz:usage", "inv_logdet needs a matrix");
        need_double(prhs[1], "X", 0);
        const mwSize m = mxGetM(prhs[1]);
        if (mxGetN(prhs[1]) != m) mexErrMsgIdAndTxt("gpz:size", "inv_logdet: the matrix must be square");
        double ld = 0.0;
        plhs[0] = mxCreateDoubleMatrix(m, m, mxREAL);
        CHECK(gpz_inv_logdet(mxGetPr(prhs[1]), (int32_t)m, 0, mxGetPr(plhs[0]), &ld, NULL), "gpz:inv_logdet");
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(ld);
        return;
    }
    if (!strcmp(cmd, "dxy")) {
        if (nrhs != 3) mexErrMsgIdAndTxt("gpz:usage", "dxy needs X,Y");
        need_double(prhs[1], "X", 0); need_double(prhs[2], "Y", 0);
        if (mxGetN(prhs[1]) != mxGetN(prhs[2])) mexErrMsgIdAndTxt("gpz:size", "dxy: X and Y need the same number of columns");
        plhs[0] = mxCreateDoubleMatrix(mxGetM(prhs[1]), mxGetM(prhs[2]), mxREAL);
        CHECK(gpz_dxy(mxGetPr(prhs[1]), (int64_t)mxGetM(prhs[1]), mxGetPr(prhs[2]), (int64_t)mxGetM(prhs[2]),
                      (int32_t)mxGetN(prhs[1]), 0, mxGetPr(plhs[0])), "gpz:dxy");
        return;
    }
    if (!strcmp(cmd, "pinv_mode")) {
        if (nrhs != 2) mexErrMsgIdAndTxt("gpz:usage", "pinv_mode needs a mode");
        g_pinv = (int)mxGetScalar(prhs[1]);
        if (g_pinv < -1 || g_pinv > 1) { g_pinv = 0; mexErrMsgIdAndTxt("gpz:usage", "pinv_mode: -1, 0 or 1"); }
        if (g_mg)
            for (int32_t r = 0; r < gpz_mgpu_size(g_mg); ++r) CHECK(gpz_ctx_set_pinv_mode(gpz_mgpu_ctx(g_mg, r), g_pinv), "gpz:pinv_mode");
        return;
    }
    if (!strcmp(cmd, "eval") || !strcmp(cmd, "solve")) {
        if (nrhs != 9) mexErrMsgIdAndTxt("gpz:usage", "%s needs theta,model,X,Y,Psi,omega,training,validation", cmd);
        ensure_context(prhs + 2);
        const double *theta = theta_of(prhs[1]);
        if (cmd[0] == 'e') {
            const mwSize p = (mwSize)gpz_mgpu_theta_len(g_mg);
            double f;
            mxArray *g = mxCreateDoubleMatrix(p, 1, mxREAL);
            mxArray *st = mxCreateDoubleMatrix(4, 1, mxREAL);
            mxGetPr(st)[2] = mxGetNaN(); mxGetPr(st)[3] = mxGetNaN();
            const double t0 = now_s();
            if (gpz_mgpu_eval(g_mg, theta, &f, mxGetPr(g), mxGetPr(st), NULL))
                mexErrMsgIdAndTxt("gpz:eval", "%s", gpz_last_error());
            g_eval_s = now_s() - t0;
            plhs[0] = mxCreateDoubleScalar(f);
            if (nlhs > 1) plhs[1] = g; else mxDestroyArray(g);
            if (nlhs > 2) plhs[2] = st; else mxDestroyArray(st);
        } else {
            mwSize dims[3] = {(mwSize)g_m, (mwSize)g_m, (mwSize)g_k};
            plhs[0] = mxCreateDoubleMatrix(g_m, g_k, mxREAL);
            mxArray *iS = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
            mxArray *part = mxCreateDoubleMatrix(1, g_k, mxREAL);
            if (gpz_mgpu_solve(g_mg, theta, mxGetPr(plhs[0]), mxGetPr(iS), mxGetPr(part)))
                mexErrMsgIdAndTxt("gpz:solve", "%s", gpz_last_error());
            if (nlhs > 1) plhs[1] = iS; else mxDestroyArray(iS);
            if (nlhs > 2) plhs[2] = part; else mxDestroyArray(part);
        }
        return;
    }
    if (!strcmp(cmd, "phi")) {
        if (!g_mg) mexErrMsgIdAndTxt("gpz:state", "no live context: call eval or solve first");
        /* the shards hold contiguous row blocks of the training selection, in rank order */
        mwSize n = 0;
        for (int32_t r = 0; r < gpz_mgpu_size(g_mg); ++r) n += (mwSize)gpz_n_train(gpz_mgpu_ctx(g_mg, r));
        plhs[0] = mxCreateDoubleMatrix(n, g_m, mxREAL);
        double *out = mxGetPr(plhs[0]);
        mwSize r0 = 0;
        for (int32_t r = 0; r < gpz_mgpu_size(g_mg); ++r) {
            gpz_ctx *c = gpz_mgpu_ctx(g_mg, r);
            const mwSize nr = (mwSize)gpz_n_train(c);
            if (!nr) continue;
            mxArray *blk = mxCreateDoubleMatrix(nr, g_m, mxREAL);
            if (gpz_get_phi(c, mxGetPr(blk))) mexErrMsgIdAndTxt("gpz:phi", "%s", gpz_last_error());
            for (int j = 0; j < g_m; ++j) memcpy(out + (size_t)j * n + r0, mxGetPr(blk) + (size_t)j * nr, nr * sizeof(double));
            mxDestroyArray(blk);
            r0 += nr;
        }
        return;
    }
    mexErrMsgIdAndTxt("gpz:usage", "unknown command '%s'", cmd);
}

// Host side of libgpz_hip.so: the evaluation context and the C ABI of include/gpz_hip.h.
//
// A context owns the device-resident data of one closure f = @(theta) GPz(theta,model,X,Y,Psi,omega,
// training,validation) (GPz/train.m:40): the training-selected rows of X (both layouts), Y, omega, the
// validation rows, and every work buffer.  An evaluation ships theta down (p doubles) and
// [f, grad, 4 statistics] up; everything else stays in HBM.
//
// Evaluation pipeline (one HIP stream, no host synchronisation until the result copy):
//   unpack theta -> PHI build (+ ln beta, omega*beta) -> SYRK slabs -> reduce -> [all-reduce #1]
//   -> per output: SIGMA, Cholesky, triangular inverse, inv(SIGMA), w, dwda -> T = PHI*[inv|w]
//   -> row epilogue (nu, delta, dbeta, dPHI, column sums) -> dP/dGamma moments -> validation sums
//   -> [all-reduce #2] -> finish (gradient packing, objective, statistics) -> copy out.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/gpz_hip.h"
#include "gpz_dev.h"
#include "gpz_kernels.h"

// Error text: per thread (the caller of a failing entry point reads its own), with the most recent failure of ANY thread behind it -
// a thread that has never failed itself (a caller whose work ran on a worker thread that did not hand its text back) still gets a
// message instead of an empty string.
static thread_local std::string g_err;
static std::mutex g_err_any_mu;
static std::string g_err_any;
static void set_error(const char *text) {
    g_err = text;
    std::lock_guard<std::mutex> g(g_err_any_mu);
    g_err_any = text;
}
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error(buf);
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(GPZ_ERR_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *gpz_last_error(void) {
    if (g_err.empty()) {
        static thread_local std::string other;   // (a copy: the shared text may change under the caller)
        std::lock_guard<std::mutex> g(g_err_any_mu);
        other = g_err_any;
        return other.c_str();
    }
    return g_err.c_str();
}
// the same error channel for the other host-side translation units (gpz_mgpu.hip)
int gpz_fail(int code, const char *fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error(buf);
    return code;
}
extern "C" int gpz_version(void) { return GPZ_VERSION; }

static int method_id_of(const char *m) {
    static const char *names[6] = {"GL", "VL", "GD", "VD", "GC", "VC"};
    for (int i = 0; i < 6; ++i)
        if (m[0] == names[i][0] && m[1] == names[i][1]) return i;
    return -1;
}
static int g_dim_of(int mid, int m, int d) {
    switch (mid) {
        case 0: return 1;
        case 1: return m;
        case 2: return d;
        case 3: return m * d;
        case 4: return d * d;
        default: return d * d * m;
    }
}
// dimensions the PHI / moment kernels are instantiated for; d is zero-padded up to the next one
static int pad_dim(int d) {
    static const int sup[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20};
    for (int s : sup)
        if (d <= s) return s;
    return d;   // wider inputs: the runtime-d kernels of k_wide.hip, no padding
}
static inline int rup(long v, int q) { return (int)(((v + q - 1) / q) * q); }

// device allocation bookkeeping.  Released blocks go to a per-device cache keyed by their exact size instead of back to the
// runtime: the stand-alone entry points (getPHI, predict*, prior ...) build and drop ~30 buffers per call, predict.m calls them
// once per NaN-pattern group, and hipFree (a device synchronisation + unmap, 38 us on average here) was 30 % of a 79-group
// predict() (profiles/README.md, round 3).  At most GPZ_CACHE_CAP bytes per device stay cached (blocks above GPZ_CACHE_BLOCK_MAX
// are freed directly: the limit sits just above the 2 GiB runtime-d workspace of d > 20, which a many-group predict() with
// missing values would otherwise allocate and free once per group); gpz_release_cached_memory() gives everything back.
#include <map>
#include <mutex>
#define GPZ_CACHE_CAP_DEFAULT (4096UL << 20)
#define GPZ_CACHE_BLOCK_MAX (2304UL << 20)
// GPZ_CACHE_CAP_MB (environment, read once): bytes per device that may stay cached, for hosts that share the GPU with another
// allocator (PyTorch's, a second process); 0 = no caching at all.  INTEGRATION.md, "device memory".
static size_t cache_cap() {
    static const size_t cap = [] {
        const long mb = gpz_options_load().cache_cap_mb;
        return mb >= 0 ? (size_t)mb << 20 : (size_t)GPZ_CACHE_CAP_DEFAULT;
    }();
    return cap;
}
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> blocks;   // (device, bytes) -> pointer
    std::map<int, size_t> held;                              // bytes cached per device
};
static DevCache &dev_cache() {
    static DevCache *c = new DevCache();   // never destroyed: the HIP runtime may be gone before static destructors run
    return *c;
}
static void *cache_take(int dev, size_t bytes) {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.blocks.find({dev, bytes});
    if (it == c.blocks.end()) return nullptr;
    void *p = it->second;
    c.blocks.erase(it);
    c.held[dev] -= bytes;
    return p;
}
static bool cache_give(int dev, size_t bytes, void *p) {
    if (bytes > GPZ_CACHE_BLOCK_MAX) return false;
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.mu);
    if (c.held[dev] + bytes > cache_cap()) return false;
    c.blocks.insert({{dev, bytes}, p});
    c.held[dev] += bytes;
    return true;
}
// Frees the cached blocks only.  This is what a failed hipMalloc retries with: it takes no lock but the block cache's own, so it
// is safe under a model-table entry's mutex (predict_missing_cov allocates while it holds one - calling the full release there
// locked that same non-recursive mutex again, and would have freed the tables the call was using).
static void cache_release_blocks() {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : c.blocks) {
        (void)hipSetDevice(kv.first.first);
        (void)hipFree(kv.second);
    }
    (void)hipSetDevice(cur);
    c.blocks.clear();
    c.held.clear();
}
static void pmc_model_cache_release_all();
extern "C" void gpz_release_cached_memory(void) {
    cache_release_blocks();
    pmc_model_cache_release_all();   // after the block cache's lock is gone; entries a running prediction holds are skipped
}
// test hook (gpz_debug_fail_alloc(k), include/gpz_hip.h): the k-th hipMalloc from now on reports out-of-memory once, so the retry
// path can be exercised without exhausting 288 GB
static std::atomic<long> g_alloc_fault_countdown{0};
extern "C" void gpz_debug_fail_alloc(int64_t kth) { g_alloc_fault_countdown.store(kth > 0 ? (long)kth : 0); }
static bool alloc_fault_due() {
    if (g_alloc_fault_countdown.load(std::memory_order_relaxed) <= 0) return false;
    return g_alloc_fault_countdown.fetch_sub(1) == 1;
}

struct Arena {
    struct Blk { void *p; size_t bytes; int dev; };
    std::vector<Blk> blks;
    size_t bytes = 0;
    template <typename T>
    int alloc(T **p, size_t count) {
        *p = nullptr;
        if (count == 0) count = 1;
        const size_t nb = count * sizeof(T);
        int dev = 0;
        (void)hipGetDevice(&dev);
        const bool fault = alloc_fault_due();   // (test hook: this allocation finds neither a cached block nor memory at first)
        void *q = fault ? nullptr : cache_take(dev, nb);
        if (!q) {
            hipError_t e = fault ? hipErrorOutOfMemory : hipMalloc(&q, nb);
            if (e != hipSuccess) {   // the block cache may be what is in the way: give it back and try once more
                (void)hipGetLastError();
                cache_release_blocks();
                e = hipMalloc(&q, nb);
            }
            if (e != hipSuccess) return fail(GPZ_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", nb, hipGetErrorString(e));
        }
        *p = (T *)q;
        blks.push_back({q, nb, dev});
        bytes += nb;
        return 0;
    }
    void release() {
        // one device synchronisation per release (hipFree did one per block): nothing may still be running on a block that
        // the next caller - possibly on another stream - takes from the cache
        int cur = 0, last = -1;
        (void)hipGetDevice(&cur);
        for (const Blk &b : blks)
            if (b.dev != last) {   // (every block of an arena normally sits on one device: one synchronisation)
                (void)hipSetDevice(b.dev);
                (void)hipDeviceSynchronize();
                last = b.dev;
            }
        if (last != -1 && last != cur) (void)hipSetDevice(cur);
        for (const Blk &b : blks)
            if (!cache_give(b.dev, b.bytes, b.p)) {
                (void)hipFree(b.p);
            }
        blks.clear();
    }
};

struct RowSet {          // a device-resident row selection of the data
    int n = 0, n_pad = 0;
    double *Xc = nullptr;   // de x n_pad
    double *Xr = nullptr;   // n_pad x de
    double *Y = nullptr;    // k x n_pad
    double *om = nullptr;   // n_pad (nullptr => ones)
    // diagonal kinds only: input-noise variances and the observed-dimension mask (nullptr => absent)
    double *Psic = nullptr, *Psir = nullptr;   // de x n_pad, n_pad x de (0 where the input is missing)
    double *Mc = nullptr, *Mr = nullptr;       // 1.0 observed / 0.0 missing
    double *ucnt = nullptr;                    // number of missing dimensions per row
    // covariance kinds, general path (Psi cube and/or missing dimensions)
    int *gid = nullptr, *rows_by_group = nullptr;
    int *orig = nullptr;                       // GC/VC general path: rows are stored sorted by NaN pattern; orig[r] = position of
    std::vector<int> orig_h;                   // stored row r in the caller's row order (device / host copy)
    double *Psi3 = nullptr;                    // n_pad x d*d
    float *PsiT = nullptr;                     // dtype f32: packed lower triangles, element-major [e][n_pad] (k_psi32.hip)
    int psi_diag = 0;                          // every Psi_i of this row set is diagonal: PsiT holds only the diagonals
    std::vector<int> group_begin;              // offsets into rows_by_group (size G+1)
    int *wgtab = nullptr;                      // missing dimensions without input noise: workgroup table of the one-launch
    int nwg_tab = 0;                           // PHI build over all patterns (PhiArgs::wgtab)
};

struct StageTimer {
    std::vector<const char *> names;
    std::vector<double> ms;
    std::vector<int64_t> calls;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    int find(const char *n) {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n || strcmp(names[i], n) == 0) return (int)i;
        names.push_back(n);
        ms.push_back(0.0);
        calls.push_back(0);
        return (int)names.size() - 1;
    }
    hipEvent_t get() {
        if (pool_used == pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            pool.push_back(e);
        }
        return pool[pool_used++];
    }
};

struct gpz_ctx {
    gpz_desc desc;
    gpz_options opt = gpz_options_load();   // latched for the life of the context (gpz_options.h)
    int mid = 0, kind = 0, d = 0, de = 0, m = 0, mp = 0, mq = 0, k = 1, hetero = 0, g_dim = 0;
    long p = 0;
    int device = 0;
    hipStream_t st = nullptr;
    Arena ar;
    RowSet tr, va;
    // parameters
    double *theta_d = nullptr;
    GpzParams pr{};
    // big buffers
    double *Phi = nullptr, *T = nullptr, *dL = nullptr;
    double *lnbeta = nullptr, *wbeta = nullptr, *phiw = nullptr;
    double *lnbeta_v = nullptr, *phiw_v = nullptr;
    double *slab = nullptr;
    size_t slab_count = 0;
    int nsplit = 1, rows_per_split = 16;       // off-diagonal tiles of PHI' W PHI
    int nsplit_d = 1, rows_per_split_d = 16;   // diagonal tiles (9/16 of the work per row: longer row ranges)
    int nsplit_l = 1, rows_per_split_l = 16;
    // communication buffers
    double *comm1 = nullptr;   // [k * mp*mp | GPZ_NS]
    size_t comm1_count = 0;
    double *comm2 = nullptr;   // [m*nm | k*2*mp | k*4 | GPZ_NS]
    size_t comm2_count = 0;
    int nm = 0;
    // m x m work
    double *A = nullptr, *Lm = nullptr, *Wm = nullptr, *Tmp = nullptr, *Sinv = nullptr, *Bext = nullptr;
    double *w = nullptr, *dwda = nullptr, *dgi = nullptr, *logdet = nullptr;
    int *info = nullptr;
    // row epilogue / moments
    double *colslab = nullptr, *scal_slab = nullptr;
    int nwg_rows = 1;
    double *mom_slab = nullptr;
    int nchunk = 1, rows_per_chunk = 1;
    // Row-tile streaming (tile_rows > 0; SURVEY.md section 5 "row-tile streaming"): PHI, T and the nu partials hold ONE tile of rows and
    // the evaluation walks the tiles twice - stage A: PHI -> PHI'W PHI accumulated over the tiles; tail: PHI again -> T-GEMM -> row
    // scalars -> moment sums into the tile's own chunks of the slab.  The per-row vectors (ln beta, omega beta, PHI w, row scalars)
    // stay whole.  Chosen when PHI + T would not fit the device (or forced by GPZ_ROW_TILE, tests); plain route only (no Psi, no
    // missing values in GC/VC).
    int tile_rows = 0, ntiles = 1, tile_nchunk = 1, tile_rpc = 1;
    double *tile_rstats = nullptr;                        // [ntiles][GPZ_NS]: the tiles' row-scalar sums
    double *partial = nullptr, *rstats = nullptr, *dGfull = nullptr, *spart = nullptr;
    double *nupart = nullptr, *rowscal = nullptr, *frec = nullptr;   // fused path
    bool fused = true;   // dPHI formed on the fly, output by output (no dPHI / dL matrices): k == 1, or k > 1 on the tuned kernels
    double *phipart = nullptr;   // PHI-build column-group partial sums (small row counts)
    int phipart_groups = 0;
    int nslots = 0;
    double *out_d = nullptr;
    double *out_h = nullptr, *theta_h = nullptr;   // pinned
    gpz_allreduce_fn ar_fn = nullptr;
    void *ar_user = nullptr;
    void *priv = nullptr;                 // owned by whoever attached it (the RCCL communicator of gpz_ctx_init_rccl),
    void (*priv_free)(void *) = nullptr;  // released with the context
    bool timing = false;
    // One evaluation = ~40 launches on one stream between the upload of theta and the download of the result block, every argument
    // fixed for the life of the context: from the third gpz_eval on it is replayed as a hipGraph (single rank, host theta, stage
    // timing off).  graph_state: 0 first call (eager), 1 capture on this call, 2 replay, -1 disabled (capture failed / GPZ_NO_GRAPH)
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_st = nullptr;   // the recording runs on a stream of its own (the null stream cannot be captured); the graph is launched on st
    int graph_state = 0;
    bool capturing = false;
    StageTimer tm;
    bool phi_valid = false;
    bool has_psi = false, has_missing = false;
    // truncating pseudo-inverse route (inv_logdet.m:7-12): 0 = when k_cond_flag asks for it, 1 = always, -1 = never
    int pinv_mode = 0;
    double *g_dev_out = nullptr;          // set for the duration of gpz_eval_dev: device destination of the gradient
    double pinv_last[4] = {0, 0, 0, 0};   // [route taken, rank kept, max singular value, Jacobi sweeps] of the last call
    // general covariance-kind path
    bool gen = false;
    bool psi_fast = false;   // gen && Psi && d <= 10 && fp64: register-resident kernels (k_psi.hip), missing dimensions included
    bool psi_miss = false;   // psi_fast with more than one NaN pattern (or a pattern with missing dimensions)
    bool psi32 = false;      // dtype f32 && gen && Psi && no missing dims: fp32 register-resident kernels (k_psi32.hip)
    int psi_kind_in = 0;     // layout of the caller's Psi: 1 n x d (diagonal kinds), 2 d x d x n cube, 3 n x d variances = diagonal cubes (GC/VC)
    bool need_psi3 = true;   // keep the fp64 cube on the device (prediction / fp64 pair kernels); the fp32 evaluation path reads PsiT only
    bool psi32_agreed = false;   // sharded runs: the ranks have agreed on diagonal vs full Psi (first evaluation)
    int ngroups = 0, nrec = 0;
    std::vector<std::vector<unsigned char>> pats;   // observed flags per pattern (host copy)
    bool pats_fixed = false;                        // table given by the caller (sharded runs): rows must match an entry
    unsigned char *pat_d = nullptr;
    double *prep_ws = nullptr;                            // QR workspace of the covariance kinds when Gamma_j does not fit the LDS
    double *Sig = nullptr, *iSig = nullptr, *lnS = nullptr, *Phi_v = nullptr, *gen_slab = nullptr, *psi32_raw = nullptr;
    double *gc_minv = nullptr;   // GC + Psi, 10 < d <= 32 (fp64): -inv(Sigma + Psi_i) of every training row as 4 x 4 tiles (k_cpsi4_minv)
    double *gcq_A = nullptr, *gcq_B = nullptr;   // ... without missing dimensions: operands of the dense form of the PHI build (k_gcq_*)
    // missing dimensions without input noise: per-pattern parameter blocks and moment slabs of the tuned kernels
    double *RcP = nullptr, *gen_tslab = nullptr, *gen_frec = nullptr, *fin_part = nullptr;
    double *gen_ws = nullptr;   // d > 20: runtime-d workspace of the general-path kernels (k_gen.hip), else nullptr
    int gen_tnch = 1;
    int *mom_chunktab = nullptr, *mom_segtab = nullptr;   // moment chunks {first row, end row} that respect the pattern
    int mom_nchunk = 0;                                   // boundaries, and each pattern's range of chunks
    int gen_nchunk = 1;
};

// ---- stage timing ------------------------------------------------------------------------------
struct Stage {
    gpz_ctx *c;
    int idx = -1;
    hipEvent_t e0{}, e1{};
    Stage(gpz_ctx *c_, const char *name) : c(c_) {
        if (!c->timing) return;
        idx = c->tm.find(name);
        e0 = c->tm.get();
        e1 = c->tm.get();
        (void)hipEventRecord(e0, c->st);
    }
    ~Stage() {
        if (idx < 0) return;
        (void)hipEventRecord(e1, c->st);
        c->tm.pending.push_back({idx, {e0, e1}});
    }
};
static void collect_timings(gpz_ctx *c) {
    for (auto &pe : c->tm.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
            c->tm.ms[pe.first] += ms;
            c->tm.calls[pe.first] += 1;
        }
    }
    c->tm.pending.clear();
    c->tm.pool_used = 0;
}

// ---- context creation ---------------------------------------------------------------------------
static int upload_rowset(gpz_ctx *c, RowSet &rs, int64_t n_tot, const double *X, const double *Y, const double *omega,
                         const uint8_t *mask, bool need_xr, const double *Psi = nullptr) {
    const int d = c->d, de = c->de, k = c->k;
    std::vector<int64_t> idx;
    idx.reserve((size_t)n_tot);
    for (int64_t i = 0; i < n_tot; ++i)
        if (!mask || mask[i]) idx.push_back(i);                            // X(selection,:)  getPHI.m:14
    if (c->gen) {
        // Group the rows by NaN pattern up front (ids in first-occurrence order over the rows seen so far, getPHI.m:43-54)
        // and store them sorted by pattern: every pattern is then a contiguous row range that the tuned kernels can
        // work on.  Sums over rows do not care about the order; per-row outputs are un-permuted on the way out.
        std::vector<int> hg0(idx.size());
        for (size_t r = 0; r < idx.size(); ++r) {
            std::vector<unsigned char> pt(d);
            for (int c_ = 0; c_ < d; ++c_) { const double xv = X[(size_t)c_ * n_tot + idx[r]]; pt[c_] = (xv != xv) ? 0 : 1; }
            int g = -1;
            for (size_t q = 0; q < c->pats.size(); ++q)
                if (c->pats[q] == pt) { g = (int)q; break; }
            if (g < 0) {
                if (c->pats_fixed) return fail(GPZ_ERR_ARG, "row %lld has a NaN pattern that is not in the given pattern table", (long long)idx[r]);
                c->pats.push_back(pt);
                g = (int)c->pats.size() - 1;
            }
            hg0[r] = g;
        }
        std::vector<int> ord(idx.size());
        for (size_t r = 0; r < idx.size(); ++r) ord[r] = (int)r;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return hg0[a] < hg0[b]; });
        std::vector<int64_t> idx2(idx.size());
        rs.orig_h.resize(idx.size());
        for (size_t r = 0; r < idx.size(); ++r) { idx2[r] = idx[ord[r]]; rs.orig_h[r] = ord[r]; }
        idx.swap(idx2);
    }
    rs.n = (int)idx.size();
    rs.n_pad = rup(rs.n > 0 ? rs.n : 1, 1024) + (c->gen ? 1024 : 0);   // general path: slack for per-pattern launches of the tuned kernels   // multiple of the PHI kernel's rows per workgroup (4 waves x 64 lanes x 4 rows)
    const size_t np = (size_t)rs.n_pad;
    // column-layout (de x n_pad) and row-layout (n_pad x de) uploads of a per-(row, dim) quantity
    auto up2 = [&](const std::vector<double> &cm, double **dc, double **dr, bool want_r) -> int {
        if (int e = c->ar.alloc(dc, np * de)) return e;
        HIPCHK(hipMemcpy(*dc, cm.data(), np * de * sizeof(double), hipMemcpyHostToDevice));
        if (want_r) {
            std::vector<double> rm(np * (size_t)de, 0.0);
            for (int c_ = 0; c_ < de; ++c_)
                for (size_t r = 0; r < idx.size(); ++r) rm[r * de + c_] = cm[(size_t)c_ * np + r];
            if (int e = c->ar.alloc(dr, np * de)) return e;
            HIPCHK(hipMemcpy(*dr, rm.data(), np * de * sizeof(double), hipMemcpyHostToDevice));
        }
        return 0;
    };
    std::vector<double> h(np * (size_t)de, 0.0), hm;
    bool any_missing = false;
    for (int c_ = 0; c_ < d; ++c_)
        for (size_t r = 0; r < idx.size(); ++r) {
            const double xv = X[(size_t)c_ * n_tot + idx[r]];
            if (xv != xv) any_missing = true;                              // isnan(X)  getPHI.m:43
            h[(size_t)c_ * np + r] = (xv != xv) ? 0.0 : xv;
        }
    if (int e = up2(h, &rs.Xc, &rs.Xr, need_xr)) return e;
    if (any_missing || c->has_missing) {
        c->has_missing = true;
        hm.assign(np * (size_t)de, 1.0);
        std::vector<double> hu(np, 0.0);
        for (int c_ = 0; c_ < d; ++c_)
            for (size_t r = 0; r < idx.size(); ++r) {
                const double xv = X[(size_t)c_ * n_tot + idx[r]];
                if (xv != xv) { hm[(size_t)c_ * np + r] = 0.0; hu[r] += 1.0; }
            }
        if (int e = up2(hm, &rs.Mc, &rs.Mr, need_xr)) return e;
        if (int e = c->ar.alloc(&rs.ucnt, np)) return e;
        HIPCHK(hipMemcpy(rs.ucnt, hu.data(), np * sizeof(double), hipMemcpyHostToDevice));
    }
    if (c->gen) {
        // pattern id per row (global table c->pats, first-occurrence order over the rows seen so far; getPHI.m:43-54)
        std::vector<int> hg(np, 0);
        for (size_t r = 0; r < idx.size(); ++r) {
            std::vector<unsigned char> pt(d);
            for (int c_ = 0; c_ < d; ++c_) { const double xv = X[(size_t)c_ * n_tot + idx[r]]; pt[c_] = (xv != xv) ? 0 : 1; }
            int g = -1;
            for (size_t q = 0; q < c->pats.size(); ++q)
                if (c->pats[q] == pt) { g = (int)q; break; }
            if (g < 0) return fail(GPZ_ERR_ARG, "internal: pattern table changed during the upload");
            hg[r] = g;
        }
        if (int e = c->ar.alloc(&rs.orig, idx.size() ? idx.size() : 1)) return e;
        if (!idx.empty()) HIPCHK(hipMemcpy(rs.orig, rs.orig_h.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
        if (int e = c->ar.alloc(&rs.gid, np)) return e;
        HIPCHK(hipMemcpy(rs.gid, hg.data(), np * sizeof(int), hipMemcpyHostToDevice));
        const int G = (int)c->pats.size();
        std::vector<int> cnt(G + 1, 0), order(idx.size());
        for (size_t r = 0; r < idx.size(); ++r) cnt[hg[r] + 1]++;
        for (int g = 0; g < G; ++g) cnt[g + 1] += cnt[g];
        rs.group_begin = cnt;
        std::vector<int> pos(cnt.begin(), cnt.end() - 1);
        for (size_t r = 0; r < idx.size(); ++r) order[pos[hg[r]]++] = (int)r;
        if (int e = c->ar.alloc(&rs.rows_by_group, idx.size() ? idx.size() : 1)) return e;
        if (!idx.empty()) HIPCHK(hipMemcpy(rs.rows_by_group, order.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
        if (Psi) {   // d x d x n_tot cube (fixPsi.m:22-38): Psi(:,:,i) is contiguous; or (psi_kind 3) the n_tot x d variances the
                     // cubes' diagonals were built from (fixPsi.m:27-31), expanded here instead of by the caller
            const bool dvar = c->psi_kind_in == 3;
            auto psi_at = [&](size_t r, int a, int b) -> double {
                if (dvar) return a == b ? Psi[(size_t)a * n_tot + idx[r]] : 0.0;
                return Psi[(size_t)idx[r] * d * d + a + (size_t)d * b];
            };
            {   // every Psi_i diagonal?  (what fixPsi.m builds from per-dimension variances; prediction and the fp32 pair kernels
                // have cheaper forms for it)
                bool dg = true;
                if (!dvar)
                    for (size_t r = 0; r < idx.size() && dg; ++r)
                        for (int a = 0; a < d && dg; ++a)
                            for (int b = 0; b < d; ++b)
                                if (a != b && psi_at(r, a, b) != 0.0) { dg = false; break; }
                rs.psi_diag = dg ? 1 : 0;
            }
            if (c->need_psi3 || !c->psi32) {
                std::vector<double> hp(np * (size_t)d * d, 0.0);
                for (size_t r = 0; r < idx.size(); ++r) {
                    if (dvar) for (int a = 0; a < d; ++a) hp[r * d * d + a + (size_t)d * a] = Psi[(size_t)a * n_tot + idx[r]];
                    else memcpy(&hp[r * d * d], Psi + (size_t)idx[r] * d * d, (size_t)d * d * sizeof(double));
                }
                if (int e = c->ar.alloc(&rs.Psi3, np * d * d)) return e;
                HIPCHK(hipMemcpy(rs.Psi3, hp.data(), np * d * d * sizeof(double), hipMemcpyHostToDevice));
            }
            if (c->psi32) {
                const int D = psi32_pad_dim(d);
                const bool diag = rs.psi_diag != 0;
                const size_t ne = diag ? (size_t)D : (size_t)D * (D + 1) / 2;
                std::vector<float> ht(ne * np, 0.0f);
                for (size_t r = 0; r < idx.size(); ++r) {
                    if (diag) {
                        for (int a = 0; a < d; ++a) ht[(size_t)a * np + r] = (float)psi_at(r, a, a);
                    } else {
                        for (int a = 0; a < d; ++a)
                            for (int b = 0; b <= a; ++b)   // lower triangle of the symmetric Psi(:,:,i): element (a, b)
                                ht[((size_t)a * (a + 1) / 2 + b) * np + r] = (float)psi_at(r, a, b);
                    }
                }
                if (int e = c->ar.alloc(&rs.PsiT, ne * np)) return e;
                HIPCHK(hipMemcpy(rs.PsiT, ht.data(), ne * np * sizeof(float), hipMemcpyHostToDevice));
            }
        }
    }
    if (Psi && !c->gen) {   // n_tot x d (fixPsi.m:42-53); entries of missing dimensions are never read by the reference
        std::vector<double> hp(np * (size_t)de, 0.0);
        for (int c_ = 0; c_ < d; ++c_)
            for (size_t r = 0; r < idx.size(); ++r) {
                const double xv = X[(size_t)c_ * n_tot + idx[r]];
                hp[(size_t)c_ * np + r] = (xv != xv) ? 0.0 : Psi[(size_t)c_ * n_tot + idx[r]];
            }
        if (int e = up2(hp, &rs.Psic, &rs.Psir, need_xr)) return e;
    }
    std::vector<double> hy(np * (size_t)k, 0.0);
    for (int o = 0; o < k; ++o)
        for (size_t r = 0; r < idx.size(); ++r) hy[(size_t)o * np + r] = Y[(size_t)o * n_tot + idx[r]];
    if (int e = c->ar.alloc(&rs.Y, np * k)) return e;
    HIPCHK(hipMemcpy(rs.Y, hy.data(), np * k * sizeof(double), hipMemcpyHostToDevice));
    if (omega) {
        std::vector<double> ho(np, 0.0);
        for (size_t r = 0; r < idx.size(); ++r) ho[r] = omega[idx[r]];
        if (int e = c->ar.alloc(&rs.om, np)) return e;
        HIPCHK(hipMemcpy(rs.om, ho.data(), np * sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

static int has_nan(const double *X, int64_t count) {
    for (int64_t i = 0; i < count; ++i)
        if (X[i] != X[i]) return 1;
    return 0;
}

static int setup_model(gpz_ctx *c, const gpz_desc *desc) {
    c->desc = *desc;
    c->mid = method_id_of(desc->method);
    if (c->mid < 0) return fail(GPZ_ERR_ARG, "unknown method '%.2s'", desc->method);
    if (desc->d < 1 || desc->m < 1 || desc->k < 1) return fail(GPZ_ERR_ARG, "d, m, k must be >= 1");
    c->kind = c->mid >= 4 ? GPZ_KIND_COV : GPZ_KIND_DIAG;
    c->d = desc->d;
    c->de = pad_dim(desc->d);
    c->m = desc->m;
    c->k = desc->k;
    c->hetero = desc->heteroscedastic ? 1 : 0;
    c->g_dim = g_dim_of(c->mid, c->m, c->d);
    c->p = (long)c->m * c->d + c->g_dim + (long)c->m * c->k + c->k + (c->hetero ? 2L * c->m * c->k : 0);
    c->mp = rup(c->m + c->k, 16);
    c->mq = rup(c->m, GPZ_CH_NB);
    c->nm = (c->kind == GPZ_KIND_COV) ? c->de + c->de * (c->de + 1) / 2 : 2 * c->de;
    c->device = desc->device;
    c->st = (hipStream_t)desc->stream;
    return 0;
}

static int alloc_params(gpz_ctx *c) {
    const size_t m = c->m, de = c->de, k = c->k;
    if (int e = c->ar.alloc(&c->theta_d, (size_t)c->p)) return e;
    if (int e = c->ar.alloc(&c->pr.P, m * de)) return e;
    if (int e = c->ar.alloc(&c->pr.G, c->kind == GPZ_KIND_COV ? m * de * de : m * de)) return e;
    if (int e = c->ar.alloc(&c->pr.G2, m * de)) return e;
    if (int e = c->ar.alloc(&c->pr.Rc, m * (de * (de + 1) / 2 + de))) return e;
    if (const size_t wl = c->kind == GPZ_KIND_COV ? prep_cov_ws_len(c->m, c->de) : 0)
        if (int e = c->ar.alloc(&c->prep_ws, wl)) return e;
    if (int e = c->ar.alloc(&c->pr.lnAlpha, m * k)) return e;
    if (int e = c->ar.alloc(&c->pr.alpha, m * k)) return e;
    if (int e = c->ar.alloc(&c->pr.b, k)) return e;
    if (int e = c->ar.alloc(&c->pr.v, m * k)) return e;
    if (int e = c->ar.alloc(&c->pr.lnTau, m * k)) return e;
    if (int e = c->ar.alloc(&c->pr.tau, m * k)) return e;
    return 0;
}

static int alloc_mm(gpz_ctx *c) {   // m x m stage buffers
    const size_t mq2 = (size_t)c->mq * c->mq, m = c->m, k = c->k;
    if (int e = c->ar.alloc(&c->A, mq2)) return e;
    if (int e = c->ar.alloc(&c->Lm, mq2)) return e;
    if (int e = c->ar.alloc(&c->Wm, mq2)) return e;
    if (int e = c->ar.alloc(&c->Tmp, mq2)) return e;
    if (int e = c->ar.alloc(&c->Sinv, mq2)) return e;
    if (int e = c->ar.alloc(&c->Bext, (size_t)c->mp * c->mp)) return e;
    if (int e = c->ar.alloc(&c->w, m * k)) return e;
    if (int e = c->ar.alloc(&c->dwda, m * k)) return e;
    if (int e = c->ar.alloc(&c->dgi, m * k)) return e;
    if (int e = c->ar.alloc(&c->logdet, k)) return e;
    if (int e = c->ar.alloc(&c->info, 4)) return e;   // [pivot failure, truncation flag | ticket of k_cond_norms, -]
    HIPCHK(hipMemset(c->info, 0, 4 * sizeof(int)));
    // LAUUM split: upper 128-tiles of an mq x mq product with mq rows
    const int ntq = (c->mq + 127) / 128, npq = ntq * (ntq + 1) / 2;
    int ns = (256 + npq - 1) / npq;
    if (ns > c->mq / 32) ns = c->mq / 32;
    if (ns < 1) ns = 1;
    c->rows_per_split_l = rup((c->mq + ns - 1) / ns, 16);
    c->nsplit_l = (c->mq + c->rows_per_split_l - 1) / c->rows_per_split_l;
    return 0;
}

// Data-dependent part of a context: path selection (tuned / general), row sets, pattern table, parameter block.
static int setup_data(gpz_ctx *c, int64_t n_tot, const double *X, const double *Y, const double *Psi, int32_t psi_kind,
                      const double *omega, const uint8_t *training, const uint8_t *validation,
                      const uint8_t *patterns = nullptr, int32_t n_patterns = 0) {
    const gpz_desc *desc = &c->desc;
    int rc = 0;
    if ((Psi != nullptr) != (psi_kind != 0)) return fail(GPZ_ERR_ARG, "Psi and psi_kind disagree");
    if (psi_kind < 0 || psi_kind > 3) return fail(GPZ_ERR_ARG, "psi_kind must be 0..3");
    c->psi_kind_in = psi_kind;
    const bool xnan = has_nan(X, n_tot * (int64_t)c->d) != 0;
    // a given pattern table means "the data set has missing values": every rank takes the general path then, also one
    // whose own rows happen to be complete (the second all-reduce carries one record block per pattern)
    const bool table = patterns && n_patterns > 0;
    if (c->kind == GPZ_KIND_COV && (Psi || xnan || table)) {
        // general path: per-pair d x d factorisations (k_gen.hip)
        if (Psi && psi_kind != 2 && psi_kind != 3)
            return fail(GPZ_ERR_ARG, "GC/VC take Psi as a d x d x n cube (fixPsi.m:22-38) or as n x d variances (psi_kind 3)");
        // the NaN-pattern table is built per rank in first-occurrence order: shards would disagree on the ids and on the
        // size of the second all-reduce, so a sharded run must be given the table of the whole data set
        if (desc->world > 1 && xnan && !table)
            return fail(GPZ_ERR_UNSUPPORTED, "row-sharded GC/VC with missing values needs the global NaN-pattern table "
                                             "(gpz_ctx_create_sharded)");
        c->gen = true;
        if (table) {   // 1 = missing, as isnan(X) (getPHI.m:43); stored here as observed flags
            for (int g = 0; g < n_patterns; ++g) {
                std::vector<unsigned char> pt((size_t)c->d);
                for (int q = 0; q < c->d; ++q) pt[q] = patterns[(size_t)g * c->d + q] ? 0 : 1;
                c->pats.push_back(pt);
            }
            c->pats_fixed = true;
        } else if (!xnan) c->pats.assign(1, std::vector<unsigned char>((size_t)c->d, (unsigned char)1));   // one pattern: all observed
        // dtype f32 selects the fp32 pair kernels only where EVERY rank does: a given pattern table means some rank holds
        // missing values (it takes the fp64 route and posts one record block per pattern), so nobody may take the fp32 route
        c->psi32 = desc->dtype == GPZ_F32 && Psi && !xnan && !table && c->d <= 20;   // fp32 pair kernels: d <= 20
    }
    if (desc->dtype != GPZ_F64 && desc->dtype != GPZ_F32) return fail(GPZ_ERR_ARG, "dtype must be GPZ_F64 or GPZ_F32");
    if (Psi && c->kind == GPZ_KIND_DIAG && psi_kind != 1)
        return fail(GPZ_ERR_ARG, "diagonal kinds take Psi as n x d (fixPsi.m:42-53)");
    c->has_psi = Psi != nullptr;
    if (c->has_psi && !c->gen) c->nm = 3 * c->de;
    // one missing value anywhere (training or validation rows) switches the mask arrays on for both row sets
    c->has_missing = xnan;
    if (hipSetDevice(c->device) != hipSuccess) return fail(GPZ_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if ((rc = upload_rowset(c, c->tr, n_tot, X, Y, omega, training, true, Psi))) return rc;
    if (c->tr.n < 1 && desc->world <= 1) return fail(GPZ_ERR_ARG, "training mask selects no rows");
    bool any_valid = false;
    if (validation)
        for (int64_t i = 0; i < n_tot && !any_valid; ++i) any_valid = validation[i] != 0;
    // with sharding a rank may hold no validation rows while others do: the caller signals "validation in use"
    // by passing a non-NULL mask
    if (validation && (any_valid || desc->world > 1)) {
        if ((rc = upload_rowset(c, c->va, n_tot, X, Y, omega, validation, c->gen, Psi))) return rc;
        if (!omega) c->va.om = nullptr;
    }
    if (c->gen) {
        c->ngroups = (int)c->pats.size();
        // the training rows were grouped before the validation rows could add their own patterns: give every row set an
        // (empty) range for the patterns it has never seen
        for (RowSet *rs : {&c->tr, &c->va})
            while ((int)rs->group_begin.size() < c->ngroups + 1)
                rs->group_begin.push_back(rs->group_begin.empty() ? 0 : rs->group_begin.back());
        if (!c->has_psi) {
            const int rpw = phi_cov_rows_per_wg(c->de, c->k);
            for (RowSet *rs : {&c->tr, &c->va}) {
                std::vector<int> tab;
                for (int g = 0; g < c->ngroups; ++g)
                    for (int r = rs->group_begin[g]; r < rs->group_begin[g + 1]; r += rpw) {
                        const int e[4] = {r, rs->group_begin[g + 1], g, 0};
                        tab.insert(tab.end(), e, e + 4);
                    }
                rs->nwg_tab = (int)(tab.size() / 4);
                if (!rs->nwg_tab) continue;
                if ((rc = c->ar.alloc(&rs->wgtab, tab.size()))) return rc;
                if (hipMemcpy(rs->wgtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(GPZ_ERR_HIP, "copy failed");
            }
        }
        c->psi_fast = !c->psi32 && c->has_psi && psi_fast_path_available(c->d);
        c->psi_miss = c->psi_fast && (c->has_missing || c->ngroups > 1);
        c->nrec = 3 + c->d + c->d * c->d;
        c->nm = c->ngroups * c->nrec;                 // comm2's moment segment holds the [G][m][nrec] records
        std::vector<unsigned char> hp((size_t)c->ngroups * c->d);
        for (int g = 0; g < c->ngroups; ++g) memcpy(&hp[(size_t)g * c->d], c->pats[g].data(), c->d);
        if ((rc = c->ar.alloc(&c->pat_d, hp.size()))) return rc;
        if (hipMemcpy(c->pat_d, hp.data(), hp.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(GPZ_ERR_HIP, "copy failed");
        if ((rc = c->ar.alloc(&c->Sig, (size_t)c->m * c->d * c->d))) return rc;
        if ((rc = c->ar.alloc(&c->iSig, (size_t)c->m * c->d * c->d))) return rc;
        if ((rc = c->ar.alloc(&c->lnS, (size_t)c->ngroups * c->m))) return rc;
        if (!c->has_psi &&
            (rc = c->ar.alloc(&c->RcP, (size_t)c->ngroups * c->m * (c->de * (c->de + 1) / 2 + c->de))))
            return rc;
    }
    // runtime-d workspace of the general-path kernels: the GC/VC routes with input noise or missing values read it in every
    // evaluation; a diagonal kind only in gpz_predict_noisy's pair table (allocated there) - never in an evaluation context
    if (c->d > 20 && c->gen &&
        (rc = c->ar.alloc(&c->gen_ws, (size_t)gen_rt_threads(c->d) * gen_ws_per_thread(c->d))))
        return rc;
    return alloc_params(c);
}

extern "C" int gpz_ctx_create(const gpz_desc *desc, int64_t n_tot, const double *X, const double *Y, const double *Psi,
                              int32_t psi_kind, const double *omega, const uint8_t *training,
                              const uint8_t *validation, gpz_ctx **out) {
    return gpz_ctx_create_sharded(desc, n_tot, X, Y, Psi, psi_kind, omega, training, validation, nullptr, 0, out);
}

extern "C" int gpz_ctx_create_sharded(const gpz_desc *desc, int64_t n_tot, const double *X, const double *Y,
                                      const double *Psi, int32_t psi_kind, const double *omega, const uint8_t *training,
                                      const uint8_t *validation, const uint8_t *patterns, int32_t n_patterns,
                                      gpz_ctx **out) {
    if (!desc || !X || !Y || !out || n_tot < 1) return fail(GPZ_ERR_ARG, "gpz_ctx_create: null argument");
    *out = nullptr;
    gpz_ctx *c = new gpz_ctx();
    gpz_opts_scope opts_scope(&c->opt);
    int rc = setup_model(c, desc);
    if (rc) { delete c; return rc; }
    auto bail = [&](int code) {
        c->ar.release();
        if (c->out_h) (void)hipHostFree(c->out_h);
        if (c->theta_h) (void)hipHostFree(c->theta_h);
        delete c;
        return code;
    };
    c->need_psi3 = false;          // an evaluation context on the fp32 pair kernels never reads the fp64 cube (6.4 GB at config 5)
    if ((rc = setup_data(c, n_tot, X, Y, Psi, psi_kind, omega, training, validation, patterns, n_patterns))) return bail(rc);
    // moment chunks of the training rows that end at NaN-pattern boundaries: ~n/target rows each (at least min_rows),
    // plus each pattern's range of chunks for the segmented slab sum
    auto build_chunks = [&](int target, int min_rows) -> int {
        int rpc = (c->tr.n + target - 1) / target;
        if (rpc < min_rows) rpc = min_rows;
        std::vector<int> ct, seg((size_t)c->ngroups + 1);
        for (int g = 0; g < c->ngroups; ++g) {
            seg[g] = (int)(ct.size() / 2);
            const int re = c->tr.group_begin[g + 1];
            for (int r = c->tr.group_begin[g]; r < re; r += rpc) {
                ct.push_back(r);
                ct.push_back(r + rpc < re ? r + rpc : re);
            }
        }
        seg[c->ngroups] = c->mom_nchunk = (int)(ct.size() / 2);
        if (ct.empty()) ct.assign(2, 0);
        if (int e = c->ar.alloc(&c->mom_chunktab, ct.size())) return e;
        if (int e = c->ar.alloc(&c->mom_segtab, seg.size())) return e;
        if (hipMemcpy(c->mom_chunktab, ct.data(), ct.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(c->mom_segtab, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return fail(GPZ_ERR_HIP, "copy failed");
        return 0;
    };
    if (c->gen) {
        c->gen_nchunk = 256;
        if (c->psi_miss) {
            if ((rc = build_chunks(256, 1))) return bail(rc);
            if (c->mom_nchunk > c->gen_nchunk) c->gen_nchunk = c->mom_nchunk;     // gen_slab holds one record set per chunk
        }
        const size_t per = (c->psi32 && psi32_raw_len(c->d) > c->nrec) ? (size_t)psi32_raw_len(c->d) : (size_t)c->nrec;
        if ((rc = c->ar.alloc(&c->gen_slab, (size_t)c->gen_nchunk * c->m * per))) return bail(rc);
        if (c->psi32 && (rc = c->ar.alloc(&c->psi32_raw, (size_t)c->m * psi32_raw_len(c->d)))) return bail(rc);
        if ((rc = c->ar.alloc(&c->fin_part, (size_t)c->ngroups * c->m * (c->d + c->d * c->d + 2)))) return bail(rc);
        if (!c->has_psi) {
            const int nmt = c->de + c->de * (c->de + 1) / 2;
            c->gen_tnch = 2048 / ((c->m + 255) / 256);
            if (c->gen_tnch < 1) c->gen_tnch = 1;
            if ((rc = build_chunks(c->gen_tnch, 32))) return bail(rc);
            const size_t nslab = c->mom_nchunk > 0 ? (size_t)c->mom_nchunk : 1;
            if ((rc = c->ar.alloc(&c->gen_tslab, nslab * c->m * (nmt + 2)))) return bail(rc);
            if ((rc = c->ar.alloc(&c->gen_frec, (size_t)c->ngroups * c->m * (nmt + 2)))) return bail(rc);
        }
        if (c->va.n_pad && (rc = c->ar.alloc(&c->Phi_v, (size_t)c->va.n_pad * c->mp))) return bail(rc);
    }
    if ((rc = alloc_mm(c))) return bail(rc);

    const size_t np = c->tr.n_pad, mp = c->mp, k = c->k, m = c->m;
    {
        // resident unless PHI + T (2 n_pad mp doubles) exceed 70 % of the free device memory; GPZ_ROW_TILE=<rows> forces a tile size
        size_t want = c->opt.row_tile > 0 ? (size_t)c->opt.row_tile : 0;
        size_t fr = 0, tot = 0;
        if (!want && hipMemGetInfo(&fr, &tot) == hipSuccess) {   // blocks the buffer cache holds are as good as free (a failed hipMalloc releases them)
            DevCache &dc = dev_cache();
            std::lock_guard<std::mutex> g(dc.mu);
            auto it = dc.held.find(c->device);
            if (it != dc.held.end()) fr += it->second;
        }
        if (!want && fr && 2.0 * (double)np * (double)mp * 8.0 > 0.7 * (double)fr) {
            // the largest tile whose PHI + T take half of that (the slabs and per-row vectors need the rest): every launch of the walk
            // then still fills the chip for many rounds
            want = (size_t)(0.35 * (double)fr / (2.0 * (double)mp * 8.0));
            if (want < 131072) want = 131072;
        }
        const bool can = !c->gen;   // every route of the row kernels (k_phi / k_rows / k_wide): diagonal kinds with or without Psi / NaNs, GC / VC plain
        if (want && can) {
            const size_t tr = (size_t)rup((long)want, 1024);
            if (tr < np) {
                c->tile_rows = (int)tr;
                c->ntiles = (int)((np + tr - 1) / tr);
            }
        }
    }
    const size_t npt = c->tile_rows ? (size_t)c->tile_rows : np;   // rows PHI / T / the nu partials hold
    if ((rc = c->ar.alloc(&c->Phi, npt * mp))) return bail(rc);
    if ((rc = c->ar.alloc(&c->T, npt * mp))) return bail(rc);
    if (c->tile_rows && (rc = c->ar.alloc(&c->tile_rstats, (size_t)c->ntiles * GPZ_NS))) return bail(rc);
    // GC + Psi in fp64, 10 < d <= 32 (evaluation contexts only: prediction and getPHI contexts have no moment stage and no T)
    if (c->psi_fast && c->mid == 4 && cpsi4_available(c->d) && !c->opt.gc_minv_off) {
        // one inverse per training row, shared by the basis functions (k_cpsi4_moments<.., SHARED>)
        if ((rc = c->ar.alloc(&c->gc_minv, (size_t)(c->tr.n > 0 ? c->tr.n : 1) * cpsi4_minv_len(c->d)))) return bail(rc);
        if (!c->psi_miss && !c->opt.gc_dense_phi_off) {   // and, without missing dimensions, the dense form of the PHI build
            const size_t kp = (size_t)gcq_kpad(c->d);
            if ((rc = c->ar.alloc(&c->gcq_A, np * kp))) return bail(rc);
            if ((rc = c->ar.alloc(&c->gcq_B, kp * mp))) return bail(rc);
            if (hipMemset(c->gcq_A, 0, np * kp * sizeof(double)) != hipSuccess) return bail(fail(GPZ_ERR_HIP, "memset failed"));
        }
    }
    c->fused = (k == 1) || !c->gen;   // the general GC/VC path chains r1 / r2 through its records: single output only
    if (!c->fused && (rc = c->ar.alloc(&c->dL, np * mp))) return bail(rc);
    if ((rc = c->ar.alloc(&c->lnbeta, np * k))) return bail(rc);
    if ((rc = c->ar.alloc(&c->wbeta, np * k))) return bail(rc);
    if ((rc = c->ar.alloc(&c->phiw, np * k))) return bail(rc);
    if (c->va.n_pad) {
        if ((rc = c->ar.alloc(&c->lnbeta_v, (size_t)c->va.n_pad * k))) return bail(rc);
        if ((rc = c->ar.alloc(&c->phiw_v, (size_t)c->va.n_pad * k))) return bail(rc);
    }
    // SYRK split over rows: one resident round of 512 workgroups (2 per CU); two rounds once a split still keeps
    // >= 16k rows, where the second round's better tail outweighs the doubled slab traffic (every split writes and
    // the slab sum re-reads mp^2 doubles: measured c2 1.44 -> 1.39 ms, c3 3.49 -> 3.35 ms, 125k-row c4 shard
    // 9.88 -> 9.71 ms with 512; full c4 unchanged with 1024).  GPZ_SYRK_WGS overrides (tuning only).
    {
        const int nt = (c->mp + 127) / 128, npairs = nt * (nt + 1) / 2;
        // rows a workgroup gets at a target of T workgroups: n_pad npairs / T.  Every split is one more mp x mp slab for k_syrk_reduce to
        // read, so short row ranges are not worth a second resident round: c2 / c3 (600 / 2000 rows per workgroup at 512) run
        // SYRK + reduce 19 / 25 us faster at 256 workgroups (0.179 -> 0.160 ms, 0.520 -> 0.495 ms)
        const long np_k = (long)npt;   // rows one SYRK launch sees (a row tile when streaming)
        auto rows_at = [&](int T) { return np_k * npairs / T; };
        int target = rows_at(1024) >= 16384 ? 1024 : rows_at(512) >= 4096 ? 512 : 256;
        if (c->opt.syrk_wgs > 0) target = c->opt.syrk_wgs;   // (developer tuning)
        // Off-diagonal tiles get s1 row ranges, diagonal tiles s2 (their workgroups run 9 MFMAs per SIMD and K step against 16:
        // the 36 products on and above the diagonal, k_gemm.hip): the pair that minimises max(1/s1, 0.6/s2) with
        // noff*s1 + nt*s2 workgroups inside the target.
        const int noff = npairs - nt, max_ns = np_k / 64 > 0 ? (int)(np_k / 64) : 1;
        int s1 = 1, s2 = 1;
        double best = 1e300;
        for (int a = 1; a <= max_ns && a <= target && noff * a + nt <= (target > npairs ? target : npairs); ++a) {
            int b = noff ? (target - noff * a) / nt : a;
            if (b > max_ns) b = max_ns;
            if (b > a) b = a;
            if (b < 1) b = 1;
            const double cost = 1.0 / a > 0.6 / b ? 1.0 / a : 0.6 / b;   // 0.595 measured (tools/syrk_split_sweep.py); 9/16 by MFMA count
            if (cost < best) { best = cost; s1 = a; s2 = b; }
        }
        if (c->opt.syrk_s1 > 0) s1 = c->opt.syrk_s1;   // (developer tuning)
        if (c->opt.syrk_s2 > 0) s2 = c->opt.syrk_s2;
        c->rows_per_split = rup((int)((np_k + s1 - 1) / s1), 16);
        c->nsplit = (int)((np_k + c->rows_per_split - 1) / c->rows_per_split);
        c->rows_per_split_d = rup((int)((np_k + s2 - 1) / s2), 16);
        c->nsplit_d = (int)((np_k + c->rows_per_split_d - 1) / c->rows_per_split_d);
        size_t need = (size_t)(c->nsplit > c->nsplit_d ? c->nsplit : c->nsplit_d) * mp * mp;
        size_t need_l = (size_t)c->nsplit_l * c->mq * c->mq;
        c->slab_count = need > need_l ? need : need_l;
        if ((rc = c->ar.alloc(&c->slab, c->slab_count))) return bail(rc);
    }
    if (npt < 1024 * 1024) {   // (indexed by row with the full row stride, also when the rows are streamed)
        c->phipart_groups = 16;
        size_t rows = np > (size_t)c->va.n_pad ? np : (size_t)c->va.n_pad;
        if ((rc = c->ar.alloc(&c->phipart, (size_t)c->phipart_groups * 2 * k * rows))) return bail(rc);
    }
    c->comm1_count = k * mp * mp + gpz_ns(c->k);
    if ((rc = c->ar.alloc(&c->comm1, c->comm1_count))) return bail(rc);
    c->comm2_count = m * c->nm + k * 2 * mp + k * 4 + gpz_ns(c->k);
    if ((rc = c->ar.alloc(&c->comm2, c->comm2_count))) return bail(rc);
    c->nwg_rows = c->tr.n < 2048 ? (c->tr.n > 0 ? c->tr.n : 1) : 2048;
    if (!c->fused) {
        if ((rc = c->ar.alloc(&c->colslab, (size_t)c->nwg_rows * 2 * mp))) return bail(rc);
        if ((rc = c->ar.alloc(&c->scal_slab, (size_t)c->nwg_rows * 4))) return bail(rc);
    } else {
        c->nslots = gpz_gemm_wave_cols() * ((c->mp + 127) / 128);
        if ((rc = c->ar.alloc(&c->nupart, (size_t)c->nslots * npt))) return bail(rc);
        if ((rc = c->ar.alloc(&c->rowscal, (size_t)4 * np))) return bail(rc);
        if ((rc = c->ar.alloc(&c->frec, (size_t)m * (c->nm + 2)))) return bail(rc);
    }
    {
        const int ncg = (c->m + 255) / 256;
        // chunks of rows per basis-function group: every chunk writes (and k_slab_sum re-reads) m x nm sums, so few enough that the
        // slab stays small beside Phi and T, many enough to fill 256 CUs (tools/mom_nc_sweep.sh: c2 768, c3/c4 512 chunks)
        const int nc_env = c->opt.mom_nc;   // (developer tuning)
        int nc = nc_env > 0 ? nc_env / ncg : (768 / ncg > 512 ? 768 / ncg : (2048 / ncg < 512 ? 2048 / ncg : 512));
        const int max_nc = c->tr.n / 32 > 0 ? c->tr.n / 32 : 1;
        if (nc > max_nc) nc = max_nc;
        if (nc < 1) nc = 1;
        if (c->gen) nc = 1;   // the general path has its own slabs
        c->rows_per_chunk = (c->tr.n + nc - 1) / nc;
        if (c->rows_per_chunk < 1) c->rows_per_chunk = 1;
        c->nchunk = c->tr.n > 0 ? (c->tr.n + c->rows_per_chunk - 1) / c->rows_per_chunk : 1;
        if (c->tile_rows) {   // every tile gets its own run of chunks (about the same number in all)
            int nct = nc;   // as many as a resident evaluation uses for all rows: a tile's launch fills the chip the same way
            if (nct > c->tile_rows / 32) nct = c->tile_rows / 32 > 0 ? c->tile_rows / 32 : 1;
            c->tile_rpc = (c->tile_rows + nct - 1) / nct;
            c->tile_nchunk = (c->tile_rows + c->tile_rpc - 1) / c->tile_rpc;
            c->nchunk = c->ntiles * c->tile_nchunk;
        }
        if ((rc = c->ar.alloc(&c->mom_slab, (size_t)c->nchunk * m * (c->nm + 2)))) return bail(rc);
    }
    if ((rc = c->ar.alloc(&c->partial, (size_t)GPZ_ROWSCAL_MAX_NWG * gpz_ns(c->k)))) return bail(rc);
    static_assert(GPZ_ROWSCAL_MAX_NWG >= GPZ_SMALL_NWG, "partial record buffer");
    if ((rc = c->ar.alloc(&c->rstats, (size_t)gpz_ns(c->k)))) return bail(rc);
    if ((rc = c->ar.alloc(&c->spart, (size_t)(c->k > 8 ? c->k : 8)))) return bail(rc);
    if ((rc = c->ar.alloc(&c->dGfull, c->kind == GPZ_KIND_COV ? m * c->d * c->d : m * c->d))) return bail(rc);
    if ((rc = c->ar.alloc(&c->out_d, (size_t)c->p + 10))) return bail(rc);
    if (hipHostMalloc((void **)&c->out_h, ((size_t)c->p + 10) * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void **)&c->theta_h, (size_t)c->p * sizeof(double)) != hipSuccess)
        return bail(fail(GPZ_ERR_ALLOC, "hipHostMalloc failed"));
    *out = c;
    return GPZ_OK;
}

extern "C" void gpz_ctx_destroy(gpz_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    if (c->priv && c->priv_free) c->priv_free(c->priv);
    c->ar.release();
    if (c->out_h) (void)hipHostFree(c->out_h);
    if (c->theta_h) (void)hipHostFree(c->theta_h);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph_st) (void)hipStreamDestroy(c->graph_st);
    for (hipEvent_t e : c->tm.pool) (void)hipEventDestroy(e);
    delete c;
}

void gpz_ctx_attach_private(gpz_ctx *c, void *priv, void (*free_fn)(void *)) {
    if (c->priv && c->priv_free) c->priv_free(c->priv);
    c->priv = priv;
    c->priv_free = free_fn;
}

extern "C" int gpz_ctx_set_allreduce(gpz_ctx *c, gpz_allreduce_fn fn, void *user) {
    if (!c) return fail(GPZ_ERR_ARG, "null context");
    c->ar_fn = fn;
    c->ar_user = user;
    return GPZ_OK;
}
extern "C" int64_t gpz_theta_len(const gpz_ctx *c) { return c ? c->p : -1; }
extern "C" int64_t gpz_theta_len_of(const gpz_desc *ds) {
    if (!ds || ds->d < 1 || ds->m < 1 || ds->k < 1) return -1;
    const int mid = method_id_of(ds->method);
    if (mid < 0) return -1;
    return (int64_t)ds->m * ds->d + g_dim_of(mid, ds->m, ds->d) + (int64_t)ds->m * ds->k + ds->k + (ds->heteroscedastic ? 2LL * ds->m * ds->k : 0);
}
extern "C" int64_t gpz_n_train(const gpz_ctx *c) { return c ? c->tr.n : -1; }
extern "C" int64_t gpz_n_valid(const gpz_ctx *c) { return c ? c->va.n : -1; }

extern "C" int gpz_ctx_set_pinv_mode(gpz_ctx *c, int mode) {
    if (!c || mode < -1 || mode > 1) return fail(GPZ_ERR_ARG, "gpz_ctx_set_pinv_mode: mode must be -1, 0 or 1");
    c->pinv_mode = mode;
    return GPZ_OK;
}
extern "C" int gpz_ctx_last_pinv(const gpz_ctx *c, double out[4]) {
    if (!c || !out) return fail(GPZ_ERR_ARG, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = c->pinv_last[i];
    return GPZ_OK;
}

extern "C" int gpz_ctx_enable_timing(gpz_ctx *c, int enable) {
    if (!c) return fail(GPZ_ERR_ARG, "null context");
    c->timing = enable != 0;
    return GPZ_OK;
}
extern "C" int gpz_ctx_reset_timings(gpz_ctx *c) {
    if (!c) return fail(GPZ_ERR_ARG, "null context");
    for (auto &v : c->tm.ms) v = 0.0;
    for (auto &v : c->tm.calls) v = 0;
    return GPZ_OK;
}
extern "C" int gpz_ctx_timings(gpz_ctx *c, const char **names, double *ms, int64_t *calls, int cap) {
    if (!c) return fail(GPZ_ERR_ARG, "null context");
    const int n = (int)c->tm.names.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = c->tm.names[i];
        if (ms) ms[i] = c->tm.ms[i];
        if (calls) calls[i] = c->tm.calls[i];
    }
    return n;
}

// Which kernels this context runs, in words, and where the evaluation graph stands: a caller that asked for dtype = f32 learns
// here whether its rows actually take the fp32 pair kernels (input noise, no missing dimension, d <= 20) or the fp64 ones, and a
// bench line can show a graph capture that failed instead of silently timing eager launches.
extern "C" int gpz_ctx_route(const gpz_ctx *c, char *buf, int cap) {
    if (!c || !buf || cap <= 0) return fail(GPZ_ERR_ARG, "gpz_ctx_route: null argument");
    const bool f32req = c->desc.dtype == GPZ_F32;
    const char *phi, *why = "";
    if (!c->gen) phi = c->d > 20 ? "tuned diagonal / covariance kernels, runtime-d form (k_wide)" : "tuned kernels (k_phi, k_rows)";
    else if (!c->has_psi) phi = "covariance kinds with missing dimensions: tuned kernels per NaN pattern";
    else if (c->psi32) phi = (c->tr.psi_diag && psi32m_available(c->d))
                                 ? "fp32 pair kernels: k_psi32_phi + k_psi32m_moments (4x4 MFMA tiles)" : "fp32 pair kernels (k_psi32)";
    else if (c->psi_fast) phi = c->d <= 10 ? "fp64 pair kernels in registers (k_psi)"
                                : cpsi4_available(c->d) ? "fp64 pair kernels on 4x4 f64 MFMA tiles (k_cpsi4)" : "fp64 pair kernels on MFMA tiles (k_cpsi4w / k_cpsi)";
    else phi = "fp64 pair kernels, workspace form (k_gen)";
    if (f32req && !c->psi32)
        why = !c->gen || !c->has_psi ? " [dtype f32 requested: no input noise on a covariance kind, nothing runs in fp32]"
              : c->d > 20            ? " [dtype f32 requested: d > 20 has no fp32 pair kernel, fp64 route]"
                                     : " [dtype f32 requested: rows with missing dimensions, fp64 route]";
    const char *gs = c->graph_state == 2 ? "replayed" : c->graph_state == -1 ? "disabled (capture failed or GPZ_NO_GRAPH)"
                     : c->timing ? "off (stage timing on)" : c->desc.world > 1 ? "off (sharded)" : "eager (not captured yet)";
    const bool f32mm = c->psi32 && !c->opt.f32_contractions_off;
    char rows[96];
    if (c->tile_rows) snprintf(rows, sizeof rows, "; rows: streamed, %d tiles of %d (PHI built twice per evaluation)", c->ntiles, c->tile_rows);
    else rows[0] = 0;
    return snprintf(buf, (size_t)cap, "pair/PHI kernels: %s%s; contractions: %s MFMA; evaluation graph: %s%s", phi, why,
                    f32mm ? "fp32-operand (fp64 master sums)" : "fp64", gs, rows);
}

// ---- pipeline stages -----------------------------------------------------------------------------
static GenRows gen_rows(const RowSet &rs) {
    GenRows r{};
    r.Xr = rs.Xr; r.gid = rs.gid; r.Psi3 = rs.Psi3; r.rows_by_group = rs.rows_by_group; r.n = rs.n; r.n_pad = rs.n_pad;
    return r;
}

// row chunks of the fp32 moment kernel: whole 64-row wave blocks, at most gen_nchunk chunks
static void psi32_chunks(const gpz_ctx *c, int *nch, int *rpc) {
    int r = (c->tr.n + c->gen_nchunk - 1) / c->gen_nchunk;
    r = rup(r > 0 ? r : 1, 64);
    *rpc = r;
    *nch = c->tr.n > 0 ? (c->tr.n + r - 1) / r : 1;
}

static int allreduce(gpz_ctx *c, double *buf, size_t count) {
    if (c->desc.world <= 1) return 0;
    if (!c->ar_fn) return fail(GPZ_ERR_COMM, "world=%d but no all-reduce hook set (gpz_ctx_set_allreduce)", c->desc.world);
    if (c->ar_fn(c->ar_user, buf, count, (void *)c->st) != 0) return fail(GPZ_ERR_COMM, "all-reduce hook failed");
    return 0;
}

// Sharded fp32 runs: the diagonal-Psi kernels leave WHITENED moment records, the full-Psi kernels plain ones, and the
// records are summed over ranks — so every rank must run the same form.  A rank whose own rows are all diagonal
// switches to the full form (its diagonals expanded to packed triangles on the device) when any other rank needs it.
static int psi32_agree(gpz_ctx *c) {
    if (!c->psi32 || c->psi32_agreed || c->desc.world <= 1) return 0;
    const double mine = (c->tr.psi_diag && (c->va.n_pad == 0 || c->va.psi_diag)) ? 0.0 : 1.0;
    HIPCHK(hipMemcpyAsync(c->rstats, &mine, sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (int e = allreduce(c, c->rstats, 1)) return e;
    double total = 0.0;
    HIPCHK(hipMemcpyAsync(&total, c->rstats, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (total > 0.0) {
        const int D = psi32_pad_dim(c->d);
        for (RowSet *rs : {&c->tr, &c->va}) {
            if (!rs->PsiT || !rs->psi_diag) continue;
            float *full = nullptr;
            const size_t np = (size_t)rs->n_pad;
            if (int e = c->ar.alloc(&full, (size_t)D * (D + 1) / 2 * np)) return e;
            HIPCHK(hipMemsetAsync(full, 0, (size_t)D * (D + 1) / 2 * np * sizeof(float), c->st));
            for (int a = 0; a < D; ++a)   // diagonal a -> packed element (a, a)
                HIPCHK(hipMemcpyAsync(full + ((size_t)a * (a + 1) / 2 + a) * np, rs->PsiT + (size_t)a * np, np * sizeof(float),
                                      hipMemcpyDeviceToDevice, c->st));
            rs->PsiT = full;
            rs->psi_diag = 0;
        }
    }
    c->psi32_agreed = true;
    return 0;
}

// GC/VC with missing dimensions and no input noise: the rows of every NaN pattern (stored contiguously) go through the
// tuned PHI kernel with that pattern's parameter block (k_gen_pattern_params).  Launches run in row order on one
// stream: a launch zero-fills up to the end of its last 1024-row block, the next pattern's launch rewrites those rows.
static int phi_by_pattern(gpz_ctx *c, RowSet &rs, double *Phi, double *lnbeta, double *wbeta, const double *w, double *phiw,
                          bool with_y) {
    const size_t np = (size_t)rs.n_pad, mp = (size_t)c->mp;
    const int de = c->de;
    const size_t tail = np - (size_t)rs.n;   // rows past the data: zero (the slack block is never written otherwise)
    if (Phi) HIPCHK(hipMemsetAsync(Phi + (size_t)rs.n * mp, 0, tail * mp * sizeof(double), c->st));
    for (int o = 0; o < c->k; ++o) {
        HIPCHK(hipMemsetAsync(lnbeta + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
        if (wbeta) HIPCHK(hipMemsetAsync(wbeta + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
        if (phiw) HIPCHK(hipMemsetAsync(phiw + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
    }
    if (!rs.nwg_tab) return 0;
    // one launch over all patterns: every workgroup looks up its row range and its pattern's parameter block
    PhiArgs a{};
    a.Xc = rs.Xc; a.ldx = (long)np; a.n = rs.n; a.n_pad = rs.n_pad;
    a.m = c->m; a.mp = c->mp; a.d = de; a.k = c->k; a.kind = GPZ_KIND_COV;
    a.P = c->pr.P; a.G = c->RcP;
    a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b;
    a.omega = rs.om;
    a.Y = (with_y && rs.Y) ? rs.Y : nullptr;
    a.Phi = Phi;
    a.lnbeta = lnbeta; a.wbeta = wbeta;
    a.w = w; a.phiw = phiw;
    a.part = c->phipart; a.part_groups = c->phipart_groups;      // few workgroups: split the basis functions as well
    a.wgtab = rs.wgtab; a.nwg_tab = rs.nwg_tab;
    if (launch_phi(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", de);
    return 0;
}

// dP/dGamma moment records of every pattern through the tuned moment kernels (fused: single output, dPHI formed on
// the fly; plain: dPHI already in T), converted to the records k_gen_finish chains (k_gen_convert_moments).
static int moments_by_pattern(gpz_ctx *c, bool fused, double *mom) {
    const int de = c->de, nmt = de + de * (de + 1) / 2, stride = fused ? nmt + 2 : nmt;
    const size_t m = (size_t)c->m;
    if (c->mom_nchunk > 0) {   // one launch: the chunk table keeps every chunk inside one pattern's rows
        if (fused) {
            FusedMomentArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.rowscal = c->rowscal;
            a.n = c->tr.n; a.m = c->m; a.d = de; a.kind = GPZ_KIND_COV; a.P = c->pr.P; a.w = c->w;
            a.v = c->hetero ? c->pr.v : nullptr; a.nchunk = c->mom_nchunk; a.rows_per_chunk = 0; a.slab = c->gen_tslab;
            a.nm = nmt; a.chunktab = c->mom_chunktab;
            if (launch_moments_fused(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", de);
        } else {
            MomentArgs a{};
            a.dPhi = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
            a.m = c->m; a.d = de; a.kind = GPZ_KIND_COV; a.P = c->pr.P; a.nchunk = c->mom_nchunk; a.rows_per_chunk = 0;
            a.slab = c->gen_tslab; a.nm = nmt; a.chunktab = c->mom_chunktab;
            if (launch_moments(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", de);
        }
    }
    launch_slab_sum_seg(c->st, c->gen_tslab, c->mom_segtab, c->ngroups, m * stride, c->gen_frec);
    launch_gen_convert_moments(c->st, c->gen_frec, stride, fused ? 1 : 0, c->Sig, c->pat_d, c->ngroups, c->m, c->d, de, mom,
                               c->nrec);
    return 0;
}

// PHI, ln beta and omega*beta of the training row set from the unpacked parameters (getPHI.m:60-125, GPz.m:43-48).
static int build_phi(gpz_ctx *c) {
    if (c->gen) {
        Stage s(c, "phi_build");
        // Sigma_j / inv(Sigma_j) / ln|Sigma_j|: everything except the whitened fp32 route (which works from the QR factor)
        const bool whitened = c->psi32 && c->tr.psi_diag && (c->va.n_pad == 0 || c->va.psi_diag);
        if (!whitened)
            launch_gen_prep(c->st, c->pr.G, c->m, c->d, c->de, c->Sig, c->iSig, c->pat_d, c->ngroups, c->lnS, c->gen_ws);
        if (!c->has_psi) {   // missing dimensions only: tuned kernels, one launch per NaN pattern
            launch_gen_pattern_params(c->st, c->Sig, c->pr.P, c->pat_d, c->ngroups, c->m, c->d, c->de, c->RcP, c->gen_ws);
            return phi_by_pattern(c, c->tr, c->Phi, c->lnbeta, c->wbeta, nullptr, nullptr, true);
        }
        if (c->psi32) {
            launch_psi32_phi(c->st, c->tr.Xr, c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m,
                             c->pr.P, c->Sig, c->pr.Rc, c->lnS, c->Phi, c->mp);
            launch_gen_fill(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->mp, c->k, c->tr.Y);
        } else if (c->psi_fast) {
            if (c->gc_minv)   // GC: Sigma + Psi_i inverted once per row - for the moment kernel, and for the dense form of the PHI build
                launch_cpsi4_minv(c->st, gen_rows(c->tr), c->d, c->de, c->Sig, c->lnS, c->psi_miss ? c->pat_d : nullptr, c->gc_minv,
                                  c->gcq_A, gcq_kpad(c->d));
            if (c->gcq_A) {
                // ln PHI = -1/2 [c_ab M^-1_ab | M^-1 x | x'M^-1 x + ln|M| - ln|Sigma|] . [p_a p_b ; -2 p ; 1]: one product on the T-GEMM kernel
                // (c->T is free until the evaluation's own T-GEMM) and an exp
                launch_gcq_tab(c->st, c->m, c->d, c->de, c->mp, c->pr.P, c->gcq_B);
                launch_tgemm(c->st, c->gcq_A, gcq_kpad(c->d), c->gcq_B, c->mp, c->T, c->tr.n_pad, c->mp, nullptr, nullptr, c->m, -1, false,
                             gcq_kpad(c->d), c->mp);
                launch_gcq_exp(c->st, c->T, c->mp, c->tr.n, c->m, c->Phi);
            } else
                launch_psi_phi(c->st, gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, c->lnS, c->Phi, c->mp,
                               c->psi_miss ? c->pat_d : nullptr, c->mid == 4);
            launch_gen_fill(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->mp, c->k, c->tr.Y);
        } else {
            launch_gen_phi(c->st, gen_rows(c->tr), c->m, c->mp, c->d, c->de, c->k, c->pr.P, c->Sig, c->lnS, c->pat_d,
                           c->Phi, c->tr.Y, c->gen_ws);
        }
        launch_gen_rowdot(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          c->tr.om, nullptr, c->lnbeta, c->wbeta, nullptr);
    } else {
        Stage s(c, "phi_build");
        PhiArgs a{};
        a.Xc = c->tr.Xc; a.ldx = c->tr.n_pad; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om; a.Y = c->tr.Y;
        a.Phi = c->Phi; a.lnbeta = c->lnbeta; a.wbeta = c->wbeta; a.w = nullptr; a.phiw = nullptr;
        a.Psic = c->tr.Psic; a.Mc = c->tr.Mc; a.ucnt = c->tr.ucnt;
        a.part = c->phipart; a.part_groups = c->phipart_groups;
        if (launch_phi(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
    }
    return 0;
}

// Row-tile streaming: rows [r0, r0 + rows_pad) of the training set (rows of them real) as the current contents of c->Phi.
struct RowTile { long r0; int rows, rows_pad; };
static RowTile row_tile(const gpz_ctx *c, int t) {
    RowTile rt;
    rt.r0 = (long)t * c->tile_rows;
    const long left_pad = (long)c->tr.n_pad - rt.r0, left = (long)c->tr.n - rt.r0;
    rt.rows_pad = (int)(left_pad < c->tile_rows ? left_pad : c->tile_rows);
    rt.rows = (int)(left < 0 ? 0 : (left < rt.rows_pad ? left : rt.rows_pad));
    return rt;
}
static int phi_tile(gpz_ctx *c, const RowTile &rt) {
    PhiArgs a{};
    const long r0 = rt.r0;
    a.Xc = c->tr.Xc + r0; a.ldx = c->tr.n_pad; a.n = rt.rows; a.n_pad = rt.rows_pad;
    a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
    a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
    a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om ? c->tr.om + r0 : nullptr; a.Y = c->tr.Y + r0;
    a.Phi = c->Phi; a.lnbeta = c->lnbeta + r0; a.wbeta = c->wbeta + r0; a.w = nullptr; a.phiw = nullptr;
    a.Psic = c->tr.Psic ? c->tr.Psic + r0 : nullptr; a.Mc = c->tr.Mc ? c->tr.Mc + r0 : nullptr;
    a.ucnt = c->tr.ucnt ? c->tr.ucnt + r0 : nullptr;
    a.part = c->phipart; a.part_groups = c->phipart_groups;   // (row-indexed with stride ldx: the tile's rows from its base)
    if (launch_phi(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
    return 0;
}
// Stage A of a streamed evaluation: per tile PHI -> PHI' W_o PHI, summed over the tiles in comm1.
static int stage_a_tiles(gpz_ctx *c) {
    const bool f32 = c->psi32 && !c->opt.f32_contractions_off;
    for (int t = 0; t < c->ntiles; ++t) {
        const RowTile rt = row_tile(c, t);
        {
            Stage s(c, "phi_build");
            if (int e = phi_tile(c, rt)) return e;
        }
        const int nsp = (rt.rows_pad + c->rows_per_split - 1) / c->rows_per_split;
        const int nsp_d = (rt.rows_pad + c->rows_per_split_d - 1) / c->rows_per_split_d;
        for (int o = 0; o < c->k; ++o) {
            {
                Stage s(c, "syrk");
                launch_syrk(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad + rt.r0, rt.rows_pad, c->mp, nsp, c->rows_per_split,
                            nsp_d, c->rows_per_split_d, c->slab, false, f32);
            }
            Stage s(c, "syrk_reduce");
            launch_syrk_reduce(c->st, c->slab, nsp, nsp_d, c->mp, c->comm1 + (size_t)o * c->mp * c->mp, c->mp, t > 0 ? 1 : 0);
        }
    }
    return 0;
}

// Stage A: theta -> PHI, ln beta, omega*beta, S_o = PHI' W_o PHI (incl. PHI' W_o y), sums; all-reduce #1.
static int stage_a(gpz_ctx *c, const double *theta, const double *theta_dev = nullptr) {
    if (theta_dev) {   // device-resident caller (gpz_eval_dev): theta never visits the host
        HIPCHK(hipMemcpyAsync(c->theta_d, theta_dev, (size_t)c->p * sizeof(double), hipMemcpyDeviceToDevice, c->st));
    } else {
        memcpy(c->theta_h, theta, (size_t)c->p * sizeof(double));
        HIPCHK(hipMemcpyAsync(c->theta_d, c->theta_h, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st));
    }
    {
        Stage s(c, "unpack");
        // also clears info[0..1] and, without validation rows, the validation sums of the result block (eval_tail's layout of comm2)
        double *vsums0 = c->va.n_pad > 0 ? nullptr : c->comm2 + (size_t)c->m * c->nm + (size_t)c->k * 2 * c->mp + (size_t)c->k * 4;
        launch_unpack(c->st, c->theta_d, c->mid, c->m, c->d, c->de, c->k, c->hetero, c->pr, c->info, vsums0, gpz_ns(c->k));
        if (c->kind == GPZ_KIND_COV) launch_prep_cov(c->st, c->pr.G, c->pr.P, c->m, c->de, c->pr.Rc, c->prep_ws);
    }
    if (int e = psi32_agree(c)) return e;
    if (c->tile_rows) { if (int e = stage_a_tiles(c)) return e; }
    else if (int e = build_phi(c)) return e;
    double *sums1 = c->comm1 + (size_t)c->k * c->mp * c->mp;
    {
        Stage s(c, "row_sums");
        launch_sums1(c->st, c->tr.om, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), sums1);
    }
    for (int o = 0; o < c->k && !c->tile_rows; ++o) {
        {
            Stage s(c, "syrk");
            launch_syrk(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad, c->tr.n_pad, c->mp, c->nsplit,
                        c->rows_per_split, c->nsplit_d, c->rows_per_split_d, c->slab, false,
                        c->psi32 && !c->opt.f32_contractions_off);   // config 5: fp32-operand MFMAs, fp64 master sums
        }
        {
            Stage s(c, "syrk_reduce");
            launch_syrk_reduce(c->st, c->slab, c->nsplit, c->nsplit_d, c->mp, c->comm1 + (size_t)o * c->mp * c->mp, c->mp);
        }
    }
    {
        Stage s(c, "allreduce1");
        if (int e = allreduce(c, c->comm1, c->comm1_count)) return e;
    }
    c->phi_valid = !c->tile_rows;   // streamed: c->Phi holds the last tile only
    return 0;
}

// Stage B for one output: inv(SIGMA_o), logdet_o, w_o, dwda_o, diag; Bext = [inv | w].
static void stage_b(gpz_ctx *c, int o) {
    const int mq = c->mq, m = c->m;
    const double *S = c->comm1 + (size_t)o * c->mp * c->mp;
    {
        Stage s(c, "chol");
        launch_build_sigma(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, m, mq, c->A, mq, c->Wm, c->logdet + o);   // clears Wm, logdet too
        for (int k0 = 0; k0 < mq; k0 += GPZ_CH_NB) {
            launch_chol_step(c->st, c->A, c->Lm, mq, mq, k0, c->logdet + o, c->info);
        }
    }
    {
        Stage s(c, "trtri");
        launch_trtri_diag(c->st, c->Lm, c->Wm, mq, mq);
        for (int gs = GPZ_CH_NB; gs < mq; gs *= 2) launch_trtri_level(c->st, c->Lm, c->Wm, c->Tmp, mq, mq, gs);
    }
    {
        Stage s(c, "lauum");
        launch_syrk(c->st, c->Wm, mq, nullptr, mq, mq, c->nsplit_l, c->rows_per_split_l, c->nsplit_l, c->rows_per_split_l, c->slab,
                    true);
        launch_syrk_reduce(c->st, c->slab, c->nsplit_l, c->nsplit_l, mq, c->Sinv, mq);
    }
    {
        Stage s(c, "solve_vectors");
        launch_post_inverse(c->st, c->Sinv, mq, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->mp, o, c->Bext,
                            c->w + (size_t)o * m, c->dwda + (size_t)o * m, c->dgi + (size_t)o * m, c->info, c->logdet);
        if (c->pinv_mode == 0) launch_cond_flag(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, c->Sinv, mq, m, c->Tmp, c->info);
    }
}

// Stage B through the rank-truncating SVD pseudo-inverse (inv_logdet.m:3-15) instead of the Cholesky inverse.
static int stage_b_pinv(gpz_ctx *c, int o) {
    const int mq = c->mq, m = c->m;
    const double *S = c->comm1 + (size_t)o * c->mp * c->mp;
    Stage s(c, "pinv_svd");
    double *sbuf = c->Tmp, *out3 = c->Tmp + mq + 8;
    unsigned long long *word = (unsigned long long *)(c->Tmp + mq);
    const int sweeps = run_jacobi_pinv(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->A, c->Wm, mq, sbuf, word, c->Sinv,
                                       mq, c->logdet + o, out3);
    if (sweeps < 0) return fail(GPZ_ERR_HIP, "pseudo-inverse (Jacobi SVD) failed: %s", hipGetErrorString(hipGetLastError()));
    double h3[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(h3, out3, sizeof h3, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    c->pinv_last[0] = 1.0;
    c->pinv_last[1] = (o == 0) ? h3[1] : fmin(c->pinv_last[1], h3[1]);
    c->pinv_last[2] = h3[2];
    c->pinv_last[3] = (double)sweeps;
    launch_post_inverse(c->st, c->Sinv, mq, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->mp, o, c->Bext,
                        c->w + (size_t)o * m, c->dwda + (size_t)o * m, c->dgi + (size_t)o * m, c->info, c->logdet);
    return 0;
}

// Everything of an evaluation after stage A (SIGMA partials reduced): solve, T-GEMM, row epilogue, moments, validation,
// all-reduce #2, finish, result copy.  pinv selects the inverse: Cholesky (false) or truncating SVD (true).
static int eval_tail(gpz_ctx *c, bool pinv) {
    const size_t mp = c->mp, m = c->m, k = c->k;
    double *mom = c->comm2;
    double *cols = mom + m * c->nm;
    double *scal = cols + k * 2 * mp;
    double *vsums = scal + k * 4;
    const bool fused = c->fused;
    for (int o = 0; o < c->k; ++o) {
        if (pinv) { if (int e = stage_b_pinv(c, o)) return e; }
        else stage_b(c, o);
        if (c->tile_rows) {
            // streamed: per tile PHI again -> T -> row scalars -> moment sums into the tile's own chunks; the sums over the tiles after the walk
            const size_t oo = (size_t)o * c->tr.n_pad;
            for (int t = 0; t < c->ntiles; ++t) {
                const RowTile rt = row_tile(c, t);
                const long r0 = rt.r0;
                { Stage s(c, "phi_build"); if (int e = phi_tile(c, rt)) return e; }
                {
                    Stage s(c, "tgemm");
                    launch_tgemm(c->st, c->Phi, c->mp, c->Bext, c->mp, c->T, rt.rows_pad, c->mp, c->nupart, c->phiw + oo + r0, c->m, c->m + o,
                                 c->psi32 && !c->opt.f32_contractions_off);
                }
                {
                    Stage s(c, "row_scalars");
                    launch_row_scalars(c->st, c->nupart, c->nslots, c->phiw + oo + r0, c->tr.Y + oo + r0, c->tr.om ? c->tr.om + r0 : nullptr,
                                       c->lnbeta + oo + r0, c->wbeta + oo + r0, rt.rows_pad, rt.rows, c->rowscal + 4 * r0, c->partial);
                    launch_slab_sum(c->st, c->partial, row_scalars_nwg(rt.rows), GPZ_NS, c->tile_rstats + (size_t)t * GPZ_NS);
                }
                Stage s(c, "moments");
                FusedMomentArgs a{};
                a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr + r0 * c->de; a.rowscal = c->rowscal + 4 * r0; a.n = rt.rows;
                a.m = c->m; a.d = c->de; a.kind = c->kind; a.P = c->pr.P; a.w = c->w + (size_t)o * m;
                a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
                a.rows_per_chunk = c->tile_rpc; a.nchunk = (rt.rows + c->tile_rpc - 1) / c->tile_rpc;
                a.slab = c->mom_slab + (size_t)t * c->tile_nchunk * m * (c->nm + 2); a.nm = c->nm;
                a.Psir = c->tr.Psir ? c->tr.Psir + r0 * c->de : nullptr; a.Mr = c->tr.Mr ? c->tr.Mr + r0 * c->de : nullptr; a.G2 = c->pr.G2;
                if (a.nchunk > 0 && launch_moments_fused(c->st, a))
                    return fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
            }
            Stage s(c, "moments");
            launch_slab_sum(c->st, c->tile_rstats, c->ntiles, GPZ_NS, c->rstats);
            HIPCHK(hipMemcpyAsync(scal + (size_t)o * 4, c->rstats, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->st));
            const RowTile last = row_tile(c, c->ntiles - 1);
            const int nch = (c->ntiles - 1) * c->tile_nchunk + (last.rows + c->tile_rpc - 1) / c->tile_rpc;   // the last tile's chunks end the slab
            launch_slab_sum(c->st, c->mom_slab, nch, m * (c->nm + 2), c->frec);
            launch_split_fused(c->st, c->frec, c->m, c->nm, c->mp, mom, cols + (size_t)o * 2 * mp, o > 0 ? 1 : 0);
            continue;
        }
        {
            Stage s(c, "tgemm");
            // dtype f32 with the fp32 pair kernels active (config 5): fp32-operand MFMA contractions (k_gemm.hip)
            launch_tgemm(c->st, c->Phi, c->mp, c->Bext, c->mp, c->T, c->tr.n_pad, c->mp, fused ? c->nupart : nullptr,
                         fused ? c->phiw + (size_t)o * c->tr.n_pad : c->phiw, c->m, c->m + o,
                         c->psi32 && !c->opt.f32_contractions_off);
        }
        if (fused) {
            {
                Stage s(c, "row_scalars");
                const size_t oo = (size_t)o * c->tr.n_pad;   // this output's columns of the k x n_pad row arrays
                launch_row_scalars(c->st, c->nupart, c->nslots, c->phiw + oo, c->tr.Y + oo, c->tr.om, c->lnbeta + oo,
                                   c->wbeta + oo, c->tr.n_pad, c->tr.n, c->rowscal, c->partial);
                launch_slab_sum(c->st, c->partial, row_scalars_nwg(c->tr.n), GPZ_NS, c->rstats);
                HIPCHK(hipMemcpyAsync(scal + (size_t)o * 4, c->rstats, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->st));
            }
            Stage s(c, "moments");
            if (c->gen && !c->has_psi) {
                if (int e = moments_by_pattern(c, true, mom)) return e;
                continue;
            }
            if (c->gen && c->psi32) {
                int nch, rpc;
                psi32_chunks(c, &nch, &rpc);
                if (c->tr.psi_diag && psi32m_available(c->d))   // diagonal Psi: the 4 x 4-tile MFMA form (k_psi32m.hip)
                    launch_psi32m_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, c->tr.Xr,
                                          c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.n, c->m, c->pr.P, c->pr.Rc, nch, rpc,
                                          c->gen_slab);
                else
                    launch_psi32_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, c->tr.Xr,
                                         c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m, c->pr.P, c->Sig,
                                         c->pr.Rc, nch, rpc, c->gen_slab, c->nrec);
                launch_slab_sum(c->st, c->gen_slab, nch, m * psi32_raw_len(c->d), c->psi32_raw);
                launch_psi32_records(c->st, c->psi32_raw, c->d, c->tr.psi_diag, c->m, mom, c->nrec);
                continue;
            }
            if (c->gen && c->psi_fast) {
                int nch = c->gen_nchunk;
                if (nch > c->tr.n) nch = c->tr.n;
                if (nch < 1) nch = 1;   // a rank without training rows still writes its (zero) records
                const int rpc = (c->tr.n + nch - 1) / nch > 0 ? (c->tr.n + nch - 1) / nch : 1;
                nch = (c->tr.n + rpc - 1) / rpc;
                if (nch < 1) nch = 1;
                if (c->psi_miss) {   // one launch over all NaN patterns, one record set per pattern
                    launch_psi_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr,
                                       gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, c->mom_nchunk, 0, c->gen_slab,
                                       c->nrec, c->pat_d, c->mom_chunktab, c->gc_minv);
                    launch_slab_sum_seg(c->st, c->gen_slab, c->mom_segtab, c->ngroups, m * c->nrec, mom);
                    continue;
                }
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr,
                                   gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec,
                                   nullptr, nullptr, c->gc_minv);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, mom);
                continue;
            }
            if (c->gen) {
                const GenRows gr = gen_rows(c->tr);
                for (int g = 0; g < c->ngroups; ++g) {
                    const int rb = c->tr.group_begin[g], nr = c->tr.group_begin[g + 1] - rb;
                    double *recs_g = mom + (size_t)g * m * c->nrec;
                    if (nr <= 0) { launch_zero(c->st, recs_g, m * c->nrec); continue; }
                    int nch = c->gen_nchunk;
                    if (nch > nr) nch = nr;
                    const int rpc = (nr + nch - 1) / nch;
                    nch = (nr + rpc - 1) / rpc;
                    launch_gen_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, gr, g, rb,
                                       nr, c->pat_d, c->m, c->d, c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, c->gen_ws);
                    launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, recs_g);
                }
                continue;
            }
            FusedMomentArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.rowscal = c->rowscal; a.n = c->tr.n; a.m = c->m;
            a.d = c->de; a.kind = c->kind; a.P = c->pr.P; a.w = c->w + (size_t)o * m;
            a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
            a.nchunk = c->nchunk; a.rows_per_chunk = c->rows_per_chunk; a.slab = c->mom_slab; a.nm = c->nm;
            a.Psir = c->tr.Psir; a.Mr = c->tr.Mr; a.G2 = c->pr.G2;
            if (launch_moments_fused(c->st, a))
                return fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
            launch_slab_sum(c->st, c->mom_slab, c->nchunk, m * (c->nm + 2), c->frec);
            // dPHI is a sum over the outputs (GPz.m:113): the moments accumulate, the column sums are per output
            launch_split_fused(c->st, c->frec, c->m, c->nm, c->mp, mom, cols + (size_t)o * 2 * mp, o > 0 ? 1 : 0);
            continue;
        }
        {
            Stage s(c, "row_epilogue");
            RowArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.n = c->tr.n; a.m = c->m; a.mp = c->mp; a.k = c->k; a.out = o;
            a.y = c->tr.Y; a.omega = c->tr.om; a.lnbeta = c->lnbeta; a.wbeta = c->wbeta; a.ldx = c->tr.n_pad;
            a.w = c->w + (size_t)o * m; a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
            a.dL = c->dL; a.colslab = c->colslab; a.scal = c->scal_slab; a.nwg = c->nwg_rows;
            launch_row_epilogue(c->st, a);
            launch_colslab_reduce(c->st, c->colslab, c->scal_slab, c->nwg_rows, c->mp, cols + (size_t)o * 2 * mp,
                                  scal + (size_t)o * 4);
        }
    }
    if (!fused) {
        {
            Stage s(c, "mul_phi");
            launch_mul_phi(c->st, c->dL, c->Phi, c->T, (size_t)c->tr.n_pad * mp);
        }
        Stage s(c, "moments");
        if (c->gen && !c->has_psi) {
            if (int e = moments_by_pattern(c, false, mom)) return e;
        } else if (c->gen && c->psi32) {
            int nch, rpc;
            psi32_chunks(c, &nch, &rpc);
            if (c->tr.psi_diag && psi32m_available(c->d))
                launch_psi32m_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, c->tr.Xr, c->de, c->d, c->tr.PsiT,
                                      (long)c->tr.n_pad, c->tr.n, c->m, c->pr.P, c->pr.Rc, nch, rpc, c->gen_slab);
            else
                launch_psi32_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, c->tr.Xr, c->de, c->d, c->tr.PsiT,
                                     (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m, c->pr.P, c->Sig, c->pr.Rc, nch, rpc,
                                     c->gen_slab, c->nrec);
            launch_slab_sum(c->st, c->gen_slab, nch, m * psi32_raw_len(c->d), c->psi32_raw);
            launch_psi32_records(c->st, c->psi32_raw, c->d, c->tr.psi_diag, c->m, mom, c->nrec);
        } else if (c->gen && c->psi_fast) {
            int nch = c->gen_nchunk;
            if (nch > c->tr.n) nch = c->tr.n;
            if (nch < 1) nch = 1;
            const int rpc = (c->tr.n + nch - 1) / nch > 0 ? (c->tr.n + nch - 1) / nch : 1;
            nch = (c->tr.n + rpc - 1) / rpc;
            if (nch < 1) nch = 1;
            if (c->psi_miss) {
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gen_rows(c->tr), c->m, c->d, c->de,
                                   c->pr.P, c->Sig, c->mom_nchunk, 0, c->gen_slab, c->nrec, c->pat_d, c->mom_chunktab, c->gc_minv);
                launch_slab_sum_seg(c->st, c->gen_slab, c->mom_segtab, c->ngroups, m * c->nrec, mom);
            } else {
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gen_rows(c->tr), c->m, c->d, c->de,
                                   c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, nullptr, nullptr, c->gc_minv);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, mom);
            }
        } else if (c->gen) {
            const GenRows gr = gen_rows(c->tr);
            for (int g = 0; g < c->ngroups; ++g) {
                const int rb = c->tr.group_begin[g], nr = c->tr.group_begin[g + 1] - rb;
                double *recs_g = mom + (size_t)g * m * c->nrec;
                if (nr <= 0) { launch_zero(c->st, recs_g, m * c->nrec); continue; }
                int nch = c->gen_nchunk;
                if (nch > nr) nch = nr;
                const int rpc = (nr + nch - 1) / nch;
                nch = (nr + rpc - 1) / rpc;
                launch_gen_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gr, g, rb, nr, c->pat_d, c->m, c->d,
                                   c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, c->gen_ws);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, recs_g);
            }
        } else {
        MomentArgs a{};
        a.dPhi = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.n = c->tr.n; a.n_pad = c->tr.n_pad; a.m = c->m; a.d = c->de;
        a.kind = c->kind; a.P = c->pr.P; a.nchunk = c->nchunk; a.rows_per_chunk = c->rows_per_chunk;
        a.slab = c->mom_slab; a.nm = c->nm;
        a.Psir = c->tr.Psir; a.Mr = c->tr.Mr; a.G2 = c->pr.G2;
        if (launch_moments(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
        launch_slab_sum(c->st, c->mom_slab, c->nchunk, m * c->nm, mom);
        }
    }
    const bool have_valid = c->va.n_pad > 0;
    if (have_valid && c->gen && !c->has_psi) {
        Stage s(c, "validation");
        if (int e = phi_by_pattern(c, c->va, nullptr, c->lnbeta_v, nullptr, c->w, c->phiw_v, false)) return e;
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    } else if (have_valid && c->gen) {
        Stage s(c, "validation");
        if (c->psi32) {
            launch_psi32_phi(c->st, c->va.Xr, c->de, c->d, c->va.PsiT, (long)c->va.n_pad, c->va.psi_diag, c->va.n, c->m,
                             c->pr.P, c->Sig, c->pr.Rc, c->lnS, c->Phi_v, c->mp);
            launch_gen_fill(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->mp, c->k, nullptr);
        } else if (c->psi_fast) {
            launch_psi_phi(c->st, gen_rows(c->va), c->m, c->d, c->de, c->pr.P, c->Sig, c->lnS, c->Phi_v, c->mp,
                           c->psi_miss ? c->pat_d : nullptr, c->mid == 4);
            launch_gen_fill(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->mp, c->k, nullptr);
        } else {
            launch_gen_phi(c->st, gen_rows(c->va), c->m, c->mp, c->d, c->de, c->k, c->pr.P, c->Sig, c->lnS, c->pat_d,
                           c->Phi_v, nullptr, c->gen_ws);
        }
        launch_gen_rowdot(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          nullptr, c->w, c->lnbeta_v, nullptr, c->phiw_v);
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    } else if (have_valid) {
        Stage s(c, "validation");
        PhiArgs a{};
        a.Xc = c->va.Xc; a.ldx = c->va.n_pad; a.n = c->va.n; a.n_pad = c->va.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->va.om; a.Y = nullptr;
        a.Phi = nullptr; a.lnbeta = c->lnbeta_v; a.wbeta = nullptr; a.w = c->w; a.phiw = c->phiw_v;
        a.Psic = c->va.Psic; a.Mc = c->va.Mc; a.ucnt = c->va.ucnt;
        if (launch_phi(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    }   // (no validation rows: k_unpack zeroed vsums at the start of the evaluation)
    {
        Stage s(c, "allreduce2");
        if (int e = allreduce(c, c->comm2, c->comm2_count)) return e;
    }
    {
        Stage s(c, "finish");
        FinishArgs a{};
        a.method_id = c->mid; a.kind = c->kind; a.m = c->m; a.d = c->d; a.k = c->k; a.hetero = c->hetero;
        a.g_dim = c->g_dim; a.pr = c->pr; a.mom = mom; a.nm = c->nm; a.cols = cols; a.scal = scal;
        a.w = c->w; a.dwda = c->dwda; a.dgi = c->dgi; a.logdet = c->logdet;
        a.sums1 = c->comm1 + k * mp * mp; a.vsums = have_valid ? vsums : nullptr; a.info = c->info;
        a.out = c->out_d; a.dGfull = c->dGfull; a.p = (int)c->p; a.nmp = c->mp; a.de = c->de;
        a.psi = (c->has_psi && !c->gen) ? 1 : 0; a.gen = c->gen ? 1 : 0;
        if (c->gen && c->psi32 && c->tr.psi_diag)   // whitened records: the stable chain through R (k_psi32.hip)
            launch_psi32_finish(c->st, mom, c->m, c->d, c->de, c->pr.G, c->pr.Rc, c->mid, a.sums1, c->k, c->out_d + 1, c->dGfull,
                                c->k == 1 ? cols : nullptr, c->mp, c->nrec);
        else if (c->gen)
            launch_gen_finish(c->st, mom, c->ngroups, c->pat_d, c->m, c->d, c->de, c->pr.G, c->Sig, c->iSig, c->mid, a.sums1,
                              c->k, c->out_d + 1, c->dGfull, c->k == 1 ? cols : nullptr, c->mp, c->nrec, c->fin_part,
                              c->has_psi ? 0 : 1, c->gen_ws);
        launch_finish(c->st, a);
    }
    if (c->g_dev_out) {   // gpz_eval_dev: the gradient stays on the device, only f and the statistics block come up
        HIPCHK(hipMemcpyAsync(c->g_dev_out, c->out_d + 1, (size_t)c->p * sizeof(double), hipMemcpyDeviceToDevice, c->st));
        HIPCHK(hipMemcpyAsync(c->out_h, c->out_d, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(hipMemcpyAsync(c->out_h + 1 + c->p, c->out_d + 1 + c->p, 9 * sizeof(double), hipMemcpyDeviceToHost, c->st));
    } else {
        HIPCHK(hipMemcpyAsync(c->out_h, c->out_d, ((size_t)c->p + 10) * sizeof(double), hipMemcpyDeviceToHost, c->st));
    }
    if (c->capturing) return 0;   // being recorded into the evaluation graph: the caller synchronises after the replay
    HIPCHK(hipStreamSynchronize(c->st));
    HIPCHK(hipGetLastError());
    return 0;
}

static int eval_common(gpz_ctx *c, const double *theta, const double *theta_dev, double *f, double *g, double *g_dev,
                       double stats[4], double diag[2]);

extern "C" int gpz_eval(gpz_ctx *c, const double *theta, double *f, double *g, double stats[4], double diag[2]) {
    if (!c || !theta || !f || !g) return fail(GPZ_ERR_ARG, "gpz_eval: null argument");
    return eval_common(c, theta, nullptr, f, g, nullptr, stats, diag);
}

extern "C" int gpz_eval_dev(gpz_ctx *c, const double *theta_dev, double *f, double *g_dev, double stats[4], double diag[2]) {
    if (!c || !theta_dev || !f || !g_dev) return fail(GPZ_ERR_ARG, "gpz_eval_dev: null argument");
    c->g_dev_out = g_dev;
    const int rc = eval_common(c, nullptr, theta_dev, f, nullptr, g_dev, stats, diag);
    c->g_dev_out = nullptr;
    return rc;
}

static int eval_common(gpz_ctx *c, const double *theta, const double *theta_dev, double *f, double *g, double *g_dev,
                       double stats[4], double diag[2]) {
    (void)g_dev;
    gpz_opts_scope opts_scope(&c->opt);
    HIPCHK(hipSetDevice(c->device));
    c->pinv_last[0] = c->pinv_last[1] = c->pinv_last[2] = c->pinv_last[3] = 0.0;
    const bool no_graph = c->opt.no_graph;   // (latched at creation: a context is either replayed or eager for its whole life)
    const bool graphable = theta && !c->g_dev_out && c->desc.world <= 1 && !c->timing && c->pinv_mode != 1 && !no_graph &&
                           c->graph_state >= 0;
    bool done = false;
    if (graphable && c->graph_state == 2) {
        memcpy(c->theta_h, theta, (size_t)c->p * sizeof(double));
        if (hipGraphLaunch(c->graph_exec, c->st) == hipSuccess) {
            HIPCHK(hipStreamSynchronize(c->st));
            HIPCHK(hipGetLastError());
            done = true;
        } else {
            (void)hipGetLastError();
            c->graph_state = -1;
        }
    } else if (graphable && c->graph_state == 1) {
        hipGraph_t graph = nullptr;
        int rc = 0;
        const char *why = "";
        hipStream_t user_st = c->st;
        hipError_t he = c->graph_st ? hipSuccess : hipStreamCreate(&c->graph_st);
        if (he == hipSuccess) {
            c->st = c->graph_st;
            he = hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal);
        }
        if (he == hipSuccess) {
            c->capturing = true;
            if (!rc && (rc = stage_a(c, theta, nullptr))) why = "stage A";   // (k_unpack clears the status words)
            if (!rc && (rc = eval_tail(c, false))) why = "stage B";
            c->capturing = false;
            const hipError_t he2 = hipStreamEndCapture(c->st, &graph);
            if (he2 != hipSuccess || !graph) { if (!rc) { rc = -1; why = "end capture"; he = he2; } }
        } else {
            rc = -1; why = "begin capture";
        }
        c->st = user_st;
        if (!rc && (he = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0)) != hipSuccess) { rc = -1; why = "instantiate"; }
        if (rc && c->opt.graph_debug)
            fprintf(stderr, "gpz: evaluation graph: %s: %s | %s\n", why, hipGetErrorString(he), gpz_last_error());
        if (graph) (void)hipGraphDestroy(graph);
        if (c->opt.graph_debug) fprintf(stderr, "gpz: evaluation graph capture %s\n", rc ? "failed" : "ok");
        if (!rc) {
            c->graph_state = 2;
            HIPCHK(hipGraphLaunch(c->graph_exec, c->st));
            HIPCHK(hipStreamSynchronize(c->st));
            HIPCHK(hipGetLastError());
            done = true;
        } else {                     // not capturable here: stay on plain launches for the life of the context
            (void)hipGetLastError();
            c->graph_exec = nullptr;
            c->graph_state = -1;
        }
    }
    if (!done) {
        if (int e = stage_a(c, theta, theta_dev)) return e;   // (k_unpack clears the status words)
        if (int e = eval_tail(c, c->pinv_mode == 1)) return e;
        if (graphable && c->graph_state == 0) c->graph_state = 1;
    }
    // k_cond_flag (info[1], returned in slot 7 of the statistics block): SIGMA is close enough to singular that
    // inv_logdet.m may truncate -> redo the solve and everything after it through the SVD route.  PHI and the
    // reduced partials of stage A are still in place; every rank sees the same SIGMA and takes the same branch.
    if (c->pinv_mode == 0 && c->out_h[1 + c->p + 7] != 0.0) {
        HIPCHK(hipMemsetAsync(c->info, 0, 2 * sizeof(int), c->st));
        if (int e = eval_tail(c, true)) return e;
    }
    const bool have_valid = c->va.n_pad > 0;
    if (c->timing) collect_timings(c);
    *f = c->out_h[0];
    if (g) memcpy(g, c->out_h + 1, (size_t)c->p * sizeof(double));
    const double *st = c->out_h + 1 + c->p;
    if (stats) {
        stats[0] = st[0];
        stats[1] = st[1];
        if (have_valid) { stats[2] = st[2]; stats[3] = st[3]; }
    }
    if (diag) { diag[0] = st[4]; diag[1] = st[5]; }
    return GPZ_OK;
}

extern "C" int gpz_solve(gpz_ctx *c, const double *theta, double *w, double *iSigma_w, double *nlogML_partial) {
    if (!c || !theta || !w || !iSigma_w) return fail(GPZ_ERR_ARG, "gpz_solve: null argument");
    gpz_opts_scope opts_scope(&c->opt);
    HIPCHK(hipSetDevice(c->device));
    if (int e = stage_a(c, theta)) return e;   // (k_unpack clears the status words)
    const size_t m = c->m, mq = c->mq;
    c->pinv_last[0] = c->pinv_last[1] = c->pinv_last[2] = c->pinv_last[3] = 0.0;
    for (int o = 0; o < c->k; ++o) {
        bool pinv = c->pinv_mode == 1;
        if (!pinv) {
            stage_b(c, o);
            if (c->pinv_mode == 0) {   // see gpz_eval: take the truncating route when k_cond_flag asks for it
                int ih[2] = {0, 0};
                HIPCHK(hipMemcpyAsync(ih, c->info, sizeof ih, hipMemcpyDeviceToHost, c->st));
                HIPCHK(hipStreamSynchronize(c->st));
                if (ih[1] != 0) {
                    HIPCHK(hipMemsetAsync(c->info, 0, 2 * sizeof(int), c->st));
                    pinv = true;
                }
            }
        }
        if (pinv) { if (int e = stage_b_pinv(c, o)) return e; }
        // inv(SIGMA) is symmetric (to rounding on the SVD route): row-major == column-major
        HIPCHK(hipMemcpy2DAsync(iSigma_w + (size_t)o * m * m, m * sizeof(double), c->Sinv, mq * sizeof(double),
                                m * sizeof(double), m, hipMemcpyDeviceToHost, c->st));
    }
    HIPCHK(hipMemcpyAsync(w, c->w, m * c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    if (nlogML_partial && c->gen) {
        launch_gen_rowdot(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          nullptr, c->w, c->lnbeta, nullptr, c->phiw);
        launch_row_stats(c->st, c->phiw, c->tr.Y, c->tr.om, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), c->rstats);
        if (int e = allreduce(c, c->rstats, gpz_ns(c->k))) return e;
        launch_solve_partial(c->st, c->pr, c->w, c->logdet, c->comm1 + (size_t)c->k * c->mp * c->mp, c->rstats, c->m,
                             c->k, c->spart);
        HIPCHK(hipMemcpyAsync(nlogML_partial, c->spart, c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    } else if (nlogML_partial) {
        PhiArgs a{};
        a.Xc = c->tr.Xc; a.ldx = c->tr.n_pad; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om; a.Y = nullptr;
        a.Phi = nullptr; a.lnbeta = c->lnbeta; a.wbeta = nullptr; a.w = c->w; a.phiw = c->phiw;
        a.Psic = c->tr.Psic; a.Mc = c->tr.Mc; a.ucnt = c->tr.ucnt;
        if (launch_phi(c->st, a)) return fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
        launch_row_stats(c->st, c->phiw, c->tr.Y, c->tr.om, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), c->rstats);
        if (int e = allreduce(c, c->rstats, gpz_ns(c->k))) return e;
        launch_solve_partial(c->st, c->pr, c->w, c->logdet, c->comm1 + (size_t)c->k * c->mp * c->mp, c->rstats, c->m,
                             c->k, c->spart);
        HIPCHK(hipMemcpyAsync(nlogML_partial, c->spart, c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    }
    int info_h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(info_h, c->info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    HIPCHK(hipGetLastError());
    if (c->timing) collect_timings(c);
    if (info_h[0] != 0) {
        for (size_t e = 0; e < m * c->k; ++e) w[e] = NAN;
        for (size_t e = 0; e < m * m * c->k; ++e) iSigma_w[e] = NAN;
        if (nlogML_partial)
            for (int o = 0; o < c->k; ++o) nlogML_partial[o] = NAN;
    }
    return GPZ_OK;
}

extern "C" int gpz_get_phi(gpz_ctx *c, double *PHI) {
    if (!c || !PHI) return fail(GPZ_ERR_ARG, "gpz_get_phi: null argument");
    if (c->tile_rows) return fail(GPZ_ERR_ARG, "gpz_get_phi: this context streams PHI in row tiles (it is never whole on the device); use gpz_phi");
    if (!c->phi_valid) return fail(GPZ_ERR_ARG, "gpz_get_phi: no evaluation has been run");
    HIPCHK(hipSetDevice(c->device));
    double *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, (size_t)c->tr.n * c->m * sizeof(double)));
    launch_transpose_out(c->st, c->Phi, c->mp, c->tr.n, c->m, tmp, c->tr.orig);
    hipError_t e = hipMemcpyAsync(PHI, tmp, (size_t)c->tr.n * c->m * sizeof(double), hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipStreamSynchronize(c->st);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(GPZ_ERR_HIP, "gpz_get_phi copy: %s", hipGetErrorString(e));
    return GPZ_OK;
}

// ---- stand-alone entry points --------------------------------------------------------------------
// A throw-away context without targets: parameters + PHI on ns rows (all rows selected).
static int make_eval_ctx(const gpz_desc *desc, const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                         gpz_ctx **out) {
    gpz_ctx *c = new gpz_ctx();
    int rc = setup_model(c, desc);
    if (rc) { delete c; return rc; }
    auto bail = [&](int code) { c->ar.release(); delete c; return code; };
    c->desc.world = 1;
    std::vector<double> y0((size_t)ns * c->k, 0.0);
    if ((rc = setup_data(c, ns, Xs, y0.data(), Psi, psi_kind, nullptr, nullptr, nullptr))) return bail(rc);
    const size_t np = c->tr.n_pad;
    if ((rc = c->ar.alloc(&c->Phi, np * c->mp))) return bail(rc);
    if ((rc = c->ar.alloc(&c->lnbeta, np * c->k))) return bail(rc);
    if ((rc = c->ar.alloc(&c->wbeta, np * c->k))) return bail(rc);
    *out = c;
    return 0;
}
static void free_eval_ctx(gpz_ctx *c) { c->ar.release(); delete c; }

static int run_phi_only(gpz_ctx *c, const double *theta) {
    HIPCHK(hipMemcpyAsync(c->theta_d, theta, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st));
    launch_unpack(c->st, c->theta_d, c->mid, c->m, c->d, c->de, c->k, c->hetero, c->pr);
    if (c->kind == GPZ_KIND_COV) launch_prep_cov(c->st, c->pr.G, c->pr.P, c->m, c->de, c->pr.Rc, c->prep_ws);
    return build_phi(c);
}

extern "C" int gpz_phi(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns, const double *Psi,
                       int32_t psi_kind, double *PHI, double *lnBeta_i, double *N) {
    if (!desc || !theta || !Xs || ns < 1) return fail(GPZ_ERR_ARG, "gpz_phi: null argument");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    int rc = run_phi_only(c, theta);
    double *tmp = nullptr, *nd = nullptr;
    if (!rc && (PHI || N)) rc = c->ar.alloc(&tmp, (size_t)ns * c->m);
    if (!rc && PHI) {
        launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp, c->tr.orig);
        if (hipMemcpyAsync(PHI, tmp, (size_t)ns * c->m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
    }
    if (!rc && N) {   // N = exp(lnN), lnN = lnPHI - 1/2 ln|Sigma_oo| - 1/2 |o| ln 2pi + 1/2 |u| ln 2   (getPHI.m:77,87,98,105,114)
        rc = c->ar.alloc(&nd, (size_t)c->tr.n_pad * c->mp);
        if (!rc) {
            NormArgs a{};
            a.Phi = c->Phi; a.ld = c->mp; a.n = (int)ns; a.m = c->m; a.d = c->d; a.de = c->de; a.kind = c->kind;
            a.gen = c->gen ? 1 : 0; a.G = c->pr.G; a.Rc = c->pr.Rc; a.Mr = c->tr.Mr; a.ucnt = c->tr.ucnt;
            a.gid = c->tr.gid; a.pat = c->pat_d; a.lnS = c->lnS; a.N = nd;
            launch_phi_norm(c->st, a);
            launch_transpose_out(c->st, nd, c->mp, ns, c->m, tmp, c->tr.orig);
            if (hipMemcpyAsync(N, tmp, (size_t)ns * c->m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
        }
    }
    if (!rc && lnBeta_i) {
        if (hipMemcpy2DAsync(lnBeta_i, (size_t)ns * sizeof(double), c->lnbeta, (size_t)c->tr.n_pad * sizeof(double),
                             (size_t)ns * sizeof(double), c->k, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = fail(GPZ_ERR_HIP, "gpz_phi: sync failed");
    if (!rc && lnBeta_i && !c->tr.orig_h.empty()) {   // rows are stored sorted by NaN pattern: back to the caller's order
        std::vector<double> t((size_t)ns);
        for (int o = 0; o < c->k; ++o) {
            double *col = lnBeta_i + (size_t)o * ns;
            for (int64_t r = 0; r < ns; ++r) t[(size_t)c->tr.orig_h[(size_t)r]] = col[r];
            memcpy(col, t.data(), (size_t)ns * sizeof(double));
        }
    }
    free_eval_ctx(c);
    return rc;
}

extern "C" int gpz_predict_full(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                const double *Xs, int64_t ns, double *mu, double *nu, double *beta_i, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !Xs || ns < 1 || !mu || !nu || !beta_i)
        return fail(GPZ_ERR_ARG, "gpz_predict_full: null argument");
    gpz_ctx *c = nullptr;
    if (has_nan(Xs, ns * (int64_t)desc->d))
        return fail(GPZ_ERR_UNSUPPORTED, "gpz_predict_full: the rows have missing values (NaN): group them by pattern and call gpz_predict_missing (predict.m:45-69)");
    if (int e = make_eval_ctx(desc, Xs, ns, nullptr, 0, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const size_t m = c->m, mp = c->mp, np = c->tr.n_pad, k = c->k;
    int rc = 0;
    double *T = nullptr, *Bext = nullptr, *wd = nullptr, *Sd = nullptr, *nud = nullptr, *dgi = nullptr, *tmp = nullptr;
    if (!rc) rc = c->ar.alloc(&T, np * mp);
    if (!rc) rc = c->ar.alloc(&Bext, mp * mp);
    if (!rc) rc = c->ar.alloc(&wd, m * k);
    if (!rc) rc = c->ar.alloc(&Sd, m * m);
    if (!rc) rc = c->ar.alloc(&nud, np);
    if (!rc) rc = c->ar.alloc(&dgi, m);
    if (!rc) rc = run_phi_only(c, theta);
    std::vector<double> hbuf(np);
    if (!rc && hipMemcpyAsync(wd, w, m * k * sizeof(double), hipMemcpyHostToDevice, c->st) != hipSuccess)
        rc = fail(GPZ_ERR_HIP, "predict: copy failed");
    for (int o = 0; o < (int)k && !rc; ++o) {
        // iSigma_w(:,:,o) is m x m (symmetric up to rounding in the reference; used as given, B[k][j] = iS(k,j))
        std::vector<double> rowmaj(m * m);
        const double *src = iSigma_w + (size_t)o * m * m;
        for (size_t a = 0; a < m; ++a)
            for (size_t b = 0; b < m; ++b) rowmaj[a * m + b] = src[a + m * b];
        if (hipMemcpy(Sd, rowmaj.data(), m * m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(GPZ_ERR_HIP, "predict: copy failed");
            break;
        }
        launch_fill_bext(c->st, Sd, (int)m, wd + (size_t)o * m, (int)m, (int)mp, o, Bext, dgi);
        launch_tgemm(c->st, c->Phi, (int)mp, Bext, (int)mp, T, (int)np, (int)mp, nullptr, nullptr, (int)m, -1);
        launch_nu(c->st, c->Phi, T, (int)mp, (int)ns, (int)m, nud);                       // predictDiag.m:69-71
        if (hipMemcpyAsync(nu + (size_t)o * ns, nud, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "predict: copy failed");
        // mu(:,o) = PHI*w(:,o) = column m+o of T                                         // predictDiag.m:65
        if (!rc && hipMemcpy2DAsync(mu + (size_t)o * ns, sizeof(double), T + m + o, mp * sizeof(double), sizeof(double),
                                    (size_t)ns, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "predict: copy failed");
        if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = fail(GPZ_ERR_HIP, "predict: sync failed");
    }
    if (!rc) {
        // beta_i = exp(lnBeta_i)   (predictDiag.m:73): wbeta holds exp(-lnbeta) (omega = 1)
        std::vector<double> lb((size_t)ns * k);
        if (hipMemcpy2D(lb.data(), (size_t)ns * sizeof(double), c->wbeta, np * sizeof(double), (size_t)ns * sizeof(double), k,
                        hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "predict: copy failed");
        else
            for (size_t e = 0; e < (size_t)ns * k; ++e) beta_i[e] = 1.0 / lb[e];
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, (int)mp, ns, (int)m, tmp);
            if (hipMemcpy(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(GPZ_ERR_HIP, "predict: copy failed");
        }
    }
    (void)hipStreamSynchronize(c->st);
    free_eval_ctx(c);
    return rc;
}

// predictNoisy (predictDiag.m:75-125, predictCov.m:70-132): inputs with noise Psi, no missing values.
extern "C" int gpz_predict_noisy(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                 const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind, double *mu,
                                 double *nu, double *beta_i, double *gamma, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !Xs || !Psi || ns < 1 || !mu || !nu || !beta_i || !gamma)
        return fail(GPZ_ERR_ARG, "gpz_predict_noisy: null argument");
    if (has_nan(Xs, ns * (int64_t)desc->d))
        return fail(GPZ_ERR_UNSUPPORTED, "gpz_predict_noisy: the rows have missing values (NaN): group them by pattern and call gpz_predict_missing (predict.m:45-69)");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const size_t m = c->m, np = c->tr.n_pad, k = c->k;
    const int d = c->d;
    int rc = run_phi_only(c, theta);
    double *wd = nullptr, *iSd = nullptr, *phiw = nullptr, *tab = nullptr, *part = nullptr, *sums = nullptr, *outb = nullptr,
           *tmp = nullptr;
    const long npair = (long)m * (m + 1) / 2;
    const int rec = 1 + d + (c->kind == GPZ_KIND_COV ? d * d : d);
    // split the pairs so that ~1024 workgroups exist
    int nchunk = (int)((1024 + (ns + 63) / 64 - 1) / ((ns + 63) / 64));
    if (nchunk > npair) nchunk = (int)npair;
    if (nchunk > 256) nchunk = 256;
    if (nchunk < 1) nchunk = 1;
    const long ppc = (npair + nchunk - 1) / nchunk;
    nchunk = (int)((npair + ppc - 1) / ppc);
    if (!rc) rc = c->ar.alloc(&wd, m * k);
    if (!rc) rc = c->ar.alloc(&iSd, m * m * k);
    if (!rc) rc = c->ar.alloc(&phiw, np * k);
    if (!rc) rc = c->ar.alloc(&tab, (size_t)npair * rec);
    if (!rc) rc = c->ar.alloc(&part, (size_t)nchunk * 3 * k * np);
    if (!rc) rc = c->ar.alloc(&sums, (size_t)3 * k * np);
    if (!rc && d > 20 && !c->gen_ws)   // a diagonal kind at d > 20: the runtime-d pair table / pair sums take their temporaries from here
        rc = c->ar.alloc(&c->gen_ws, (size_t)gen_rt_threads(d) * gen_ws_per_thread(d));
    if (!rc) rc = c->ar.alloc(&outb, (size_t)3 * k * np);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(wd, w, m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(iSd, iSigma_w, m * m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
    }
    if (!rc) {
        // mu = PHI*w (lnbeta = ElnS is already there)                                       predictDiag.m:82
        launch_gen_rowdot(c->st, c->Phi, c->mp, c->tr.n, (long)np, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          nullptr, wd, c->lnbeta, nullptr, phiw);
        launch_zero(c->st, part, (size_t)nchunk * 3 * k * np);
        launch_pair_table(c->st, c->kind, c->m, d, c->de, c->pr.P, c->pr.G, c->Sig, c->iSig, tab, rec, c->gen_ws);
        launch_predict_noisy(c->st, c->kind, c->tr.n, (long)np, c->m, d, c->de, c->k, c->tr.Xr, c->tr.Psir, c->tr.Psi3, tab,
                             rec, wd, c->hetero ? c->pr.v : nullptr, iSd, nchunk, ppc, part, c->gen_ws,
                             (c->tr.psi_diag ? 1 : 0) | (c->mid == 4 ? 2 : 0));
        launch_slab_sum(c->st, part, nchunk, (size_t)3 * k * np, sums);
        launch_predict_noisy_final(c->st, sums, (long)np, c->tr.n, c->k, phiw, c->lnbeta, c->pr.b, outb, outb + k * np,
                                   outb + 2 * k * np);
        auto down = [&](double *dst, const double *src) {
            return hipMemcpy2DAsync(dst, (size_t)ns * sizeof(double), src, np * sizeof(double), (size_t)ns * sizeof(double), k,
                                    hipMemcpyDeviceToHost, c->st);
        };
        hipError_t e = down(gamma, outb);
        if (e == hipSuccess) e = down(nu, outb + k * np);
        if (e == hipSuccess) e = down(beta_i, outb + 2 * k * np);
        if (e == hipSuccess) e = down(mu, phiw);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = fail(GPZ_ERR_HIP, "gpz_predict_noisy: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_noisy: kernel failed");
    free_eval_ctx(c);
    return rc;
}

// predict.m:60-69 calls once per NaN-pattern group with the same model, and the entry point is stateless: Sigma_j / inv(Sigma_j)
// (k_gen_prep) and the basis-pair table (k_pmc_pairs: m (m + 1) / 2 d x d inversions) depend on theta, w and iSigma_w only and were
// half of a many-group call (profiles/r03_predict_wide_kernel_stats.txt).  The last model's tables stay on the device, one entry
// per device, keyed by the CONTENTS of theta, w, iSigma_w; gpz_release_cached_memory() drops them.
struct PmcModelCache {
    std::mutex mu;                     // held for the whole call: one group at a time per device
    int m = 0, d = 0, k = 0, mid = -1, hetero = -1;
    std::vector<double> theta, w, iS;
    double *Sig = nullptr, *iSig = nullptr, *tab = nullptr;
    void drop() {
        if (Sig) (void)hipFree(Sig);
        if (iSig) (void)hipFree(iSig);
        if (tab) (void)hipFree(tab);
        Sig = iSig = tab = nullptr;
        m = d = k = 0; mid = hetero = -1;
        theta.clear(); w.clear(); iS.clear();
    }
};
static PmcModelCache *pmc_model_cache(int dev) {
    static std::mutex mu;
    static std::map<int, PmcModelCache *> *by_dev = new std::map<int, PmcModelCache *>();   // never destroyed (see dev_cache)
    std::lock_guard<std::mutex> g(mu);
    auto it = by_dev->find(dev);
    if (it != by_dev->end()) return it->second;
    return (*by_dev)[dev] = new PmcModelCache();
}
static void pmc_model_cache_release_all() {
    int cur = 0, ndev = 0;
    (void)hipGetDevice(&cur);
    (void)hipGetDeviceCount(&ndev);
    for (int dev = 0; dev < ndev; ++dev) {
        PmcModelCache *e = pmc_model_cache(dev);
        std::unique_lock<std::mutex> g(e->mu, std::try_to_lock);
        if (!g.owns_lock()) continue;      // a prediction is using this entry right now (possibly this very thread): leave it
        if (!e->Sig && !e->tab) continue;
        (void)hipSetDevice(dev);
        e->drop();
    }
    (void)hipSetDevice(cur);
}

// GC/VC branch of gpz_predict_missing (predictCov.m:134-337); see k_pmiss_cov.hip.
static int predict_missing_cov(const gpz_desc *desc, const std::vector<unsigned char> &flags, const double *theta, const double *w, const double *iSigma_w,
                               const double *priors, const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                               double *mu, double *nu, double *beta_i, double *gamma, double *PHI) {
    if (Psi && psi_kind != 2 && psi_kind != 3) return fail(GPZ_ERR_ARG, "GC/VC take Psi as a d x d x n cube (fixPsi.m:22-38) or n x d variances (psi_kind 3)");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const size_t m = c->m, mp = c->mp, np = c->tr.n_pad, k = c->k;
    const int n = c->tr.n, d = c->d, de = c->de;
    int rc = 0;
    if (hipMemcpyAsync(c->theta_d, theta, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st) != hipSuccess)
        rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    launch_unpack(c->st, c->theta_d, c->mid, c->m, c->d, c->de, c->k, c->hetero, c->pr);
    unsigned long long obs = 0ull;       // the 64-bit form of the pattern (the routes up to d = 64 take it by value)
    int n_obs = 0;
    for (int a = 0; a < d; ++a)
        if (flags[a]) { ++n_obs; if (a < 64) obs |= 1ull << a; }
    const bool generic = d > 64;          // any width: temporaries in a device workspace (k_pmiss_covg.hip)
    const int nrec = 2 + n_obs * n_obs + n_obs * (d - n_obs) + (d - n_obs) * (d - n_obs), ntab = d * d + d + 1 + 3 * (int)k;
    const long npairs = (long)m * (m + 1) / 2;
    // the model's tables of the previous group, if it was the same model (see PmcModelCache)
    PmcModelCache *mc = pmc_model_cache(c->device);
    std::unique_lock<std::mutex> mc_lock(mc->mu);
    const size_t sig_n = m * (size_t)d * d, tab_n = (size_t)npairs * ntab;
    bool cacheable = (tab_n + 2 * sig_n) * sizeof(double) <= (2048UL << 20) && !gpz_opts().pmc_no_model_cache;
    bool hit = cacheable && mc->tab && mc->m == (int)m && mc->d == d && mc->k == (int)k && mc->mid == c->mid &&
               mc->hetero == (int)c->hetero && mc->theta.size() == (size_t)c->p &&
               memcmp(mc->theta.data(), theta, (size_t)c->p * sizeof(double)) == 0 &&
               memcmp(mc->w.data(), w, m * k * sizeof(double)) == 0 &&
               memcmp(mc->iS.data(), iSigma_w, m * m * k * sizeof(double)) == 0;
    if (cacheable && !hit) {
        mc->drop();
        if (hipMalloc((void **)&mc->Sig, sig_n * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&mc->iSig, sig_n * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&mc->tab, tab_n * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            mc->drop();
            cacheable = false;
        }
    }
    double *SigU = cacheable ? mc->Sig : c->Sig, *iSigU = cacheable ? mc->iSig : c->iSig;
    if (!hit) launch_gen_prep(c->st, c->pr.G, c->m, d, de, SigU, iSigU, c->pat_d, c->ngroups, c->lnS, c->gen_ws);
    // rows per block: X_hat / Psi_hat of a block stay below ~512 MB
    long rb = (1L << 26) / ((long)m * d * d);
    if (rb > n) rb = n;
    if (rb < 1) rb = 1;
    const bool fast = pmc_fast(d, (int)k);
    if (fast && rb > 64) rb = 64;   // the register-resident kernels deal the rows of a block over the lanes of a wave
    if (generic) rb = 1;            // one row at a time: its tables are what the workspace-resident kernels read
    const int rows_blk = (int)rb;
    // pair chunks = slabs of `part`: one wave per chunk on the register-resident route (fill the chip), 64 otherwise
    const long want = fast ? 2048 : 64;
    int nchunk = (int)(npairs < want ? npairs : want);
    const long ppc = (npairs + nchunk - 1) / nchunk;
    nchunk = (int)((npairs + ppc - 1) / ppc);
    double *wd = nullptr, *iSd = nullptr, *prd = nullptr, *rec = nullptr, *tab = nullptr, *Ex = nullptr, *Pio = nullptr,
           *Xhat = nullptr, *Phat = nullptr, *part = nullptr, *sums = nullptr, *phiw = nullptr, *outb = nullptr, *tmp = nullptr;
    if (!rc) rc = c->ar.alloc(&wd, m * k);
    if (!rc) rc = c->ar.alloc(&iSd, m * m * k);
    if (!rc) rc = c->ar.alloc(&prd, m);
    if (!rc) rc = c->ar.alloc(&rec, m * nrec);
    if (cacheable) tab = mc->tab;
    else if (!rc) rc = c->ar.alloc(&tab, (size_t)npairs * ntab);
    if (!rc) rc = c->ar.alloc(&Ex, (size_t)rows_blk * mp);
    if (!rc) rc = c->ar.alloc(&Pio, (size_t)rows_blk * mp);
    if (!rc) rc = c->ar.alloc(&Xhat, (size_t)rows_blk * m * d);
    if (!rc && Psi) rc = c->ar.alloc(&Phat, (size_t)rows_blk * m * d * d);
    if (!rc) rc = c->ar.alloc(&part, (size_t)nchunk * 3 * k * np);
    if (!rc) rc = c->ar.alloc(&sums, 3 * k * np);
    if (!rc) rc = c->ar.alloc(&phiw, np * k);
    if (!rc) rc = c->ar.alloc(&outb, 3 * k * np);
    double *work2 = nullptr;
    if (!rc && fast) rc = c->ar.alloc(&work2, m * ((size_t)d * (d + 1) / 2 + (size_t)d * d + d + 1));
    // d > 64: 3 d^2 + 2 d doubles of workspace per thread, at most 2 GB of it (and at least one wave's worth) per launch
    double *gws = nullptr, *gpat = nullptr;
    long gthreads = 0;
    if (!rc && generic) {
        const size_t per = pmg_ws_per_thread(d);
        gthreads = (long)((2048UL << 20) / (per * sizeof(double)));
        gthreads = gthreads > 65536 ? 65536 : (gthreads < 64 ? 64 : gthreads / 64 * 64);
        rc = c->ar.alloc(&gws, (size_t)gthreads * per);
        if (!rc) rc = c->ar.alloc(&gpat, (size_t)(3 * d + 1) / 2 + 1);   // 3 d ints
    }
    if (!rc) {
        hipError_t e = hipMemcpyAsync(wd, w, m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(iSd, iSigma_w, m * m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(prd, priors, m * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc) {
        launch_zero(c->st, c->Phi, np * mp);
        launch_zero(c->st, part, (size_t)nchunk * 3 * k * np);
        // d > 64: workspace-resident kernels (k_pmiss_covg.hip); 32 < d <= 64: the scratch-resident kernels with 64-wide temporaries
        // (k_pmiss_cov64.hip); else every route of k_pmiss_cov.hip
        if (generic)
            launch_pmc_generic(c->st, flags.data(), n, (long)np, c->m, (int)mp, d, de, c->k, c->tr.Xr, c->tr.Psi3, c->pr.P, SigU, iSigU, prd,
                               wd, c->hetero ? c->pr.v : nullptr, iSd, rec, tab, Ex, Pio, Xhat, Phat, nchunk, ppc, part, c->Phi, hit,
                               (int *)gpat, gws, gthreads);
        else
        (d > 32 ? launch_pmc_wide : launch_pmc)(c->st, obs, n, (long)np, c->m, (int)mp, d, de, c->k, c->tr.Xr, c->tr.Psi3, c->pr.P, SigU,
                                               iSigU, prd, wd, c->hetero ? c->pr.v : nullptr, iSd, rows_blk, rec, tab, Ex, Pio, Xhat,
                                               Phat, nchunk, ppc, part, c->Phi, work2, hit);
        launch_slab_sum(c->st, part, nchunk, 3 * k * np, sums);
        launch_gen_rowdot(c->st, c->Phi, c->mp, n, (long)np, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b, nullptr, wd,
                          c->lnbeta, nullptr, phiw);
        launch_predict_noisy_final(c->st, sums, (long)np, n, c->k, phiw, c->lnbeta, c->pr.b, outb, outb + k * np,
                                   outb + 2 * k * np);
        auto down = [&](double *dst, const double *src) {
            return hipMemcpy2DAsync(dst, (size_t)ns * sizeof(double), src, np * sizeof(double), (size_t)ns * sizeof(double), k,
                                    hipMemcpyDeviceToHost, c->st);
        };
        hipError_t e = down(gamma, outb);
        if (e == hipSuccess) e = down(nu, outb + k * np);
        if (e == hipSuccess) e = down(beta_i, outb + 2 * k * np);
        if (e == hipSuccess) e = down(mu, phiw);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: kernel failed");
    if (cacheable && !hit) {
        if (rc) mc->drop();            // never keep tables of a call that failed
        else {
            mc->m = (int)m; mc->d = d; mc->k = (int)k; mc->mid = c->mid; mc->hetero = (int)c->hetero;
            mc->theta.assign(theta, theta + c->p);
            mc->w.assign(w, w + m * k);
            mc->iS.assign(iSigma_w, iSigma_w + m * m * k);
        }
    }
    mc_lock.unlock();
    free_eval_ctx(c);
    return rc;
}

// GL/VL/GD/VD branch (predictDiag.m:127-297; k_pmiss.hip).  OBS = ObsMask (the pattern by value, LDS tiles: d <= GPZ_PM_MAXD_DIAG) or
// ObsFlags (the pattern as device bytes, uploaded here from *flags: any d).
template <typename OBS>
static int predict_missing_diag(const gpz_desc *desc, OBS obs, const std::vector<unsigned char> *flags, const double *theta,
                                const double *w, const double *iSigma_w, const double *priors, const double *Xs, int64_t ns,
                                const double *Psi, int32_t psi_kind, double *mu, double *nu, double *beta_i, double *gamma, double *PHI) {
    const int d = desc->d;
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const size_t m = c->m, mp = c->mp, np = c->tr.n_pad, k = c->k;
    const int n = c->tr.n, de = c->de;
    int rc = 0;
    HIPCHK(hipMemcpyAsync(c->theta_d, theta, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st));
    launch_unpack(c->st, c->theta_d, c->mid, c->m, c->d, c->de, c->k, c->hetero, c->pr);
    double *No = nullptr, *Pio = nullptr, *B = nullptr, *T = nullptr, *wd = nullptr, *iSd = nullptr, *prd = nullptr,
           *rec = nullptr, *sums = nullptr, *phiw = nullptr, *outb = nullptr, *tmp = nullptr;
    if constexpr (std::is_same<OBS, ObsFlags>::value) {
        double *fl = nullptr;
        rc = c->ar.alloc(&fl, (size_t)d / 8 + 1);
        if (!rc && hipMemcpy(fl, flags->data(), (size_t)d, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
        obs.f = (const unsigned char *)fl;
    }
    const int nrec = 2 * d + 1 + 3 * (int)k;
    // pair chunks of cw * mp pairs: a NaN-pattern group is often a few dozen rows, and then the launches per chunk are what it
    // costs - wider chunks, fewer of them, as far as the chunk's T (np x width) stays under 2 GB
    int cw = 32;   // (8 until round 3: the 128-row GEMM of a small group ran 32 workgroups per launch)
    while (cw > 1 && (double)np * (double)(cw * mp) * 8.0 > 2e9) cw >>= 1;
    const size_t width = (size_t)rup((long)cw * (long)mp, 64);   // whole 64-pair blocks of the pair-table kernel
    if (!rc) rc = c->ar.alloc(&No, np * mp);
    if (!rc) rc = c->ar.alloc(&Pio, np * mp);
    if (!rc) rc = c->ar.alloc(&B, mp * width);
    if (!rc) rc = c->ar.alloc(&T, np * (width > mp ? width : mp));
    if (!rc) rc = c->ar.alloc(&wd, m * k);
    if (!rc) rc = c->ar.alloc(&iSd, m * m * k);
    if (!rc) rc = c->ar.alloc(&prd, m);
    if (!rc) rc = c->ar.alloc(&rec, width * nrec);
    const int nsp = pm_accum_splits(n);   // pair splits of the accumulation kernel: one slab of sums each
    double *sums_s = nullptr;
    if (!rc) rc = c->ar.alloc(&sums_s, (size_t)nsp * 3 * k * np);
    if (!rc) rc = c->ar.alloc(&sums, 3 * k * np);
    if (!rc) rc = c->ar.alloc(&phiw, np * k);
    if (!rc) rc = c->ar.alloc(&outb, 3 * k * np);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(wd, w, m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(iSd, iSigma_w, m * m * k * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(prd, priors, m * sizeof(double), hipMemcpyHostToDevice, c->st);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc) {
        const double *Psir = c->has_psi ? c->tr.Psir : nullptr;
        launch_pm_no(c->st, c->tr.Xr, Psir, de, n, (long)np, c->m, (int)mp, d, obs, c->pr.P, c->pr.G, prd, No, Pio);
        // PHI = No .* (Pio * Nij') .* exp(lnz)                                              predictDiag.m:158-161
        launch_pm_nij(c->st, c->m, (int)mp, d, de, obs, c->pr.P, c->pr.G, B);
        // the GEMMs run over the group's rows rounded up to the kernel's 128-row tile, not over the 1024-row padding of the row
        // set: a NaN-pattern group is often a few dozen rows (7 of 8 row tiles were zeros)
        const int npg = rup(n, 128);
        launch_tgemm(c->st, Pio, (int)mp, B, (int)mp, T, npg, (int)mp, nullptr, nullptr, c->m, -1);
        launch_pm_phi(c->st, No, T, (int)mp, n, (long)np, c->m, d, de, c->pr.G, c->Phi);
        // mu = PHI*w, ElnS = PHI*v (+ b)                                                    predictDiag.m:163-164,203
        launch_gen_rowdot(c->st, c->Phi, c->mp, n, (long)np, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b, nullptr, wd,
                          c->lnbeta, nullptr, phiw);
        launch_zero(c->st, sums_s, (size_t)nsp * 3 * k * np);
        const long npairs = (long)m * (m + 1) / 2;
        for (long q0 = 0; q0 < npairs; q0 += (long)width) {                                  // predictDiag.m:170-200
            const int npq = (int)((npairs - q0 < (long)width) ? npairs - q0 : (long)width);
            launch_pm_pairtab(c->st, q0, npairs, c->m, (int)mp, (int)width, d, de, c->k, obs, c->has_psi ? 1 : 0, c->pr.P,
                              c->pr.G, wd, c->hetero ? c->pr.v : nullptr, iSd, B, rec, nrec);
            launch_tgemm(c->st, Pio, (int)mp, B, (int)width, T, npg, (int)width, nullptr, nullptr, c->m, -1, false, (int)mp,
                         (int)width);
            launch_pm_accum(c->st, c->tr.Xr, Psir, de, n, (long)np, (int)width, d, c->k, obs, npq, T, rec, nrec, sums_s, nsp);
        }
        launch_slab_sum(c->st, sums_s, nsp, 3 * k * np, sums);
        launch_predict_noisy_final(c->st, sums, (long)np, n, c->k, phiw, c->lnbeta, c->pr.b, outb, outb + k * np,
                                   outb + 2 * k * np);
        auto down = [&](double *dst, const double *src) {
            return hipMemcpy2DAsync(dst, (size_t)ns * sizeof(double), src, np * sizeof(double), (size_t)ns * sizeof(double), k,
                                    hipMemcpyDeviceToHost, c->st);
        };
        hipError_t e = down(gamma, outb);
        if (e == hipSuccess) e = down(nu, outb + k * np);
        if (e == hipSuccess) e = down(beta_i, outb + 2 * k * np);
        if (e == hipSuccess) e = down(mu, phiw);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_predict_missing: kernel failed");
    free_eval_ctx(c);
    return rc;
}

// predictMissing / predictNoisyMissing (predictDiag.m:127-297, predictCov.m:134-337) for ONE group of rows sharing a NaN pattern (the caller
// groups the rows as predict.m:45-69 does; the pattern is taken from the first row, predictDiag.m:3).
extern "C" int gpz_predict_missing(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                   const double *priors, const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                                   double *mu, double *nu, double *beta_i, double *gamma, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !priors || !Xs || ns < 1 || !mu || !nu || !beta_i || !gamma)
        return fail(GPZ_ERR_ARG, "gpz_predict_missing: null argument");
    const int d = desc->d;
    const bool covk = method_id_of(desc->method) >= 4;
    // any d (the reference is generic in it): the tuned routes cover d <= 64 (GC/VC) and d <= GPZ_PM_MAXD_DIAG (GL/VL/GD/VD); wider
    // inputs run the workspace-resident / LDS-free forms of the same kernels (include/gpz_hip.h has the cost line)
    std::vector<unsigned char> flags((size_t)d, 0);
    ObsMask obs = {{0ull, 0ull, 0ull, 0ull}};
    int nobs = 0;
    for (int c = 0; c < d; ++c) {
        const double xv = Xs[(size_t)c * ns];
        if (xv == xv) {
            flags[c] = 1;
            if (c < GPZ_PM_MAXD) obs.w[c >> 6] |= 1ull << (c & 63);
            ++nobs;
        }
    }
    for (int c = 0; c < d; ++c)
        for (int64_t i = 0; i < ns; ++i) {
            const double xv = Xs[(size_t)c * ns + i];
            if ((xv == xv) != (flags[c] != 0))
                return fail(GPZ_ERR_ARG, "gpz_predict_missing: the rows of a group must share one NaN pattern (predict.m:45-57)");
        }
    if (nobs == d)
        return fail(GPZ_ERR_ARG, "gpz_predict_missing: no dimension is missing (use gpz_predict_full / gpz_predict_noisy)");
    if (covk)
        return predict_missing_cov(desc, flags, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
    if (d > GPZ_PM_MAXD_DIAG) {
        ObsFlags of{nullptr};
        return predict_missing_diag(desc, of, &flags, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
    }
    return predict_missing_diag(desc, obs, nullptr, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
}

// prior = getPrior(X,Psi,theta,model,[])   (getPrior.m): N once, then the fixed point on the device; the convergence
// test on the m-vector (getPrior.m:18) runs on the host between iterations.
extern "C" int gpz_prior(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns, const double *Psi,
                         int32_t psi_kind, double *prior, int32_t *iterations) {
    if (!desc || !theta || !Xs || ns < 1 || !prior) return fail(GPZ_ERR_ARG, "gpz_prior: null argument");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const int m = c->m;
    int rc = run_phi_only(c, theta);
    double *nd = nullptr, *pd = nullptr, *slab = nullptr, *colsum = nullptr;
    const int nwg = ns < 1024 ? (int)ns : 1024;
    if (!rc) rc = c->ar.alloc(&nd, (size_t)c->tr.n_pad * c->mp);
    if (!rc) rc = c->ar.alloc(&pd, (size_t)m);
    if (!rc) rc = c->ar.alloc(&slab, (size_t)nwg * m);
    if (!rc) rc = c->ar.alloc(&colsum, (size_t)m);
    std::vector<double> pr(m, 1.0 / m), old(m), cs(m);                     // getPrior.m:5
    int it = 0;
    if (!rc) {
        NormArgs a{};
        a.Phi = c->Phi; a.ld = c->mp; a.n = (int)ns; a.m = c->m; a.d = c->d; a.de = c->de; a.kind = c->kind;
        a.gen = c->gen ? 1 : 0; a.G = c->pr.G; a.Rc = c->pr.Rc; a.Mr = c->tr.Mr; a.ucnt = c->tr.ucnt;
        a.gid = c->tr.gid; a.pat = c->pat_d; a.lnS = c->lnS; a.N = nd;
        launch_phi_norm(c->st, a);
        for (it = 1; it <= 100 && !rc; ++it) {                             // getPrior.m:7
            old = pr;
            hipError_t e = hipMemcpyAsync(pd, pr.data(), m * sizeof(double), hipMemcpyHostToDevice, c->st);
            launch_prior_iter(c->st, nd, c->mp, (int)ns, m, pd, slab, nwg);
            launch_slab_sum(c->st, slab, nwg, (size_t)m, colsum);
            if (e == hipSuccess) e = hipMemcpyAsync(cs.data(), colsum, m * sizeof(double), hipMemcpyDeviceToHost, c->st);
            if (e == hipSuccess) e = hipStreamSynchronize(c->st);
            if (e != hipSuccess) { rc = fail(GPZ_ERR_HIP, "gpz_prior: %s", hipGetErrorString(e)); break; }
            double num = 0.0, den = 0.0;
            for (int j = 0; j < m; ++j) {
                pr[j] = cs[j] / (double)ns;                                // mean(w)   getPrior.m:15
                num += (old[j] - pr[j]) * (old[j] - pr[j]);
                den += (old[j] + pr[j]) * (old[j] + pr[j]);
            }
            if (sqrt(num) / sqrt(den) < 1e-10) break;                      // getPrior.m:17-19
        }
    }
    if (!rc) {
        memcpy(prior, pr.data(), m * sizeof(double));
        if (iterations) *iterations = it > 100 ? 100 : it;
    }
    free_eval_ctx(c);
    return rc;
}

extern "C" int gpz_inv_logdet(const double *Ain, int32_t m, int32_t device, double *Xi, double *logdet, int32_t *info) {
    if (!Ain || m < 1 || !Xi || !logdet) return fail(GPZ_ERR_ARG, "gpz_inv_logdet: null argument");
    gpz_ctx *c = new gpz_ctx();
    gpz_opts_scope opts_scope(&c->opt);
    c->device = device;
    c->m = m; c->k = 1; c->mq = rup(m, GPZ_CH_NB); c->mp = rup(m + 1, 16);
    auto bail = [&](int code) { c->ar.release(); delete c; return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(GPZ_ERR_HIP, "hipSetDevice(%d) failed", device));
    int rc = alloc_mm(c);
    if (rc) return bail(rc);
    double *S = nullptr, *alpha0 = nullptr;
    if ((rc = c->ar.alloc(&S, (size_t)m * m))) return bail(rc);
    if ((rc = c->ar.alloc(&alpha0, (size_t)m))) return bail(rc);
    if ((rc = c->ar.alloc(&c->slab, (size_t)c->nsplit_l * c->mq * c->mq))) return bail(rc);
    if (hipMemcpy(S, Ain, (size_t)m * m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(GPZ_ERR_HIP, "copy failed"));
    (void)hipMemset(alpha0, 0, (size_t)m * sizeof(double));
    (void)hipMemset(c->info, 0, 2 * sizeof(int));
    const int mq = c->mq;
    launch_build_sigma(c->st, S, m, alpha0, m, mq, c->A, mq, nullptr, c->logdet);
    for (int k0 = 0; k0 < mq; k0 += GPZ_CH_NB) {
        launch_chol_step(c->st, c->A, c->Lm, mq, mq, k0, c->logdet, c->info);
    }
    launch_zero(c->st, c->Wm, (size_t)mq * mq);
    launch_trtri_diag(c->st, c->Lm, c->Wm, mq, mq);
    for (int gs = GPZ_CH_NB; gs < mq; gs *= 2) launch_trtri_level(c->st, c->Lm, c->Wm, c->Tmp, mq, mq, gs);
    launch_syrk(c->st, c->Wm, mq, nullptr, mq, mq, c->nsplit_l, c->rows_per_split_l, c->nsplit_l, c->rows_per_split_l, c->slab,
                    true);
    launch_syrk_reduce(c->st, c->slab, c->nsplit_l, c->nsplit_l, mq, c->Sinv, mq);
    launch_cond_flag(c->st, S, m, alpha0, c->Sinv, mq, m, c->Tmp, c->info);
    int info_h[2] = {0, 0};
    double ld = 0.0;
    hipError_t e = hipMemcpy(info_h, c->info, 2 * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
    int dropped = 0;
    if (info_h[1] != 0) {
        // numerically singular or not positive definite: the truncating SVD route of inv_logdet.m:3-15
        double *out3 = c->Tmp + mq + 8;
        if (run_jacobi_pinv(c->st, S, m, nullptr, m, c->A, c->Wm, mq, c->Tmp, (unsigned long long *)(c->Tmp + mq), c->Sinv, mq,
                            c->logdet, out3) < 0)
            return bail(fail(GPZ_ERR_HIP, "gpz_inv_logdet: Jacobi SVD failed"));
        double h3[3] = {0, 0, 0};
        e = hipMemcpy(h3, out3, sizeof h3, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
        dropped = m - (int)h3[1];
        info_h[0] = 0;
    }
    e = hipMemcpy2D(Xi, (size_t)m * sizeof(double), c->Sinv, (size_t)mq * sizeof(double), (size_t)m * sizeof(double), m,
                    hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&ld, c->logdet, sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
    if (info_h[0] != 0) {   // non-finite input
        for (size_t q = 0; q < (size_t)m * m; ++q) Xi[q] = NAN;
        ld = NAN;
        dropped = -1;
    }
    *logdet = ld;
    if (info) *info = dropped;
    c->ar.release();
    delete c;
    return GPZ_OK;
}

extern "C" int gpz_dxy(const double *X, int64_t nx, const double *Y, int64_t ny, int32_t d, int32_t device, double *D) {
    if (!X || !Y || !D || nx < 1 || ny < 1 || d < 1) return fail(GPZ_ERR_ARG, "gpz_dxy: bad argument");
    HIPCHK(hipSetDevice(device));
    Arena ar;
    double *dx = nullptr, *dy = nullptr, *dd = nullptr;
    int rc = ar.alloc(&dx, (size_t)nx * d);
    if (!rc) rc = ar.alloc(&dy, (size_t)ny * d);
    if (!rc) rc = ar.alloc(&dd, (size_t)nx * ny);
    if (!rc) {
        hipError_t e = hipMemcpy(dx, X, (size_t)nx * d * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dy, Y, (size_t)ny * d * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_dxy(nullptr, dx, nx, dy, ny, d, dd);
            e = hipMemcpy(D, dd, (size_t)nx * ny * sizeof(double), hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_dxy: %s", hipGetErrorString(e));
    }
    ar.release();
    return rc;
}

extern "C" int gpz_nan_groups(const double *X, int64_t n, int32_t d, int32_t device, int32_t *group_id, int32_t *n_groups) {
    if (!X || !group_id || !n_groups || n < 1 || d < 1) return fail(GPZ_ERR_ARG, "gpz_nan_groups: bad argument");
    HIPCHK(hipSetDevice(device));
    Arena ar;
    double *dx = nullptr;
    unsigned char *work = nullptr;
    int *ng = nullptr, *gid = nullptr;
    int rc = ar.alloc(&dx, (size_t)n * d);
    if (!rc) rc = ar.alloc(&work, nan_groups_work_bytes((long)n, d));
    if (!rc) rc = ar.alloc(&ng, (size_t)1);
    if (!rc) rc = ar.alloc(&gid, (size_t)n);
    if (!rc) {
        hipError_t e = hipMemcpy(dx, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_nan_groups(nullptr, dx, n, d, work, ng, gid);
            e = hipMemcpy(group_id, gid, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
        }
        int g = 0;
        if (e == hipSuccess) e = hipMemcpy(&g, ng, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(GPZ_ERR_HIP, "gpz_nan_groups: %s", hipGetErrorString(e));
        else *n_groups = g;
    }
    ar.release();
    return rc;
}

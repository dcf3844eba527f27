// Internal header of the host side of libgpz_hip.so: the evaluation context (struct gpz_ctx), its device-memory arena and the
// helpers the host translation units share.
//   gpz_arena.hip    error text, released-buffer cache, allocation test hook
//   gpz_ctx.hip      context creation / destruction, accessors, gpz_ctx_route
//   gpz_eval.hip     the evaluation pipeline: stage A (PHI, PHI'W PHI), the m x m stage, the tail; gpz_solve, gpz_get_phi
//   gpz_graph.hip    one evaluation as recorded hipGraph segments: recording, cuts, replay; gpz_eval / gpz_eval_dev
//   gpz_predict.hip  the stand-alone entry points (gpz_phi, gpz_predict_*, gpz_prior, gpz_inv_logdet, gpz_dxy, gpz_nan_groups)
#pragma once
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
#include <map>

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return gpz_fail(GPZ_ERR_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
    } while (0)

inline int method_id_of(const char *m) {
    static const char *names[6] = {"GL", "VL", "GD", "VD", "GC", "VC"};
    for (int i = 0; i < 6; ++i)
        if (m[0] == names[i][0] && m[1] == names[i][1]) return i;
    return -1;
}
inline int g_dim_of(int mid, int m, int d) {
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
inline int pad_dim(int d) {
    static const int sup[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20};
    for (int s : sup)
        if (d <= s) return s;
    return d;   // wider inputs: the runtime-d kernels of k_wide.hip, no padding
}
inline int rup(long v, int q) { return (int)(((v + q - 1) / q) * q); }

// device allocation bookkeeping.  Released blocks go to a per-device cache keyed by their exact size instead of back to the
// runtime: the stand-alone entry points (getPHI, predict*, prior ...) build and drop ~30 buffers per call, predict.m calls them
// once per NaN-pattern group, and hipFree (a device synchronisation + unmap, 38 us on average here) was 30 % of a 79-group
// predict() (profiles/README.md, round 3).  At most GPZ_CACHE_CAP bytes per device stay cached (blocks above GPZ_CACHE_BLOCK_MAX
// are freed directly: the limit sits just above the 2 GiB runtime-d workspace of d > 20, which a many-group predict() with
// missing values would otherwise allocate and free once per group); gpz_release_cached_memory() gives everything back.
#define GPZ_CACHE_CAP_DEFAULT (4096UL << 20)
#define GPZ_CACHE_BLOCK_MAX (2304UL << 20)
namespace gpzi {
struct DevCache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> blocks;   // (device, bytes) -> pointer
    std::map<int, size_t> held;                              // bytes cached per device
};
DevCache &dev_cache();
void *cache_take(int dev, size_t bytes);
bool cache_give(int dev, size_t bytes, void *p);
void cache_release_blocks();            // frees the cached blocks only (what a failed hipMalloc retries with)
bool alloc_fault_due();                 // test hook gpz_debug_fail_alloc
void pmc_model_cache_release_all();     // gpz_predict.hip: the model tables gpz_predict_missing keeps between NaN-pattern groups
}   // namespace gpzi
using namespace gpzi;

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
            if (e != hipSuccess) {   // then the model tables gpz_predict_missing keeps (entries in use - their mutex is held - are skipped)
                (void)hipGetLastError();
                pmc_model_cache_release_all();
                e = hipMalloc(&q, nb);
            }
            if (e != hipSuccess) return gpz_fail(GPZ_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", nb, hipGetErrorString(e));
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
    double *om = nullptr;   // n_pad, or k x n_pad for an n x k omega (nullptr => ones)
    double *xmu = nullptr;  // de: column means of the rows (missing entries count as 0): centre of k_small_tail's feature expansion
    double *Xs = nullptr;   // n_pad x xs_ld: rows [1 | x - xmu | 0], or [1 | (x - xmu) mk | mk | 0] with missing values (zero rows past n)
    int xs_ld = 0;          // (k_small_tail builds its features from these rows)
    long om_ld = 0;         // omega of output o, row i: om[o*om_ld + i]; 0 = one column for every output (GPz.m:48)
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
    struct Pending { int idx; hipEvent_t e0, e1; bool count_call; };
    std::vector<Pending> pending;
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
    float *Bext32 = nullptr;   // dtype f32: Bext rounded once per evaluation for the fp32-operand T-GEMM
    char *oz_A = nullptr, *oz_B = nullptr;   // GPZ_TGEMM_INT8: digit planes of PHI and of Bext (k_oz.hip)
    double *oz_cs = nullptr;                 // column scales of Bext
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
    bool syrk_small = false;   // mp <= 256, fp64 operands: PHI' W PHI by k_syrk_small (one workgroup holds the whole triangle)
    bool small_tail_dp = false;   // diagonal kinds + input noise, mp <= 256, k = 1: k_small_tail writes dPHI (into T's buffer), k_moments_diag sums it
    bool small_tail = false;   // mp <= 256, k = 1, no Psi / missing values / row tiles: T-GEMM + row scalars + moments as ONE kernel (k_small.hip), T never allocated
    int st_nwg = 0, st_nf = 0;
    double *st_slab = nullptr;
    bool fused = true;   // dPHI formed on the fly, output by output (no dPHI / dL matrices): k == 1, or k > 1 on the tuned kernels
    double *phipart = nullptr;   // PHI-build column-group partial sums (small row counts)
    int phipart_groups = 0;
    long phipart_rows = 0;       // rows one launch may have to use it (the stride is the launch's own n_pad)
    int nslots = 0;
    double *out_d = nullptr;
    double *out_h = nullptr, *theta_h = nullptr;   // pinned
    gpz_allreduce_fn ar_fn = nullptr;
    void *ar_user = nullptr;
    void *priv = nullptr;                 // owned by whoever attached it (the RCCL communicator of gpz_ctx_init_rccl),
    void (*priv_free)(void *) = nullptr;  // released with the context
    int timing = 0;   // gpz_ctx_enable_timing: 0 off, 1 HIP events around every stage (eager launches), 2 around the dominant stages only
    bool time_rest = false;   // level 3: the replay of level 2 with events around EVERY segment and around the all-reduce hooks as well, so that
                              // (wall time of a call) - (sum of all stage times) = what the device spent between segments: launch-to-launch gaps
    // One evaluation = ~150 launches on one stream between the upload of theta and the download of the result block, every argument
    // fixed for the life of the context: from the third gpz_eval on it is REPLAYED as hipGraphs (host theta, timing 0 or 2).  The
    // recording is cut into segments where something must happen between graph launches:
    //   * at the two exchange points of a sharded context (world > 1): the all-reduce hook runs eagerly between two segments - on
    //     whatever it is (RCCL inside the library, the loopback reducer of gpz_mgpu, a torch.distributed callback);
    //   * with timing = 2, before and after the dominant stages (PHI build, PHI'W PHI, T = PHI [inv|w], moments): HIP events recorded
    //     INSIDE a graph cannot be timed on ROCm 7 (tools/graph_event_probe.hip), events between graph launches can.
    // One set of segments per timing mode.  state: 0 first call (eager), 1 record on this call, 2 replay, -1 disabled (recording failed
    // / GPZ_NO_GRAPH).
    struct GraphSeg {
        hipGraphExec_t exec = nullptr;   // nullptr: nothing was recorded between two cuts
        int stage = -1;                  // >= 0: events around this segment, accumulated under this stage of the timer
        bool count_call = false;         // the first segment of a stage instance counts as its call
        double *hook_buf = nullptr;      // all-reduce after this segment
        size_t hook_count = 0;
    };
    struct GraphSet { std::vector<GraphSeg> segs; int state = 0; };
    GraphSet gset[2];                    // [0]: timing 0, [1]: timing 2
    GraphSet *cap = nullptr;             // the set being recorded
    int cap_stage = -1;                  // dominant stage open while recording (timing 2)
    bool cap_stage_first = false, cap_failed = false;
    hipStream_t graph_st = nullptr;   // the recording runs on a stream of its own (the null stream cannot be captured); the graphs are launched on st
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
namespace gpzi {
int graph_cut(gpz_ctx *c, bool last = false);   // gpz_eval.hip: close the segment being recorded (and open the next)
}
inline bool stage_is_dominant(const char *name) {
    return !strcmp(name, "tgemm") || !strcmp(name, "syrk") || !strcmp(name, "phi_build") || !strcmp(name, "moments") || !strcmp(name, "tail_small");
}
struct Stage {
    gpz_ctx *c;
    int idx = -1;
    bool cut = false;
    hipEvent_t e0{}, e1{};
    Stage(gpz_ctx *c_, const char *name) : c(c_) {
        if (c->capturing) {   // recording graph segments: a dominant stage becomes segments of its own, timed from outside on replay
            if (c->timing == 2 && stage_is_dominant(name) && c->cap_stage < 0) {
                gpzi::graph_cut(c);
                c->cap_stage = c->tm.find(name);
                c->cap_stage_first = true;
                cut = true;
            }
            return;
        }
        if (c->timing == 0 || (c->timing == 2 && !stage_is_dominant(name))) return;
        idx = c->tm.find(name);
        e0 = c->tm.get();
        e1 = c->tm.get();
        (void)hipEventRecord(e0, c->st);
    }
    ~Stage() {
        if (cut) {
            gpzi::graph_cut(c);
            c->cap_stage = -1;
            return;
        }
        if (idx < 0) return;
        (void)hipEventRecord(e1, c->st);
        c->tm.pending.push_back({idx, e0, e1, true});
    }
};
inline void collect_timings(gpz_ctx *c) {
    for (auto &pe : c->tm.pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
            c->tm.ms[pe.idx] += ms;
            if (pe.count_call) c->tm.calls[pe.idx] += 1;
        }
    }
    c->tm.pending.clear();
    c->tm.pool_used = 0;
}

// ---- shared between the host translation units ---------------------------------------------------
namespace gpzi {
int has_nan(const double *X, int64_t count);
int setup_model(gpz_ctx *c, const gpz_desc *desc);
int alloc_mm(gpz_ctx *c);   // m x m stage buffers
int setup_data(gpz_ctx *c, int64_t n_tot, const double *X, const double *Y, const double *Psi, int32_t psi_kind,
               const double *omega, const uint8_t *training, const uint8_t *validation, const uint8_t *patterns = nullptr,
               int32_t n_patterns = 0);
int build_phi(gpz_ctx *c);
int stage_a(gpz_ctx *c, const double *theta, const double *theta_dev = nullptr);   // gpz_eval.hip: unpack .. all-reduce 1
int eval_tail(gpz_ctx *c, bool pinv);                                                // gpz_eval.hip: the m x m stage .. result copy
}   // namespace gpzi

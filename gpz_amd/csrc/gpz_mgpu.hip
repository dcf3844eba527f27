// Native multi-GPU driver of libgpz_hip.so: one synchronous C call evaluates the objective and gradient on all GPUs of
// the node (SURVEY.md §8b "threading", §8e).  No Python, no torch: RCCL is called from inside the library.
//
// The reference calls  [f,g] = funObj(x)  from ONE MATLAB process and needs the result before its next statement
// (minFunc_2012/minFunc/minFunc.m:314, WolfeLineSearch.m:34,114,194; the closure is GPz/train.m:40).  gpz_mgpu_eval
// keeps that contract: the caller blocks, inside the library
//   * the training-selected rows (and the validation rows) are split into contiguous balanced blocks, one gpz_ctx per
//     device holding its block in HBM (created once, gpz_mgpu_create);
//   * one persistent host thread per device runs the single-device pipeline of gpz_ctx.hip on that device's own
//     stream.  One thread per device rather than one thread for all: an evaluation is 100-250 kernel launches per
//     device (the blocked Cholesky alone is ~100), a 125 000-row shard of c4 takes ~8 ms of GPU time, and a single
//     host thread feeding 8 devices would need ~2000 launches in that time — it would be the bottleneck;
//   * the two all-reduces of an evaluation ([PHI'W PHI | PHI'W y | sums] and the gradient records) are RCCL
//     ncclAllReduce calls on the per-device communicators of ncclCommInitAll, each issued by its device's thread on its
//     device's stream (the "one thread per device" usage of the NCCL API: no group call needed), in-place on the
//     library's own device buffers;
//   * every rank finishes the m-sized work redundantly (identical bits), rank 0's f, g, statistics are returned.
// Nothing n-sized ever leaves a device.
//
// RCCL is bound with dlopen at the first use instead of a link-time dependency: a process that already carries an RCCL
// (PyTorch-ROCm bundles one next to its own HIP runtime, without a SONAME) must not get a second one next to it, and a
// single-GPU caller needs none at all.  Search order: an RCCL already loaded into the process, then librccl.so.1 /
// librccl.so through the usual library path (/opt/rocm/lib via this library's runpath).
//
// reducer = GPZ_REDUCER_LOOPBACK replaces RCCL by an in-library reducer for shards that share ONE device (RCCL refuses
// duplicate devices in a communicator): it is how the whole sharded path — partitioning, threads, the two exchange
// points, redundant finish — is tested on single-GPU boxes.  It sums the ranks' buffers in rank order with one kernel,
// so results are bitwise reproducible.  The shards of a loopback handle share their device's stream: the host threads
// still run concurrently and meet at the exchange points, the device executes their kernels in arrival order.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpz_hip.h"
#include "gpz_kernels.h"
#include "gpz_mgpu_sync.h"

// ---- RCCL binding ------------------------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;          // (optional: what the communicator itself reports, gpz_*_comm_info)
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    std::string where;
};
static std::string g_rccl_why;
static const char *rccl_why() { return g_rccl_why.empty() ? "no diagnostic" : g_rccl_why.c_str(); }

static RcclApi *rccl_api() {
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    std::lock_guard<std::mutex> lk(mu);
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                      // an RCCL the process already carries (e.g. PyTorch's)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) { api.where = std::string(n) + " (already loaded)"; break; }
    if (!h)
        for (int q = 1; q >= 0 && !h; --q) {
            (void)dlerror();
            if ((h = dlopen(names[q], RTLD_NOW | RTLD_LOCAL))) api.where = names[q];
            else if (const char *e = dlerror()) g_rccl_why += std::string(g_rccl_why.empty() ? "" : "; ") + e;   // read ONCE: dlerror() clears itself
        }
    if (!h) return nullptr;
    g_rccl_why.clear();
#define BIND(f)                                                    \
    (void)dlerror();                                               \
    api.f = (decltype(api.f))dlsym(h, "nccl" #f);                  \
    if (!api.f) { const char *e = dlerror(); g_rccl_why = std::string("nccl" #f ": ") + (e ? e : "symbol not found"); return nullptr; }
    BIND(GetUniqueId) BIND(CommInitRank) BIND(CommInitAll) BIND(CommDestroy) BIND(CommAbort) BIND(AllReduce) BIND(GetErrorString)
#undef BIND
    api.CommCount = (decltype(api.CommCount))dlsym(h, "ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))dlsym(h, "ncclCommUserRank");
    api.CommCuDevice = (decltype(api.CommCuDevice))dlsym(h, "ncclCommCuDevice");
    api.handle = h;
    return &api;
}

// ---- multi-process use: one rank per process, communicator from a broadcast unique id ----------------------------
struct RankComm {
    ncclComm_t comm = nullptr;
};
static int rccl_hook(void *user, void *buf, size_t count, void *stream) {
    RankComm *rc = (RankComm *)user;
    RcclApi *api = rccl_api();
    if (!api || !rc->comm) return 1;
    return api->AllReduce(buf, buf, count, ncclDouble, ncclSum, rc->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}
static void rank_comm_free(void *p) {
    RankComm *rc = (RankComm *)p;
    RcclApi *api = rccl_api();
    if (api && rc->comm) (void)api->CommDestroy(rc->comm);
    delete rc;
}

extern "C" int gpz_rccl_unique_id(void *id128) {
    if (!id128) return gpz_fail(GPZ_ERR_ARG, "gpz_rccl_unique_id: null argument");
    RcclApi *api = rccl_api();
    if (!api) return gpz_fail(GPZ_ERR_COMM, "RCCL not found (librccl.so.1): %s", rccl_why());
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return gpz_fail(GPZ_ERR_COMM, "ncclGetUniqueId: %s", api->GetErrorString(r));
    static_assert(sizeof(id) == GPZ_RCCL_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return GPZ_OK;
}

extern "C" int gpz_ctx_init_rccl(gpz_ctx *ctx, const void *id128, int32_t rank, int32_t world, int32_t device) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return gpz_fail(GPZ_ERR_ARG, "gpz_ctx_init_rccl: bad argument");
    RcclApi *api = rccl_api();
    if (!api) return gpz_fail(GPZ_ERR_COMM, "RCCL not found (librccl.so.1): %s", rccl_why());
    if (hipSetDevice(device) != hipSuccess) return gpz_fail(GPZ_ERR_HIP, "hipSetDevice(%d) failed", device);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    RankComm *rc = new RankComm();
    ncclResult_t r = api->CommInitRank(&rc->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete rc;
        return gpz_fail(GPZ_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, api->GetErrorString(r));
    }
    gpz_ctx_attach_private(ctx, rc, rank_comm_free);
    return gpz_ctx_set_allreduce(ctx, rccl_hook, rc);
}

extern "C" int gpz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- single-process multi-device driver ---------------------------------------------------------------------------
// Barrier, AbortGate, CmdLoop: gpz_mgpu_sync.h (host-only, also compiled under ThreadSanitizer by the tests)
using gpz_sync::Barrier;

__global__ void k_loopback_sum(double *const *bufs, int nb, size_t count) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
        double s = bufs[0][e];
        for (int r = 1; r < nb; ++r) s += bufs[r][e];      // fixed rank order: bitwise reproducible
        for (int r = 0; r < nb; ++r) bufs[r][e] = s;
    }
}

struct gpz_mgpu;
struct RankSlot {
    gpz_mgpu *h = nullptr;
    int rank = 0;
    int exchange = 0;                              // all-reduces seen in the current command (1 = [PHI'W PHI | ...], 2 = gradient records)
    bool secondary = false;                        // this rank's failure is the echo of another rank's (poisoned barrier / aborted communicator)
};

struct gpz_mgpu {
    int n = 0, reducer = GPZ_REDUCER_RCCL;
    long p = 0;
    int m = 0, k = 1;
    std::vector<int> dev;
    std::vector<hipStream_t> streams;
    std::vector<gpz_ctx *> ctx;
    std::vector<ncclComm_t> comms;
    std::vector<RankSlot> slots;
    gpz_sync::CmdLoop core;                        // one persistent thread per rank; commands: 1 eval, 2 solve
    // per-rank results (rank 0's are handed to the caller)
    std::vector<double> f, stats, diag;            // n, 4n, 2n
    std::vector<std::vector<double>> g, w, iS, part;
    std::vector<std::string> err;
    double *out_w = nullptr, *out_iS = nullptr, *out_part = nullptr;
    // failure handling (RCCL reducer): a rank that fails before or at an exchange point would leave the others inside
    // ncclAllReduce for ever, so the failing rank's thread aborts every communicator of the handle (ncclCommAbort ends
    // the in-flight collectives) and the handle is dead from then on: later calls return GPZ_ERR_COMM.  Enqueues hold
    // the gate shared, the abort holds it exclusively, so no thread enqueues on a communicator that is being freed.
    gpz_sync::AbortGate gate;
    int inject_rank = -1, inject_exchange = 0;     // gpz_mgpu_debug_fail_at (tests)
    // loopback reducer
    Barrier bar;
    std::vector<double *> lb_ptr;
    double **lb_ptr_d = nullptr;
};

static int mgpu_hook(void *user, void *buf, size_t count, void *stream) {
    RankSlot *s = (RankSlot *)user;
    gpz_mgpu *h = s->h;
    ++s->exchange;
    if (h->inject_rank == s->rank && h->inject_exchange == s->exchange) {     // injected failure of this rank at this exchange point
        if (h->reducer == GPZ_REDUCER_LOOPBACK) h->bar.poison();
        return 1;
    }
    if (h->reducer == GPZ_REDUCER_RCCL) {
        RcclApi *api = rccl_api();
        int res = 1;
        const bool alive = h->gate.enqueue([&] {
            if (!h->comms[s->rank]) return -1;
            return api->AllReduce(buf, buf, count, ncclDouble, ncclSum, h->comms[s->rank], (hipStream_t)stream) == ncclSuccess ? 0 : 1;
        }, &res);
        if (!alive || res < 0) { s->secondary = true; return 1; }
        return res;
    }
    // loopback: every rank's contribution complete -> rank 0 sums in rank order into all buffers -> everyone continues
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { h->bar.poison(); return 1; }
    h->lb_ptr[s->rank] = (double *)buf;
    if (!h->bar.wait()) { s->secondary = true; return 1; }
    int ok = 0;
    if (s->rank == 0) {
        if (hipMemcpyAsync(h->lb_ptr_d, h->lb_ptr.data(), h->n * sizeof(double *), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
            ok = 1;
        int nb = (int)((count + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(k_loopback_sum, dim3(nb), dim3(256), 0, (hipStream_t)stream, h->lb_ptr_d, h->n, count);
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) ok = 1;
        if (ok) h->bar.poison();
    }
    if (!h->bar.wait()) { if (!ok) s->secondary = true; return 1; }
    return ok;
}

// RCCL reducer, after a failure on one rank: end every in-flight collective of the handle and mark it dead.
static void abort_comms(gpz_mgpu *h, int r, const char *why) {
    h->gate.abort(r, why, [&] {
        RcclApi *api = h->comms.empty() ? nullptr : rccl_api();
        for (ncclComm_t &c : h->comms) {
            if (api && c) (void)api->CommAbort(c);     // frees the communicator as well: never destroyed again
            c = nullptr;
        }
    });
}

// the work of rank r for one command (runs on the rank's own thread)
static int rank_command(gpz_mgpu *h, int r, int cmd, const double *theta) {
    int rc = 0;
    h->slots[r].exchange = 0;
    h->slots[r].secondary = false;
    if (cmd == 1)
        rc = gpz_eval(h->ctx[r], theta, &h->f[r], h->g[r].data(), &h->stats[4 * r], &h->diag[2 * r]);
    else if (cmd == 2)
        rc = gpz_solve(h->ctx[r], theta, r == 0 ? h->out_w : h->w[r].data(), r == 0 ? h->out_iS : h->iS[r].data(),
                       !h->out_part ? nullptr : (r == 0 ? h->out_part : h->part[r].data()));   // same branch on every rank
    if (rc) {
        h->err[r] = gpz_last_error();
        h->bar.poison();                        // loopback: do not leave the other ranks at an exchange point
        if (h->reducer == GPZ_REDUCER_RCCL && h->n > 1) abort_comms(h, r, h->err[r].c_str());   // RCCL: nor inside ncclAllReduce
    }
    return rc;
}

static int run_command(gpz_mgpu *h, int cmd, const double *theta) {
    if (h->gate.is_dead())
        return gpz_fail(GPZ_ERR_COMM, "this multi-GPU handle is dead: its communicators were aborted after a failure (%s); destroy it and "
                                      "create a new one (the MEX gateway does so on its next call)", h->gate.reason().c_str());
    h->core.submit(cmd, theta);
    h->bar.reset();
    h->inject_rank = -1;                            // an injected failure fires once
    for (int pass = 0; pass < 2; ++pass)           // the rank whose failure was not the echo of another rank's comes first
        for (int r = 0; r < h->n; ++r)
            if (h->core.rc[r] && (pass == 1 || !h->slots[r].secondary))
                return gpz_fail(h->core.rc[r], "rank %d (device %d): %s", r, h->dev[r], h->err[r].c_str());
    return GPZ_OK;
}

static void mgpu_free(gpz_mgpu *h) {
    h->core.stop();
    for (int r = 0; r < (int)h->ctx.size(); ++r)
        if (h->ctx[r]) gpz_ctx_destroy(h->ctx[r]);
    RcclApi *api = h->comms.empty() ? nullptr : rccl_api();
    for (ncclComm_t c : h->comms)
        if (api && c) (void)api->CommDestroy(c);
    for (int r = 0; r < (int)h->streams.size(); ++r)
        if (h->streams[r] && (r == 0 || h->streams[r] != h->streams[0])) {
            (void)hipSetDevice(h->dev[r]);
            (void)hipStreamDestroy(h->streams[r]);
        }
    if (h->lb_ptr_d) {
        (void)hipSetDevice(h->dev[0]);
        (void)hipFree(h->lb_ptr_d);
    }
    delete h;
}

static inline void shard_bounds(int64_t n, int r, int w, int64_t *lo, int64_t *hi) {
    *lo = (int64_t)r * n / w;
    *hi = (int64_t)(r + 1) * n / w;
}

extern "C" int gpz_mgpu_create(const gpz_desc *desc, int32_t n_gpus, const int32_t *devices, int32_t reducer, int64_t n_tot,
                               const double *X, const double *Y, const double *Psi, int32_t psi_kind, const double *omega,
                               const uint8_t *training, const uint8_t *validation, gpz_mgpu **out) {
    if (!desc || !X || !Y || !out || n_tot < 1) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_create: null argument");
    *out = nullptr;
    if (reducer != GPZ_REDUCER_RCCL && reducer != GPZ_REDUCER_LOOPBACK) return gpz_fail(GPZ_ERR_ARG, "unknown reducer %d", reducer);
    if ((Psi != nullptr) != (psi_kind != 0)) return gpz_fail(GPZ_ERR_ARG, "Psi and psi_kind disagree");
    const int ndev = gpz_device_count();
    if (ndev < 1) return gpz_fail(GPZ_ERR_HIP, "no HIP device");
    if (n_gpus <= 0) n_gpus = ndev;                                   // default: every GPU of the node
    const int d = desc->d, k = desc->k;
    if (d < 1 || k < 1 || desc->m < 1) return gpz_fail(GPZ_ERR_ARG, "d, m, k must be >= 1");
    if (desc->omega_cols > 1 && desc->omega_cols != k)
        return gpz_fail(GPZ_ERR_ARG, "omega must be n x 1 or n x k (omega_cols = %d, k = %d)", desc->omega_cols, k);
    gpz_mgpu *h = new gpz_mgpu();
    h->n = n_gpus;
    h->reducer = reducer;
    h->dev.resize(n_gpus);
    for (int r = 0; r < n_gpus; ++r) h->dev[r] = devices ? devices[r] : (reducer == GPZ_REDUCER_LOOPBACK ? 0 : r);
    for (int r = 0; r < n_gpus; ++r) {
        if (h->dev[r] < 0 || h->dev[r] >= ndev) { mgpu_free(h); return gpz_fail(GPZ_ERR_ARG, "device %d not present (%d devices)", h->dev[r], ndev); }
        if (reducer == GPZ_REDUCER_LOOPBACK && h->dev[r] != h->dev[0]) { mgpu_free(h); return gpz_fail(GPZ_ERR_ARG, "the loopback reducer needs all shards on one device"); }
        for (int q = 0; q < r; ++q)
            if (reducer == GPZ_REDUCER_RCCL && h->dev[q] == h->dev[r]) { mgpu_free(h); return gpz_fail(GPZ_ERR_ARG, "RCCL needs distinct devices (device %d listed twice); use GPZ_REDUCER_LOOPBACK to test on one GPU", h->dev[r]); }
    }
    // rows the evaluation reads: X(training,:) and X(validation,:)   (getPHI.m:14, GPz.m:243)
    std::vector<int64_t> it, iv;
    for (int64_t i = 0; i < n_tot; ++i) {
        if (!training || training[i]) it.push_back(i);
        if (validation && validation[i]) iv.push_back(i);
    }
    if ((int64_t)it.size() < n_gpus) { mgpu_free(h); return gpz_fail(GPZ_ERR_ARG, "%lld training rows cannot be split over %d GPUs", (long long)it.size(), n_gpus); }
    const bool use_valid = !iv.empty();
    // GC/VC with missing values: the NaN-pattern table of the whole data set, first-occurrence order (getPHI.m:43-54)
    std::vector<uint8_t> pats;
    int npat = 0;
    const bool cov = desc->method[1] == 'C';
    if (cov) {
        bool any = false;
        std::vector<std::vector<uint8_t>> tab;
        auto scan = [&](int64_t i) {
            std::vector<uint8_t> pt((size_t)d);
            for (int c = 0; c < d; ++c) { const double xv = X[(size_t)c * n_tot + i]; pt[c] = xv != xv ? 1 : 0; if (xv != xv) any = true; }
            for (auto &q : tab) if (q == pt) return;
            tab.push_back(pt);
        };
        for (int64_t i = 0; i < n_tot; ++i)
            if ((!training || training[i]) || (validation && validation[i])) scan(i);
        if (any) {
            npat = (int)tab.size();
            for (auto &q : tab) pats.insert(pats.end(), q.begin(), q.end());
        }
    }
    h->streams.assign(n_gpus, nullptr);
    h->ctx.assign(n_gpus, nullptr);
    int rc = 0;
    for (int r = 0; r < n_gpus && !rc; ++r) {
        int64_t lo, hi, lv = 0, hv = 0;
        shard_bounds((int64_t)it.size(), r, n_gpus, &lo, &hi);
        if (use_valid) shard_bounds((int64_t)iv.size(), r, n_gpus, &lv, &hv);
        const int64_t nt = hi - lo, nr = nt + (hv - lv);
        std::vector<int64_t> rows((size_t)nr);
        for (int64_t q = 0; q < nt; ++q) rows[q] = it[lo + q];
        for (int64_t q = 0; q < hv - lv; ++q) rows[nt + q] = iv[lv + q];
        std::vector<double> Xr((size_t)nr * d), Yr((size_t)nr * k), Or, Pr;
        for (int c = 0; c < d; ++c)
            for (int64_t q = 0; q < nr; ++q) Xr[(size_t)c * nr + q] = X[(size_t)c * n_tot + rows[q]];
        for (int o = 0; o < k; ++o)
            for (int64_t q = 0; q < nr; ++q) Yr[(size_t)o * nr + q] = Y[(size_t)o * n_tot + rows[q]];
        if (omega) {
            const int oc = desc->omega_cols > 1 ? desc->omega_cols : 1;   // n_tot x 1 or n_tot x k (GPz.m:48)
            Or.resize((size_t)nr * oc);
            for (int o = 0; o < oc; ++o)
                for (int64_t q = 0; q < nr; ++q) Or[(size_t)o * nr + q] = omega[(size_t)o * n_tot + rows[q]];
        }
        if (psi_kind == 1 || psi_kind == 3) {
            Pr.resize((size_t)nr * d);
            for (int c = 0; c < d; ++c)
                for (int64_t q = 0; q < nr; ++q) Pr[(size_t)c * nr + q] = Psi[(size_t)c * n_tot + rows[q]];
        } else if (psi_kind == 2) {
            Pr.resize((size_t)nr * d * d);
            for (int64_t q = 0; q < nr; ++q) memcpy(&Pr[(size_t)q * d * d], Psi + (size_t)rows[q] * d * d, (size_t)d * d * sizeof(double));
        }
        std::vector<uint8_t> tr((size_t)nr, 0), va;
        for (int64_t q = 0; q < nt; ++q) tr[q] = 1;
        if (use_valid) {
            va.assign((size_t)nr, 0);
            for (int64_t q = nt; q < nr; ++q) va[q] = 1;
        }
        // one stream per DEVICE: loopback shards share their device's stream (and with it one hardware queue and one
        // scratch allocation — the run-time-d kernels of k_gen.hip need up to 26 KB of scratch per lane, and several
        // queues of one device asking for that at the same moment exhaust the runtime's scratch pool)
        if (reducer == GPZ_REDUCER_LOOPBACK && r > 0) h->streams[r] = h->streams[0];
        else if (hipSetDevice(h->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&h->streams[r], hipStreamNonBlocking) != hipSuccess) {
            rc = gpz_fail(GPZ_ERR_HIP, "stream creation on device %d failed", h->dev[r]);
            break;
        }
        gpz_desc dr = *desc;
        dr.device = h->dev[r];
        dr.stream = (void *)h->streams[r];
        dr.rank = r;
        dr.world = n_gpus;
        rc = gpz_ctx_create_sharded(&dr, nr, Xr.data(), Yr.data(), psi_kind ? Pr.data() : nullptr, psi_kind,
                                    omega ? Or.data() : nullptr, tr.data(), use_valid ? va.data() : nullptr,
                                    npat ? pats.data() : nullptr, npat, &h->ctx[r]);
    }
    if (rc) {
        std::string msg = gpz_last_error();
        mgpu_free(h);
        return gpz_fail(rc, "%s", msg.c_str());
    }
    h->p = (long)gpz_theta_len(h->ctx[0]);
    h->m = desc->m;
    h->k = k;
    if (n_gpus > 1 && reducer == GPZ_REDUCER_RCCL) {
        RcclApi *api = rccl_api();
        if (!api) { mgpu_free(h); return gpz_fail(GPZ_ERR_COMM, "RCCL not found (librccl.so.1): %s", rccl_why()); }
        h->comms.assign(n_gpus, nullptr);
        ncclResult_t r = api->CommInitAll(h->comms.data(), n_gpus, h->dev.data());
        if (r != ncclSuccess) {
            h->comms.clear();
            mgpu_free(h);
            return gpz_fail(GPZ_ERR_COMM, "ncclCommInitAll over %d devices: %s", n_gpus, api->GetErrorString(r));
        }
    }
    if (reducer == GPZ_REDUCER_LOOPBACK) {
        h->lb_ptr.assign(n_gpus, nullptr);
        (void)hipSetDevice(h->dev[0]);
        if (hipMalloc((void **)&h->lb_ptr_d, n_gpus * sizeof(double *)) != hipSuccess) { mgpu_free(h); return gpz_fail(GPZ_ERR_ALLOC, "hipMalloc failed"); }
    }
    h->bar.n = n_gpus;
    h->slots.resize(n_gpus);
    h->f.assign(n_gpus, 0.0);
    h->stats.assign(4 * n_gpus, 0.0);
    h->diag.assign(2 * n_gpus, 0.0);
    h->g.resize(n_gpus); h->w.resize(n_gpus); h->iS.resize(n_gpus); h->part.resize(n_gpus);
    h->err.resize(n_gpus);
    for (int r = 0; r < n_gpus; ++r) {
        h->slots[r].h = h;
        h->slots[r].rank = r;
        h->g[r].resize((size_t)h->p);
        if (r > 0) {   // the other ranks' copies of the solve outputs land in scratch
            h->w[r].resize((size_t)h->m * k);
            h->iS[r].resize((size_t)h->m * h->m * k);
            h->part[r].resize((size_t)k);
        }
        if (n_gpus > 1) (void)gpz_ctx_set_allreduce(h->ctx[r], mgpu_hook, &h->slots[r]);
    }
    h->core.init = [h](int r) { (void)hipSetDevice(h->dev[r]); };
    h->core.run = [h](int r, int cmd, const void *arg) { return rank_command(h, r, cmd, (const double *)arg); };
    h->core.start(n_gpus);
    *out = h;
    return GPZ_OK;
}

extern "C" void gpz_mgpu_destroy(gpz_mgpu *h) {
    if (h) mgpu_free(h);
}

extern "C" int gpz_mgpu_eval(gpz_mgpu *h, const double *theta, double *f, double *g, double stats[4], double diag[2]) {
    if (!h || !theta || !f || !g) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_eval: null argument");
    for (int r = 0; r < h->n; ++r)
        for (int q = 0; q < 4; ++q) h->stats[4 * r + q] = stats ? stats[q] : 0.0;     // slots 2, 3 stay untouched without validation
    if (int e = run_command(h, 1, theta)) return e;
    *f = h->f[0];
    memcpy(g, h->g[0].data(), (size_t)h->p * sizeof(double));
    if (stats) memcpy(stats, &h->stats[0], 4 * sizeof(double));
    if (diag) { diag[0] = h->diag[0]; diag[1] = h->diag[1]; }
    return GPZ_OK;
}

extern "C" int gpz_mgpu_solve(gpz_mgpu *h, const double *theta, double *w, double *iSigma_w, double *nlogML_partial) {
    if (!h || !theta || !w || !iSigma_w) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_solve: null argument");
    h->out_w = w;
    h->out_iS = iSigma_w;
    h->out_part = nlogML_partial;
    return run_command(h, 2, theta);
}

// ---- prediction over several GPUs ------------------------------------------------------------------------------------------
// The rows of one NaN-pattern group are independent (predict.m:60-69 calls predictDiag / predictCov per group; inside, every
// sample is a row of PHI and a row of the pairwise sums): contiguous row blocks go to the devices, each block through the
// single-device entry its content selects (predictFull / predictNoisy / predictMissing / predictNoisyMissing, the choice of
// predictDiag.m:39-55), one host thread per block.  Results are those of the single-device call, row for row.
extern "C" int gpz_mgpu_predict(const gpz_desc *desc, int32_t n_gpus, const int32_t *devices, const double *theta, const double *w,
                                const double *iSigma_w, const double *priors, const double *Xs, int64_t ns, const double *Psi,
                                int32_t psi_kind, double *mu, double *nu, double *beta_i, double *gamma, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !Xs || ns < 1 || !mu || !nu || !beta_i)
        return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_predict: null argument");
    if ((Psi != nullptr) != (psi_kind != 0)) return gpz_fail(GPZ_ERR_ARG, "Psi and psi_kind disagree");
    const int ndev = gpz_device_count();
    if (ndev < 1) return gpz_fail(GPZ_ERR_HIP, "no HIP device");
    if (n_gpus <= 0) n_gpus = ndev;
    const int d = desc->d, k = desc->k, m = desc->m;
    bool miss = false;
    for (int c = 0; c < d && !miss; ++c) { const double xv = Xs[(size_t)c * ns]; miss = xv != xv; }   // the group's pattern: first row (predictDiag.m:3)
    if (miss && (!priors || !gamma)) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_predict: missing values need priors and gamma");
    if (psi_kind && !gamma) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_predict: input noise needs gamma");
    // A block costs a thread, a temporary context on its device and an upload of theta, w and iSigma_w (m*m*k doubles): only
    // worth it for a few thousand rows.  NaN-pattern groups are often a few dozen rows (predict.m:45-57) — those run as ONE
    // block on the first device.  An explicit device list keeps the caller's split (tests).
    const gpz_options call_opts = gpz_options_load();   // (no context: a snapshot per call)
    const int64_t min_rows = call_opts.predict_min_rows_per_block;   // GPZ_PREDICT_MIN_ROWS_PER_BLOCK, default 4096 (tests: split small groups too)
    int nblk = n_gpus;
    if (!devices) nblk = (int)std::min<int64_t>(n_gpus, std::max<int64_t>(1, ns / min_rows));
    if ((int64_t)nblk > ns) nblk = (int)ns;
    std::vector<int> dev(nblk);
    for (int r = 0; r < nblk; ++r) {
        dev[r] = devices ? devices[r] : r % ndev;
        if (dev[r] < 0 || dev[r] >= ndev) return gpz_fail(GPZ_ERR_ARG, "device %d not present (%d devices)", dev[r], ndev);
    }
    std::vector<int> rc(nblk, 0);
    std::vector<std::string> err(nblk);
    auto work = [&](int r) {
        int64_t lo, hi;
        shard_bounds(ns, r, nblk, &lo, &hi);
        const int64_t nr = hi - lo;
        std::vector<double> Xb((size_t)nr * d), Pb, o_mu((size_t)nr * k), o_nu((size_t)nr * k), o_be((size_t)nr * k), o_ga((size_t)nr * k), o_phi;
        for (int c = 0; c < d; ++c) memcpy(&Xb[(size_t)c * nr], Xs + (size_t)c * ns + lo, (size_t)nr * sizeof(double));
        if (psi_kind == 1 || psi_kind == 3) {
            Pb.resize((size_t)nr * d);
            for (int c = 0; c < d; ++c) memcpy(&Pb[(size_t)c * nr], Psi + (size_t)c * ns + lo, (size_t)nr * sizeof(double));
        } else if (psi_kind == 2) {
            Pb.assign(Psi + (size_t)lo * d * d, Psi + (size_t)hi * d * d);
        }
        if (PHI) o_phi.resize((size_t)nr * m);
        gpz_desc dr = *desc;
        dr.device = dev[r];
        dr.stream = nullptr;
        int e;
        if (miss)
            e = gpz_predict_missing(&dr, theta, w, iSigma_w, priors, Xb.data(), nr, psi_kind ? Pb.data() : nullptr, psi_kind,
                                    o_mu.data(), o_nu.data(), o_be.data(), o_ga.data(), PHI ? o_phi.data() : nullptr);
        else if (psi_kind)
            e = gpz_predict_noisy(&dr, theta, w, iSigma_w, Xb.data(), nr, Pb.data(), psi_kind, o_mu.data(), o_nu.data(),
                                  o_be.data(), o_ga.data(), PHI ? o_phi.data() : nullptr);
        else {
            e = gpz_predict_full(&dr, theta, w, iSigma_w, Xb.data(), nr, o_mu.data(), o_nu.data(), o_be.data(),
                                 PHI ? o_phi.data() : nullptr);
            std::fill(o_ga.begin(), o_ga.end(), 0.0);                       // predictFull has no gamma term
        }
        rc[r] = e;
        if (e) { err[r] = gpz_last_error(); return; }
        for (int o = 0; o < k; ++o) {
            memcpy(mu + (size_t)o * ns + lo, &o_mu[(size_t)o * nr], (size_t)nr * sizeof(double));
            memcpy(nu + (size_t)o * ns + lo, &o_nu[(size_t)o * nr], (size_t)nr * sizeof(double));
            memcpy(beta_i + (size_t)o * ns + lo, &o_be[(size_t)o * nr], (size_t)nr * sizeof(double));
            if (gamma) memcpy(gamma + (size_t)o * ns + lo, &o_ga[(size_t)o * nr], (size_t)nr * sizeof(double));
        }
        if (PHI)
            for (int j = 0; j < m; ++j) memcpy(PHI + (size_t)j * ns + lo, &o_phi[(size_t)j * nr], (size_t)nr * sizeof(double));
    };
    if (nblk == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < nblk; ++r) th.emplace_back(work, r);
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < nblk; ++r)
        if (rc[r]) return gpz_fail(rc[r], "block %d (device %d): %s", r, dev[r], err[r].c_str());
    return GPZ_OK;
}

extern "C" int gpz_mgpu_debug_fail_at(gpz_mgpu *h, int32_t rank, int32_t exchange) {
    if (!h || rank < 0 || rank >= h->n || exchange < 1 || exchange > 2) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_debug_fail_at: bad argument");
    if (h->n < 2) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_debug_fail_at: a one-shard handle has no exchange point");
    h->inject_rank = rank;
    h->inject_exchange = exchange;
    return GPZ_OK;
}
extern "C" int32_t gpz_mgpu_alive(const gpz_mgpu *h) { return (h && !h->gate.is_dead()) ? 1 : 0; }
extern "C" int32_t gpz_mgpu_size(const gpz_mgpu *h) { return h ? h->n : -1; }
extern "C" int64_t gpz_mgpu_theta_len(const gpz_mgpu *h) { return h ? h->p : -1; }
extern "C" gpz_ctx *gpz_mgpu_ctx(gpz_mgpu *h, int32_t rank) { return (h && rank >= 0 && rank < h->n) ? h->ctx[rank] : nullptr; }
// What the communicator behind an all-reduce saw of itself (a first real multi-GPU run proves N ranks on N devices from its own output).
static int comm_info(ncclComm_t comm, int device, int32_t info[4], char *bus_id, int32_t cap) {
    info[0] = info[1] = info[2] = -1;
    info[3] = device;
    RcclApi *api = comm ? rccl_api() : nullptr;
    if (api) {
        int v = -1;
        if (api->CommCount && api->CommCount(comm, &v) == ncclSuccess) info[0] = v;
        if (api->CommUserRank && api->CommUserRank(comm, &v) == ncclSuccess) info[1] = v;
        if (api->CommCuDevice && api->CommCuDevice(comm, &v) == ncclSuccess) info[2] = v;
    }
    if (bus_id && cap > 0) {
        bus_id[0] = 0;
        if (hipDeviceGetPCIBusId(bus_id, cap, device) != hipSuccess) { (void)hipGetLastError(); bus_id[0] = 0; }
    }
    return GPZ_OK;
}
extern "C" int gpz_ctx_comm_info(const gpz_ctx *ctx, int32_t info[4], char *bus_id, int32_t cap) {
    if (!ctx || !info) return gpz_fail(GPZ_ERR_ARG, "gpz_ctx_comm_info: null argument");
    void *user = nullptr;
    const bool ours = gpz_ctx_allreduce_is(ctx, rccl_hook, &user);   // the communicator of gpz_ctx_init_rccl, if that is the hook in place
    return comm_info(ours && user ? ((RankComm *)user)->comm : nullptr, gpz_ctx_device(ctx), info, bus_id, cap);
}
extern "C" int gpz_mgpu_comm_info(const gpz_mgpu *h, int32_t rank, int32_t info[4], char *bus_id, int32_t cap) {
    if (!h || !info || rank < 0 || rank >= h->n) return gpz_fail(GPZ_ERR_ARG, "gpz_mgpu_comm_info: bad argument");
    return comm_info(h->reducer == GPZ_REDUCER_RCCL && rank < (int)h->comms.size() ? h->comms[rank] : nullptr, h->dev[rank], info, bus_id, cap);
}
extern "C" const char *gpz_rccl_origin(void) {
    RcclApi *api = rccl_api();
    return api ? api->where.c_str() : "";
}

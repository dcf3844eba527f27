// Host side of libgpz_hip.so, part 1 of 4 (gpz_ctx.h): error text, the released-buffer cache behind the arena, the allocation test hook.
#include "gpz_ctx.h"

// Error text.  The return code of an entry point is the authority; this is its text.  Every failure is stamped with a process-wide
// sequence number: gpz_last_error() returns the calling thread's own last failure unless a failure on ANOTHER thread is newer (work
// that failed on one of the library's worker threads after this thread's last own failure), then that one.  A thread that has never
// failed itself gets the newest failure of the process, or "" when there has been none.
static thread_local std::string g_err;
static thread_local unsigned long long g_err_seq = 0;
static std::mutex g_err_any_mu;
static std::string g_err_any;
static unsigned long long g_err_any_seq = 0;
static void set_error(const char *text) {
    g_err = text;
    std::lock_guard<std::mutex> g(g_err_any_mu);
    g_err_any = text;
    g_err_seq = ++g_err_any_seq;
}

extern "C" const char *gpz_last_error(void) {
    static thread_local std::string other;   // (a copy: the shared text may change under the caller)
    {
        std::lock_guard<std::mutex> g(g_err_any_mu);
        if (g_err_any_seq <= g_err_seq) return g_err.c_str();
        other = g_err_any;
    }
    return other.c_str();
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

namespace gpzi {
// GPZ_CACHE_CAP_MB (environment, read once): bytes per device that may stay cached, for hosts that share the GPU with another
// allocator (PyTorch's, a second process); 0 = no caching at all.  INTEGRATION.md, "device memory".
static size_t cache_cap() {
    static const size_t cap = [] {
        const long mb = gpz_options_load().cache_cap_mb;
        return mb >= 0 ? (size_t)mb << 20 : (size_t)GPZ_CACHE_CAP_DEFAULT;
    }();
    return cap;
}
DevCache &dev_cache() {
    static DevCache *c = new DevCache();   // never destroyed: the HIP runtime may be gone before static destructors run
    return *c;
}
void *cache_take(int dev, size_t bytes) {
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.blocks.find({dev, bytes});
    if (it == c.blocks.end()) return nullptr;
    void *p = it->second;
    c.blocks.erase(it);
    c.held[dev] -= bytes;
    return p;
}
bool cache_give(int dev, size_t bytes, void *p) {
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
void cache_release_blocks() {
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
}   // namespace gpzi
extern "C" void gpz_release_cached_memory(void) {
    cache_release_blocks();
    pmc_model_cache_release_all();   // after the block cache's lock is gone; entries a running prediction holds are skipped
}
namespace gpzi {
// test hook (gpz_debug_fail_alloc(k), include/gpz_hip.h): the k-th hipMalloc from now on reports out-of-memory once, so the retry
// path can be exercised without exhausting 288 GB
static std::atomic<long> g_alloc_fault_countdown{0};
}   // namespace gpzi
extern "C" void gpz_debug_fail_alloc(int64_t kth) { g_alloc_fault_countdown.store(kth > 0 ? (long)kth : 0); }
namespace gpzi {
bool alloc_fault_due() {
    if (g_alloc_fault_countdown.load(std::memory_order_relaxed) <= 0) return false;
    return g_alloc_fault_countdown.fetch_sub(1) == 1;
}

}   // namespace gpzi
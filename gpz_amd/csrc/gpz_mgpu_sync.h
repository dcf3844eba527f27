// Host-side synchronisation core of the single-process multi-device driver (gpz_mgpu.hip): command hand-off to one persistent
// thread per rank, the poisonable barrier of the loopback reducer, and the gate that keeps enqueues and the abort of the
// communicators apart.  No HIP, no RCCL, no gpz types: the per-rank work and what "abort" does are callbacks, so this file also
// compiles host-only with a stub rank function under ThreadSanitizer (tests/stubs/mgpu_sync_tsan.cpp, SURVEY.md section 5
// "race detection").
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

namespace gpz_sync {

// A barrier that can be poisoned: a rank that fails before an exchange point releases the others with an error instead of
// leaving them waiting.
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    unsigned long gen = 0;
    bool poisoned = false;
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (poisoned) return false;
        const unsigned long g = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return gen != g || poisoned; });
        // a rank released by the generation change has passed the barrier even if another rank poisoned it afterwards
        return gen != g;
    }
    void poison() {
        std::lock_guard<std::mutex> lk(mu);
        poisoned = true;
        cv.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> lk(mu);
        poisoned = false;
        waiting = 0;
    }
};

// Enqueues on a rank's communicator hold the gate shared; the abort after a failure holds it exclusively and marks the handle
// dead, so no thread enqueues on a communicator that is being freed and the abort runs once.
struct AbortGate {
    std::shared_mutex mu;
    std::atomic<bool> dead{false};
    std::string why;                                   // written once, under the exclusive lock, before `dead` is published
    // f() runs under the shared lock unless the handle is dead; returns false without calling it in that case
    template <typename F>
    bool enqueue(F &&f, int *result) {
        std::shared_lock<std::shared_mutex> lk(mu);
        if (dead.load(std::memory_order_acquire)) return false;
        *result = f();
        return true;
    }
    // the first caller frees everything (free_all runs under the exclusive lock); later callers return at once
    template <typename F>
    void abort(int rank, const char *reason, F &&free_all) {
        std::unique_lock<std::shared_mutex> lk(mu);
        if (dead.load(std::memory_order_relaxed)) return;
        why = std::string("rank ") + std::to_string(rank) + ": " + (reason ? reason : "");
        free_all();
        dead.store(true, std::memory_order_release);
    }
    bool is_dead() const { return dead.load(std::memory_order_acquire); }
    std::string reason() {
        std::shared_lock<std::shared_mutex> lk(mu);
        return why;
    }
};

// One persistent thread per rank; submit() hands one command to all of them and returns when every rank has finished it.
// run(rank, cmd, arg) is the per-rank work (it handles its own failure: poisons the barrier / aborts the gate); init(rank)
// runs once at the start of the rank's thread (device selection).
struct CmdLoop {
    enum { QUIT = -1 };
    int n = 0;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned long gen = 0;
    int cmd = 0, pending = 0;
    const void *arg = nullptr;
    std::vector<std::thread> workers;
    std::vector<int> rc;                               // per rank, of the last command (written by the rank, read after submit returns)
    std::function<void(int)> init;
    std::function<int(int, int, const void *)> run;

    void start(int nranks) {
        n = nranks;
        rc.assign(n, 0);
        for (int r = 0; r < n; ++r) workers.emplace_back([this, r] { loop(r); });
    }
    void loop(int r) {
        if (init) init(r);
        unsigned long seen = 0;
        for (;;) {
            int c;
            const void *a;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return gen != seen; });
                seen = gen;
                c = cmd;
                a = arg;
            }
            if (c == QUIT) return;
            const int res = run(r, c, a);
            {
                std::lock_guard<std::mutex> lk(mu);
                rc[r] = res;                           // under the lock that submit() takes to read `pending`: ordered before its return
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void submit(int c, const void *a) {
        {
            std::lock_guard<std::mutex> lk(mu);
            cmd = c;
            arg = a;
            pending = n;
            ++gen;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    void stop() {
        if (workers.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            cmd = QUIT;
            ++gen;
        }
        cv_go.notify_all();
        for (auto &t : workers) t.join();
        workers.clear();
    }
};

}   // namespace gpz_sync

// ThreadSanitizer harness for the synchronisation core of the multi-device driver (gpz_amd/csrc/gpz_mgpu_sync.h, the file
// gpz_mgpu.hip builds on): CmdLoop + poisonable Barrier (loopback reducer) + AbortGate (RCCL reducer) with a STUB rank function.
// TEST INFRASTRUCTURE - host-only, no HIP.  The stub does what gpz_eval does at its two exchange points:
//   loopback mode   publish a pointer, barrier, rank 0 "sums", barrier (mgpu_hook's loopback branch)
//   gate mode       enqueue on a fake communicator through AbortGate; the fake collective blocks until every rank has arrived or
//                   the communicator is aborted (what ncclAllReduce + ncclCommAbort do); a failing rank aborts all of them
// with failures injected at random (rank, exchange point).  Checked besides TSAN's own report: no hang (the caller runs this under
// a timeout), every command returns, the failing rank is reported, sums are right when nothing failed, a loopback handle keeps
// working after a failure, a gate handle is dead after one.
//   (build line: tests/test_sanitizers.py)
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "gpz_mgpu_sync.h"

using namespace gpz_sync;

struct FakeComm {                         // one "collective" at a time, in two halves like ncclAllReduce on a stream:
    std::mutex mu;                        //   enqueue()  returns at once (called under the gate's shared lock)
    std::condition_variable cv;           //   wait()     blocks until n ranks have enqueued or the communicator is aborted (the
    int n = 0, arrived = 0;               //              hipStreamSynchronize that follows, outside the gate)
    unsigned long gen = 0;
    bool aborted = false;
    double acc = 0.0, result = 0.0;
    int enqueue(double v, unsigned long *ticket) {
        std::lock_guard<std::mutex> lk(mu);
        if (aborted) return 1;
        *ticket = gen;
        acc += v;
        if (++arrived == n) {
            result = acc;
            acc = 0.0;
            arrived = 0;
            ++gen;
            cv.notify_all();
        }
        return 0;
    }
    int wait(unsigned long ticket, double *v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return gen != ticket || aborted; });
        if (gen == ticket) return 1;      // released by the abort
        *v = result;
        return 0;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

struct Handle {
    int n = 0;
    bool gate_mode = false;
    CmdLoop core;
    Barrier bar;
    AbortGate gate;
    FakeComm comm;
    bool comm_alive = true;                  // guarded by the gate (freed under its exclusive lock)
    std::vector<double *> ptr;               // loopback: each rank's buffer, published before the first barrier
    std::vector<double> val, out;
    std::vector<char> secondary;
    int inject_rank = -1, inject_exchange = 0;
};

static int exchange(Handle *h, int r, int which, double *v) {
    if (h->inject_rank == r && h->inject_exchange == which) {
        if (!h->gate_mode) h->bar.poison();
        return 1;
    }
    if (h->gate_mode) {
        int res = 1;
        unsigned long ticket = 0;
        const bool alive = h->gate.enqueue([&] { return h->comm_alive ? h->comm.enqueue(*v, &ticket) : -1; }, &res);
        if (!alive || res || h->comm.wait(ticket, v)) { h->secondary[r] = 1; return 1; }
        return 0;
    }
    h->ptr[r] = v;
    if (!h->bar.wait()) { h->secondary[r] = 1; return 1; }
    if (r == 0) {
        double s = 0.0;
        for (int q = 0; q < h->n; ++q) s += *h->ptr[q];
        for (int q = 0; q < h->n; ++q) *h->ptr[q] = s;
    }
    if (!h->bar.wait()) { h->secondary[r] = 1; return 1; }
    return 0;
}

static int rank_fn(Handle *h, int r, int cmd, const void *arg) {
    h->secondary[r] = 0;
    double v = h->val[r] * *(const double *)arg;
    int rc = exchange(h, r, 1, &v);
    if (!rc) {
        v += 1.0;
        rc = exchange(h, r, 2, &v);
    }
    if (rc) {
        h->bar.poison();
        if (h->gate_mode) h->gate.abort(r, "injected", [&] { h->comm.abort(); h->comm_alive = false; });
        return 7;
    }
    h->out[r] = v + cmd;
    return 0;
}

static int run(bool gate_mode, int n, int commands, unsigned seed, double fail_prob) {
    std::mt19937 rng(seed);
    int failures = 0, handles = 0;
    Handle *h = nullptr;
    auto make = [&] {
        h = new Handle();
        ++handles;
        h->n = n;
        h->gate_mode = gate_mode;
        h->bar.n = n;
        h->comm.n = n;
        h->ptr.assign(n, nullptr);
        h->val.resize(n);
        h->out.assign(n, 0.0);
        h->secondary.assign(n, 0);
        for (int r = 0; r < n; ++r) h->val[r] = r + 1.0;
        Handle *hh = h;
        h->core.run = [hh](int r, int cmd, const void *arg) { return rank_fn(hh, r, cmd, arg); };
        h->core.start(n);
    };
    make();
    for (int c = 0; c < commands; ++c) {
        if (gate_mode && h->gate.is_dead()) {            // what a caller does with a dead handle: destroy it, create a new one
            h->core.stop();
            delete h;
            make();
        }
        const bool inject = std::uniform_real_distribution<double>(0, 1)(rng) < fail_prob;
        h->inject_rank = inject ? (int)(rng() % n) : -1;
        h->inject_exchange = 1 + (int)(rng() % 2);
        const double scale = 1.0 + (c % 7);
        h->core.submit(1 + (c & 1), &scale);
        h->bar.reset();
        int bad = -1;
        for (int pass = 0; pass < 2 && bad < 0; ++pass)
            for (int r = 0; r < n; ++r)
                if (h->core.rc[r] && (pass == 1 || !h->secondary[r])) { bad = r; break; }
        if (inject) {
            ++failures;
            if (bad != h->inject_rank) { fprintf(stderr, "command %d: failure of rank %d reported as rank %d\n", c, h->inject_rank, bad); return 1; }
            if (gate_mode && !h->gate.is_dead()) { fprintf(stderr, "command %d: handle not dead after a failure\n", c); return 1; }
        } else {
            if (bad >= 0) { fprintf(stderr, "command %d: spurious failure on rank %d\n", c, bad); return 1; }
            const double s1 = scale * n * (n + 1) / 2.0, want = (s1 + 1.0) * n + 1 + (c & 1);
            for (int r = 0; r < n; ++r)
                if (h->out[r] != want) { fprintf(stderr, "command %d rank %d: %.17g != %.17g\n", c, r, h->out[r], want); return 1; }
        }
    }
    h->core.stop();
    delete h;
    printf("%s: %d ranks, %d commands, %d injected failures, %d handles: ok\n", gate_mode ? "gate (RCCL-like)" : "loopback", n, commands, failures, handles);
    return 0;
}

int main(int argc, char **argv) {
    const int commands = argc > 1 ? atoi(argv[1]) : 1000;
    if (run(false, 8, commands, 1, 0.05)) return 1;
    if (run(true, 8, commands, 2, 0.05)) return 1;
    if (run(false, 2, commands, 3, 0.3)) return 1;
    if (run(true, 3, commands, 4, 0.3)) return 1;
    return 0;
}

// Host side of libgpz_hip.so (gpz_ctx.h): one evaluation as recorded hipGraph segments - recording, the cuts at the exchange points of a
// sharded context and around the dominant stages (timing level 2), replay - behind gpz_eval / gpz_eval_dev.  The pipeline itself
// (stage_a, eval_tail) is gpz_eval.hip.
#include "gpz_ctx.h"

namespace gpzi {
static int eval_common(gpz_ctx *c, const double *theta, const double *theta_dev, double *f, double *g, double *g_dev,
                       double stats[4], double diag[2]);

}   // namespace gpzi
extern "C" int gpz_eval(gpz_ctx *c, const double *theta, double *f, double *g, double stats[4], double diag[2]) {
    if (!c || !theta || !f || !g) return gpz_fail(GPZ_ERR_ARG, "gpz_eval: null argument");
    return eval_common(c, theta, nullptr, f, g, nullptr, stats, diag);
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_eval_dev(gpz_ctx *c, const double *theta_dev, double *f, double *g_dev, double stats[4], double diag[2]) {
    if (!c || !theta_dev || !f || !g_dev) return gpz_fail(GPZ_ERR_ARG, "gpz_eval_dev: null argument");
    c->g_dev_out = g_dev;
    const int rc = eval_common(c, nullptr, theta_dev, f, nullptr, g_dev, stats, diag);
    c->g_dev_out = nullptr;
    return rc;
}
namespace gpzi {

// Close the graph segment being recorded (gpz_ctx.h: struct GraphSeg) and, unless it is the last, open the next one.
int graph_cut(gpz_ctx *c, bool last) {
    if (!c->capturing || !c->cap) return -1;
    gpz_ctx::GraphSeg s;
    s.stage = c->cap_stage;
    s.count_call = c->cap_stage_first;
    c->cap_stage_first = false;
    if (!c->cap_failed) {
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(c->st, &g);
        if (e == hipSuccess && g) {
            size_t nn = 0;
            e = hipGraphGetNodes(g, nullptr, &nn);
            if (e == hipSuccess && nn > 0) e = hipGraphInstantiate(&s.exec, g, nullptr, nullptr, 0);
        } else if (e == hipSuccess) e = hipErrorUnknown;
        if (g) (void)hipGraphDestroy(g);
        if (e == hipSuccess && c->opt.debug_fail_cut > 0 && (int)c->cap->segs.size() + 1 == c->opt.debug_fail_cut) e = hipErrorUnknown;   // (developer build: test hook)
        if (e == hipSuccess && !last) e = hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->cap_failed = true;
            if (c->opt.graph_debug) fprintf(stderr, "gpz: evaluation graph: segment %zu: %s\n", c->cap->segs.size(), hipGetErrorString(e));
        }
    }
    c->cap->segs.push_back(s);
    return c->cap_failed ? -1 : 0;
}
static void graph_set_drop(gpz_ctx::GraphSet &gs) {
    for (auto &s : gs.segs)
        if (s.exec) (void)hipGraphExecDestroy(s.exec);
    gs.segs.clear();
}
// One evaluation from the recorded segments: graph launches with, between them, the all-reduce hook of the exchange points and the
// events of the dominant stages (timing 2).
static int graph_replay(gpz_ctx *c, gpz_ctx::GraphSet &gs) {
    const bool all = c->timing == 2 && c->time_rest;   // level 3: every segment and every exchange point between events
    const int rest = all ? c->tm.find("rest") : -1, exch = all ? c->tm.find("exchange") : -1;
    for (auto &s : gs.segs) {
        hipEvent_t e0{}, e1{};
        const int stage = s.stage >= 0 ? s.stage : ((all && s.exec) ? rest : -1);
        if (stage >= 0) {
            e0 = c->tm.get();
            e1 = c->tm.get();
            HIPCHK(hipEventRecord(e0, c->st));
        }
        if (s.exec) HIPCHK(hipGraphLaunch(s.exec, c->st));
        if (stage >= 0) {
            HIPCHK(hipEventRecord(e1, c->st));
            c->tm.pending.push_back({stage, e0, e1, s.stage >= 0 ? s.count_call : true});
        }
        if (s.hook_count) {
            hipEvent_t h0{}, h1{};
            if (all) { h0 = c->tm.get(); h1 = c->tm.get(); HIPCHK(hipEventRecord(h0, c->st)); }
            if (c->ar_fn(c->ar_user, s.hook_buf, s.hook_count, (void *)c->st) != 0) return gpz_fail(GPZ_ERR_COMM, "all-reduce hook failed");
            if (all) { HIPCHK(hipEventRecord(h1, c->st)); c->tm.pending.push_back({exch, h0, h1, true}); }
        }
    }
    HIPCHK(hipStreamSynchronize(c->st));
    HIPCHK(hipGetLastError());
    return 0;
}

static int eval_common(gpz_ctx *c, const double *theta, const double *theta_dev, double *f, double *g, double *g_dev,
                       double stats[4], double diag[2]) {
    (void)g_dev;
    gpz_opts_scope opts_scope(&c->opt);
    HIPCHK(hipSetDevice(c->device));
    c->pinv_last[0] = c->pinv_last[1] = c->pinv_last[2] = c->pinv_last[3] = 0.0;
    const bool no_graph = c->opt.no_graph;   // (latched at creation: a context is either replayed or eager for its whole life)
    gpz_ctx::GraphSet &gs = c->gset[c->timing == 2 ? 1 : 0];
    const bool graphable = theta && !c->g_dev_out && c->timing != 1 && c->pinv_mode != 1 && !no_graph && gs.state >= 0;
    bool done = false;
    if (graphable && gs.state == 2) {
        memcpy(c->theta_h, theta, (size_t)c->p * sizeof(double));
        if (int e = graph_replay(c, gs)) return e;
        done = true;
    } else if (graphable && gs.state == 1) {
        // record: nothing runs yet (the hooks of a sharded context are not called either) - the replay below is this call's evaluation
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
            c->cap = &gs;
            c->cap_stage = -1;
            c->cap_stage_first = c->cap_failed = false;
            if (!rc && (rc = stage_a(c, theta, nullptr))) why = "stage A";   // (k_unpack clears the status words)
            if (!rc && c->cap_failed) { rc = -1; why = "segment"; }          // a cut inside stage A failed: stop recording here
            if (!rc && (rc = eval_tail(c, false))) why = "stage B";
            (void)graph_cut(c, true);
            if (c->cap_failed && !rc) { rc = -1; why = "segment"; }
            c->capturing = false;
            c->cap = nullptr;
            if (rc) {
                // A cut that failed (hipStreamEndCapture / hipGraphInstantiate / the re-opened capture) leaves graph_st NOT capturing:
                // whatever the pipeline issued after it ran for real on graph_st, without the all-reduce hooks, into the buffers the
                // eager evaluation below is about to write from the user's stream.  Close a capture that is still open, then wait for
                // graph_st: the eager evaluation starts from theta and overwrites everything those launches left behind.
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(c->graph_st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
                    hipGraph_t g = nullptr;
                    (void)hipStreamEndCapture(c->graph_st, &g);
                    if (g) (void)hipGraphDestroy(g);
                }
                (void)hipStreamSynchronize(c->graph_st);
                (void)hipGetLastError();
            }
        } else {
            rc = -1; why = "begin capture";
        }
        c->st = user_st;
        if (rc && c->opt.graph_debug)
            fprintf(stderr, "gpz: evaluation graph: %s: %s | %s\n", why, hipGetErrorString(he), gpz_last_error());
        if (c->opt.graph_debug) fprintf(stderr, "gpz: evaluation graph recording %s (%zu segments)\n", rc ? "failed" : "ok", gs.segs.size());
        if (!rc) {
            gs.state = 2;
            memcpy(c->theta_h, theta, (size_t)c->p * sizeof(double));
            if (int e = graph_replay(c, gs)) return e;
            done = true;
        } else {                     // not recordable here: stay on plain launches for the life of the context
            (void)hipGetLastError();
            graph_set_drop(gs);
            gs.state = -1;
        }
    }
    if (!done) {
        if (int e = stage_a(c, theta, theta_dev)) return e;   // (k_unpack clears the status words)
        if (int e = eval_tail(c, c->pinv_mode == 1)) return e;
        if (graphable && gs.state == 0) gs.state = 1;
    }
    // k_cond_flag (info[1], returned in slot 7 of the statistics block): SIGMA is close enough to singular that
    // inv_logdet.m may truncate -> redo the solve and everything after it through the SVD route.  PHI and the
    // reduced partials of stage A are still in place; every rank sees the same SIGMA and takes the same branch.
    if (c->pinv_mode == 0 && c->out_h[1 + c->p + 7] != 0.0) {
        HIPCHK(hipMemsetAsync(c->info, 0, 2 * sizeof(int), c->st));
        if (int e = eval_tail(c, true)) return e;
    }
    const bool have_valid = c->va.n_pad > 0;
    if (c->timing || !c->tm.pending.empty()) collect_timings(c);
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

}   // namespace gpzi

// Sanitizer driver for mex/gpz_mex.cpp (tests/test_sanitizers.py): every command of the gateway, with good arguments and with each
// class of bad ones, against the host-only stand-in of the library (gpz_stub.cpp) and the stand-in MEX runtime (mex_runtime.cpp),
// built with AddressSanitizer + UBSan.  TEST INFRASTRUCTURE.  Exit code 0 = every call behaved as expected; the sanitizers
// abort the process on a finding.  (MATLAB frees the mxArrays a failing MEX call leaves behind; the stand-in runtime does not, so
// the test runs with leak detection off.)
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "mex.h"

extern "C" {
mxArray *mexrt_double(const double *src, int ndim, const size_t *dims);
mxArray *mexrt_logical(const unsigned char *src, size_t n);
mxArray *mexrt_string(const char *s);
mxArray *mexrt_struct(void);
void mexrt_set_field(mxArray *s, const char *name, mxArray *v);
int mexrt_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs);
const char *mexrt_last_error(void);
const char *mexrt_last_id(void);
void mexrt_unload(void);
}

static int g_bad = 0;
static mxArray *mat(size_t r, size_t c, double v0 = 0.25, size_t p = 0) {
    std::vector<double> v(r * c * (p ? p : 1));
    for (size_t i = 0; i < v.size(); ++i) v[i] = v0 + 0.001 * (double)i;
    size_t dims[3] = {r, c, p};
    return mexrt_double(v.data(), p ? 3 : 2, dims);
}
static mxArray *scalar(double v) { return mat(1, 1, v); }
static mxArray *model(int d, int m, int k, const char *method, bool hetero = true, const char *dtype = nullptr, double n_gpus = -1,
                      const char *reducer = nullptr) {
    mxArray *s = mexrt_struct();
    mexrt_set_field(s, "d", scalar(d));
    mexrt_set_field(s, "m", scalar(m));
    mexrt_set_field(s, "k", scalar(k));
    mexrt_set_field(s, "method", mexrt_string(method));
    mexrt_set_field(s, "heteroscedastic", scalar(hetero ? 1.0 : 0.0));
    if (dtype) mexrt_set_field(s, "dtype", mexrt_string(dtype));
    if (n_gpus >= 0) mexrt_set_field(s, "n_gpus", scalar(n_gpus));
    if (reducer) mexrt_set_field(s, "reducer", mexrt_string(reducer));
    return s;
}
static long theta_len(int d, int m, int k, const char *method, bool hetero) {
    const long gam = method[1] == 'L' ? (method[0] == 'G' ? 1 : m) : method[1] == 'D' ? (method[0] == 'G' ? d : (long)m * d)
                                                                                         : (method[0] == 'G' ? (long)d * d : (long)m * d * d);
    return (long)m * d + gam + (long)m * k + k + (hetero ? 2L * m * k : 0);
}
// call; expect = nullptr: must succeed, else the error identifier it must raise.  Outputs are read completely and destroyed.
static void call(const char *what, int nlhs, std::vector<const mxArray *> args, const char *expect) {
    mxArray *out[8] = {};
    const int rc = mexrt_call(nlhs, out, (int)args.size(), args.data());
    if (!expect && rc) { fprintf(stderr, "%s: unexpected error %s: %s\n", what, mexrt_last_id(), mexrt_last_error()); ++g_bad; }
    if (expect && !rc) { fprintf(stderr, "%s: expected %s, the call succeeded\n", what, expect); ++g_bad; }
    if (expect && rc && strcmp(expect, mexrt_last_id())) { fprintf(stderr, "%s: expected %s, got %s: %s\n", what, expect, mexrt_last_id(), mexrt_last_error()); ++g_bad; }
    for (int q = 0; q < 8; ++q)
        if (out[q]) {
            volatile double s = 0;
            const double *p = mxGetPr(out[q]);
            for (size_t e = 0; p && e < mxGetNumberOfElements(out[q]); ++e) s = s + p[e];
            mxDestroyArray(out[q]);
        }
}

int main() {
    const int n = 37, d = 3, m = 5, k = 2;
    mxArray *cmd_eval = mexrt_string("eval"), *cmd_solve = mexrt_string("solve"), *cmd_phi = mexrt_string("phi");
    mxArray *cmd_reset = mexrt_string("reset"), *cmd_gpus = mexrt_string("gpus"), *cmd_builds = mexrt_string("builds"), *cmd_comm = mexrt_string("comm");
    mxArray *X = mat(n, d), *Y = mat(n, k), *om = mat(n, 1, 1.0);
    std::vector<unsigned char> trb(n), vab(n);
    for (int i = 0; i < n; ++i) { trb[i] = i % 4 != 0; vab[i] = !trb[i]; }
    mxArray *tr = mexrt_logical(trb.data(), n), *va = mexrt_logical(vab.data(), n), *none = mat(0, 0);
    const char *methods[] = {"GL", "VL", "GD", "VD", "GC", "VC"};
    call("gpus before anything", 1, {cmd_gpus}, nullptr);
    call("phi without a context", 1, {cmd_phi}, "gpz:state");
    for (const char *me : methods) {
        const bool diag = me[1] != 'C';
        mxArray *mo = model(d, m, k, me), *th = mat((size_t)theta_len(d, m, k, me, true), 1);
        mxArray *Psi = diag ? mat(n, d, 0.1) : mat(d, d, 0.1, n), *PsiBad = diag ? mat(n, d + 1, 0.1) : mat(d, d + 1, 0.1, n);
        call("eval", 3, {cmd_eval, th, mo, X, Y, Psi, om, tr, va}, nullptr);
        call("eval nlhs=1", 1, {cmd_eval, th, mo, X, Y, Psi, om, tr, va}, nullptr);
        call("eval no optional arguments", 2, {cmd_eval, th, mo, X, Y, none, none, none, none}, nullptr);
        call("solve", 3, {cmd_solve, th, mo, X, Y, Psi, om, tr, va}, nullptr);
        call("phi", 1, {cmd_phi}, nullptr);
        call("builds", 1, {cmd_builds}, nullptr);
        mxArray *th_short = mat((size_t)theta_len(d, m, k, me, true) - 1, 1);
        call("eval short theta", 1, {cmd_eval, th_short, mo, X, Y, Psi, om, tr, va}, "gpz:theta");
        call("solve short theta", 1, {cmd_solve, th_short, mo, X, Y, Psi, om, tr, va}, "gpz:theta");
        mxArray *Xbad = mat(n, d + 1), *Ybad = mat(n - 1, k), *omshort = mat(n - 2, 1);
        call("eval X with another width", 1, {cmd_eval, th, mo, Xbad, Y, none, none, none, none}, "gpz:size");
        call("eval Y with fewer rows", 1, {cmd_eval, th, mo, X, Ybad, none, none, none, none}, "gpz:size");
        call("eval short omega", 1, {cmd_eval, th, mo, X, Y, none, omshort, none, none}, "gpz:size");
        call("eval Psi of the wrong shape", 1, {cmd_eval, th, mo, X, Y, PsiBad, none, none, none}, "gpz:size");
        call("eval too few arguments", 1, {cmd_eval, th, mo, X, Y}, "gpz:usage");
        call("eval X of another class", 1, {cmd_eval, th, mo, tr, Y, none, none, none, none}, "gpz:type");
        mxArray *PsiSmall = diag ? mat(n - 5, d, 0.1) : mat(d, d, 0.1, n - 5);            // fewer rows than X: must not reach the library
        call("eval Psi with fewer rows", 1, {cmd_eval, th, mo, X, Y, PsiSmall, none, none, none}, "gpz:size");
        mxDestroyArray(PsiSmall);
        // stand-alone entries
        mxArray *c_getphi = mexrt_string("getphi"), *c_prior = mexrt_string("prior"), *c_predict = mexrt_string("predict");
        mxArray *w = mat(m, k), *iS = mat(m, m, 0.5, k), *pri = mat(1, m, 0.2), *Xs = mat(11, d), *PsiS = diag ? mat(11, d, 0.1) : mat(d, d, 0.1, 11);
        call("getphi", 3, {c_getphi, mo, th, Xs, PsiS}, nullptr);
        call("getphi no Psi", 1, {c_getphi, mo, th, Xs, none}, nullptr);
        call("getphi short theta", 1, {c_getphi, mo, th_short, Xs, none}, "gpz:theta");
        call("getphi X width", 1, {c_getphi, mo, th, Xbad, none}, "gpz:size");
        call("getphi usage", 1, {c_getphi, mo, th}, "gpz:usage");
        call("prior", 1, {c_prior, mo, th, Xs, PsiS}, nullptr);
        call("prior short theta", 1, {c_prior, mo, th_short, Xs, none}, "gpz:theta");
        call("predict", 5, {c_predict, mo, th, w, iS, pri, Xs, PsiS}, nullptr);
        call("predict nlhs=2, no priors, no Psi", 2, {c_predict, mo, th, w, iS, none, Xs, none}, nullptr);
        mxArray *wbad = mat(m, k + 1), *iSbad = mat(m, m, 0.5, k + 1), *pribad = mat(1, m + 1);
        call("predict w size", 1, {c_predict, mo, th, wbad, iS, pri, Xs, none}, "gpz:size");
        call("predict iSigma_w size", 1, {c_predict, mo, th, w, iSbad, pri, Xs, none}, "gpz:size");
        call("predict priors size", 1, {c_predict, mo, th, w, iS, pribad, Xs, none}, "gpz:size");
        call("predict usage", 1, {c_predict, mo, th, w}, "gpz:usage");
        for (mxArray *a : {mo, th, Psi, PsiBad, th_short, Xbad, Ybad, omshort, c_getphi, c_prior, c_predict, w, iS, pri, Xs, PsiS, wbad, iSbad, pribad})
            mxDestroyArray(a);
    }
    {   // model struct problems, optional fields
        mxArray *th = mat((size_t)theta_len(d, m, k, "VD", true), 1);
        mxArray *incomplete = mexrt_struct();
        mexrt_set_field(incomplete, "m", scalar(m));
        call("eval incomplete model", 1, {cmd_eval, th, incomplete, X, Y, none, none, none, none}, "gpz:model");
        mxArray *badmethod = model(d, m, k, "XX");
        call("eval unknown method", 1, {cmd_eval, th, badmethod, X, Y, none, none, none, none}, "gpz:model");
        mxArray *f32 = model(d, m, k, "VD", true, "f32", 2.0, "loopback");
        call("eval with dtype / n_gpus / reducer", 3, {cmd_eval, th, f32, X, Y, none, none, none, none}, nullptr);
        call("gpus", 1, {cmd_gpus}, nullptr);
        call("comm", 1, {cmd_comm}, nullptr);
        mxArray *baddt = model(d, m, k, "VD", true, "f16");
        call("eval bad dtype", 1, {cmd_eval, th, baddt, X, Y, none, none, none, none}, "gpz:model");
        mxArray *badred = model(d, m, k, "VD", true, nullptr, 1.0, "tree");
        call("eval bad reducer", 1, {cmd_eval, th, badred, X, Y, none, none, none, none}, "gpz:model");
        for (mxArray *a : {th, incomplete, badmethod, f32, baddt, badred}) mxDestroyArray(a);
    }
    {   // per-output weights (omega n x k), the complete hash of a big X on several threads (> 4 MB: chunked), verify_every / verify_info
        const int nb = 150001, db = 5;                                    // 6 MB of X: two chunks
        mxArray *Xb = mat(nb, db), *Yb = mat(nb, k), *omk = mat(nb, k, 1.0), *om3 = mat(nb, k + 1, 1.0), *omrow = mat(nb - 1, k, 1.0);
        mxArray *th = mat((size_t)theta_len(db, m, k, "VD", true), 1);
        mxArray *mo = model(db, m, k, "VD");
        mexrt_set_field(mo, "verify_every", scalar(1.0));
        mxArray *mo0 = model(db, m, k, "VD");
        mexrt_set_field(mo0, "verify_every", scalar(0.0));
        mxArray *c_vi = mexrt_string("verify_info");
        call("eval omega n x k", 3, {cmd_eval, th, mo, Xb, Yb, none, omk, none, none}, nullptr);
        call("eval again: complete hash re-checked", 1, {cmd_eval, th, mo, Xb, Yb, none, omk, none, none}, nullptr);
        mxGetPr(Xb)[12345] += 1.0;                                        // in-place edit between the samples: stale context rebuilt
        call("eval after an in-place edit of X", 1, {cmd_eval, th, mo, Xb, Yb, none, omk, none, none}, nullptr);
        call("verify_info", 1, {c_vi}, nullptr);
        call("eval omega n x (k+1)", 1, {cmd_eval, th, mo, Xb, Yb, none, om3, none, none}, "gpz:size");
        call("eval omega (n-1) x k", 1, {cmd_eval, th, mo, Xb, Yb, none, omrow, none, none}, "gpz:size");
        call("eval verify_every = 0", 1, {cmd_eval, th, mo0, Xb, Yb, none, omk, none, none}, "gpz:model");
        for (mxArray *a : {Xb, Yb, omk, om3, omrow, th, mo, mo0, c_vi}) mxDestroyArray(a);
    }
    {   // small entries
        mxArray *c_il = mexrt_string("inv_logdet"), *c_dxy = mexrt_string("dxy"), *c_pm = mexrt_string("pinv_mode"), *c_x = mexrt_string("nonsense");
        mxArray *A = mat(4, 4), *Ar = mat(4, 3), *B = mat(6, 3), *Cm = mat(5, 2), *mode1 = scalar(1.0), *mode5 = scalar(5.0);
        call("inv_logdet", 2, {c_il, A}, nullptr);
        call("inv_logdet non-square", 1, {c_il, Ar}, "gpz:size");
        call("inv_logdet usage", 1, {c_il}, "gpz:usage");
        call("dxy", 1, {c_dxy, B, Ar}, nullptr);
        call("dxy widths differ", 1, {c_dxy, B, Cm}, "gpz:size");
        call("dxy usage", 1, {c_dxy, B}, "gpz:usage");
        call("pinv_mode", 0, {c_pm, mode1}, nullptr);
        call("pinv_mode out of range", 0, {c_pm, mode5}, "gpz:usage");
        call("unknown command", 1, {c_x}, "gpz:usage");
        const mxArray *not_a_string = A;
        call("command is not a string", 1, {not_a_string}, "gpz:usage");
        call("no arguments", 1, {}, "gpz:usage");
        for (mxArray *a : {c_il, c_dxy, c_pm, c_x, A, Ar, B, Cm, mode1, mode5}) mxDestroyArray(a);
    }
    call("reset", 0, {cmd_reset}, nullptr);
    call("gpus after reset", 1, {cmd_gpus}, nullptr);
    mexrt_unload();
    for (mxArray *a : {cmd_eval, cmd_solve, cmd_phi, cmd_reset, cmd_gpus, cmd_builds, X, Y, om, tr, va, none}) mxDestroyArray(a);
    if (g_bad) { fprintf(stderr, "%d unexpected outcomes\n", g_bad); return 1; }
    printf("gateway: every command, good and bad arguments: ok\n");
    return 0;
}

// Host-only stand-in for libgpz_hip.so behind the MEX gateway, for the sanitizer build of mex/gpz_mex.cpp
// (tests/test_sanitizers.py: AddressSanitizer + UBSan has no GPU side).  TEST INFRASTRUCTURE: it computes nothing of GPz - every
// entry point checks its arguments the way the library documents them in include/gpz_hip.h and WRITES EVERY ELEMENT of every output
// it is handed at the documented size, so a gateway that allocates an output too small, passes a wrong length or reads past an
// input is caught by AddressSanitizer.  Inputs are read completely as well (checksummed into the outputs).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "gpz_hip.h"

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
static double sum(const double *p, size_t n) {
    double s = 0;
    for (size_t i = 0; p && i < n; ++i) s += p[i];
    return s;
}
static void fill(double *p, size_t n, double v) {
    for (size_t i = 0; p && i < n; ++i) p[i] = v + (double)i;
}
static bool diag_kind(const gpz_desc *d) { return d->method[1] == 'L' || d->method[1] == 'D'; }
static size_t psi_len(const gpz_desc *d, int64_t n, int32_t kind) {
    return kind == 2 ? (size_t)d->d * d->d * n : kind ? (size_t)n * d->d : 0;
}

struct gpz_ctx { int64_t n_train; int m; int pinv; };
struct gpz_mgpu {
    gpz_desc desc;
    int n_gpus;
    int64_t p, n_train;
    std::vector<gpz_ctx> ctx;
    double data_sum;
};

extern "C" {
const char *gpz_last_error(void) { return g_err.c_str(); }
int gpz_version(void) { return GPZ_VERSION; }
int gpz_device_count(void) { return 2; }
void gpz_release_cached_memory(void) {}
void gpz_debug_fail_alloc(int64_t) {}
int64_t gpz_theta_len_of(const gpz_desc *d) {
    if (!d || d->d < 1 || d->m < 1 || d->k < 1) return -1;
    const char a = d->method[0], b = d->method[1];
    if ((a != 'G' && a != 'V') || (b != 'L' && b != 'D' && b != 'C')) return -1;
    const int64_t gam = b == 'L' ? (a == 'G' ? 1 : d->m) : b == 'D' ? (a == 'G' ? d->d : (int64_t)d->m * d->d)
                                                                       : (a == 'G' ? (int64_t)d->d * d->d : (int64_t)d->m * d->d * d->d);
    return (int64_t)d->m * d->d + gam + (int64_t)d->m * d->k + d->k + (d->heteroscedastic ? 2LL * d->m * d->k : 0);   // as gpz_ctx.hip
}
int gpz_mgpu_create(const gpz_desc *desc, int32_t n_gpus, const int32_t *, int32_t reducer, int64_t n_tot, const double *X,
                    const double *Y, const double *Psi, int32_t psi_kind, const double *omega, const uint8_t *training,
                    const uint8_t *validation, gpz_mgpu **out) {
    if (!desc || !X || !Y || !out || n_tot < 1) return fail(GPZ_ERR_ARG, "gpz_mgpu_create: null argument");
    if (reducer != GPZ_REDUCER_RCCL && reducer != GPZ_REDUCER_LOOPBACK) return fail(GPZ_ERR_ARG, "unknown reducer %d", reducer);
    if ((Psi != nullptr) != (psi_kind != 0)) return fail(GPZ_ERR_ARG, "Psi and psi_kind disagree");
    if (gpz_theta_len_of(desc) < 0) return fail(GPZ_ERR_ARG, "unknown method");
    if (psi_kind && (psi_kind == 1) != diag_kind(desc) && psi_kind != 3) return fail(GPZ_ERR_ARG, "psi_kind %d does not fit the method", psi_kind);
    gpz_mgpu *h = new gpz_mgpu();
    h->desc = *desc;
    h->n_gpus = n_gpus > 0 ? n_gpus : 2;
    h->p = gpz_theta_len_of(desc);
    int64_t nt = 0;
    for (int64_t i = 0; i < n_tot; ++i) nt += (!training || training[i]) ? 1 : 0;
    int64_t nv = 0;
    for (int64_t i = 0; validation && i < n_tot; ++i) nv += validation[i];
    h->n_train = nt;
    h->data_sum = sum(X, (size_t)n_tot * desc->d) + sum(Y, (size_t)n_tot * desc->k) + sum(Psi, psi_len(desc, n_tot, psi_kind)) +
                  sum(omega, omega ? (size_t)n_tot * (size_t)(desc->omega_cols > 1 ? desc->omega_cols : 1) : 0) + (double)nv;   // reads every element the gateway promises (n x 1 or n x k)
    for (int r = 0; r < h->n_gpus; ++r) {
        const int64_t lo = (int64_t)r * nt / h->n_gpus, hi = (int64_t)(r + 1) * nt / h->n_gpus;
        h->ctx.push_back({hi - lo, desc->m, 0});
    }
    *out = h;
    return GPZ_OK;
}
void gpz_mgpu_destroy(gpz_mgpu *h) { delete h; }
int32_t gpz_mgpu_size(const gpz_mgpu *h) { return h ? h->n_gpus : 0; }
int gpz_mgpu_comm_info(const gpz_mgpu *h, int32_t rank, int32_t info[4], char *bus, int32_t cap) {
    if (!h || !info || rank < 0 || rank >= h->n_gpus) return fail(GPZ_ERR_ARG, "gpz_mgpu_comm_info: bad argument");
    info[0] = info[1] = info[2] = -1;
    info[3] = 0;
    if (bus && cap > 0) bus[0] = 0;
    return 0;
}
int32_t gpz_mgpu_alive(const gpz_mgpu *h) { return h ? 1 : 0; }
int64_t gpz_mgpu_theta_len(const gpz_mgpu *h) { return h ? h->p : -1; }
gpz_ctx *gpz_mgpu_ctx(gpz_mgpu *h, int32_t r) { return (h && r >= 0 && r < h->n_gpus) ? &h->ctx[r] : nullptr; }
int64_t gpz_n_train(const gpz_ctx *c) { return c ? c->n_train : -1; }
int gpz_ctx_set_pinv_mode(gpz_ctx *c, int mode) {
    if (!c || mode < -1 || mode > 1) return fail(GPZ_ERR_ARG, "pinv mode must be -1, 0 or 1");
    c->pinv = mode;
    return GPZ_OK;
}
int gpz_mgpu_eval(gpz_mgpu *h, const double *theta, double *f, double *g, double stats[4], double diag[2]) {
    if (!h || !theta || !f) return fail(GPZ_ERR_ARG, "gpz_mgpu_eval: null argument");
    *f = sum(theta, (size_t)h->p) + h->data_sum;
    fill(g, (size_t)h->p, 1.0);
    fill(stats, 4, 2.0);
    fill(diag, 2, 3.0);
    return GPZ_OK;
}
int gpz_mgpu_solve(gpz_mgpu *h, const double *theta, double *w, double *iS, double *part) {
    if (!h || !theta || !w) return fail(GPZ_ERR_ARG, "gpz_mgpu_solve: null argument");
    const size_t m = h->desc.m, k = h->desc.k;
    fill(w, m * k, sum(theta, (size_t)h->p));
    fill(iS, m * m * k, 4.0);
    fill(part, k, 5.0);
    return GPZ_OK;
}
int gpz_get_phi(gpz_ctx *c, double *PHI) {
    if (!c || !PHI) return fail(GPZ_ERR_ARG, "gpz_get_phi: null argument");
    fill(PHI, (size_t)c->n_train * c->m, 6.0);
    return GPZ_OK;
}
int gpz_phi(const gpz_desc *d, const double *theta, const double *Xs, int64_t ns, const double *Psi, int32_t kind, double *PHI,
            double *lnB, double *N) {
    if (!d || !theta || !Xs || ns < 1 || gpz_theta_len_of(d) < 0) return fail(GPZ_ERR_ARG, "gpz_phi: bad argument");
    const double s = sum(theta, (size_t)gpz_theta_len_of(d)) + sum(Xs, (size_t)ns * d->d) + sum(Psi, psi_len(d, ns, kind));
    fill(PHI, (size_t)ns * d->m, s);
    fill(lnB, (size_t)ns * d->k, s);
    fill(N, (size_t)ns * d->m, s);
    return GPZ_OK;
}
int gpz_prior(const gpz_desc *d, const double *theta, const double *Xs, int64_t ns, const double *Psi, int32_t kind, double *prior,
              int32_t *it) {
    if (!d || !theta || !Xs || !prior || ns < 1 || gpz_theta_len_of(d) < 0) return fail(GPZ_ERR_ARG, "gpz_prior: bad argument");
    fill(prior, (size_t)d->m, sum(theta, (size_t)gpz_theta_len_of(d)) + sum(Xs, (size_t)ns * d->d) + sum(Psi, psi_len(d, ns, kind)));
    if (it) *it = 3;
    return GPZ_OK;
}
int gpz_mgpu_predict(const gpz_desc *d, int32_t, const int32_t *, const double *theta, const double *w, const double *iS,
                     const double *priors, const double *Xs, int64_t ns, const double *Psi, int32_t kind, double *mu, double *nu,
                     double *beta_i, double *gamma, double *PHI) {
    if (!d || !theta || !w || !iS || !Xs || ns < 1 || gpz_theta_len_of(d) < 0) return fail(GPZ_ERR_ARG, "gpz_mgpu_predict: bad argument");
    const size_t m = d->m, k = d->k;
    bool miss = false;
    for (int c = 0; c < d->d; ++c) miss = miss || Xs[(size_t)c * ns] != Xs[(size_t)c * ns];
    if (miss && !priors) return fail(GPZ_ERR_ARG, "rows with missing values need priors");
    const double s = sum(theta, (size_t)gpz_theta_len_of(d)) + sum(w, m * k) + sum(iS, m * m * k) + sum(priors, priors ? m : 0) +
                     sum(Psi, psi_len(d, ns, kind));
    fill(mu, (size_t)ns * k, s);
    fill(nu, (size_t)ns * k, s);
    fill(beta_i, (size_t)ns * k, s);
    fill(gamma, (size_t)ns * k, s);
    fill(PHI, (size_t)ns * m, s);
    return GPZ_OK;
}
int gpz_inv_logdet(const double *A, int32_t m, int32_t, double *Xi, double *logdet, int32_t *info) {
    if (!A || m < 1 || !Xi || !logdet) return fail(GPZ_ERR_ARG, "gpz_inv_logdet: bad argument");
    fill(Xi, (size_t)m * m, sum(A, (size_t)m * m));
    *logdet = 1.0;
    if (info) *info = 0;
    return GPZ_OK;
}
int gpz_dxy(const double *X, int64_t nx, const double *Y, int64_t ny, int32_t d, int32_t, double *D) {
    if (!X || !Y || !D || nx < 1 || ny < 1 || d < 1) return fail(GPZ_ERR_ARG, "gpz_dxy: bad argument");
    fill(D, (size_t)nx * ny, sum(X, (size_t)nx * d) + sum(Y, (size_t)ny * d));
    return GPZ_OK;
}
}

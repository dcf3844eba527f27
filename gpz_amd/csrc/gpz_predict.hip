// Host side of libgpz_hip.so, part 4 of 4 (gpz_ctx.h): the entry points that take no context - gpz_phi, gpz_predict_full / _noisy /
// _missing, gpz_prior, gpz_inv_logdet, gpz_dxy, gpz_nan_groups (predict.m, predictDiag.m, predictCov.m, getPrior.m, inv_logdet.m, Dxy.m).
#include "gpz_ctx.h"

namespace gpzi {
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

}   // namespace gpzi
extern "C" int gpz_phi(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns, const double *Psi,
                       int32_t psi_kind, double *PHI, double *lnBeta_i, double *N) {
    if (!desc || !theta || !Xs || ns < 1) return gpz_fail(GPZ_ERR_ARG, "gpz_phi: null argument");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    int rc = run_phi_only(c, theta);
    double *tmp = nullptr, *nd = nullptr;
    if (!rc && (PHI || N)) rc = c->ar.alloc(&tmp, (size_t)ns * c->m);
    if (!rc && PHI) {
        launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp, c->tr.orig);
        if (hipMemcpyAsync(PHI, tmp, (size_t)ns * c->m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = gpz_fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
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
                rc = gpz_fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
        }
    }
    if (!rc && lnBeta_i) {
        if (hipMemcpy2DAsync(lnBeta_i, (size_t)ns * sizeof(double), c->lnbeta, (size_t)c->tr.n_pad * sizeof(double),
                             (size_t)ns * sizeof(double), c->k, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = gpz_fail(GPZ_ERR_HIP, "gpz_phi: copy failed");
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = gpz_fail(GPZ_ERR_HIP, "gpz_phi: sync failed");
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
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_predict_full(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                const double *Xs, int64_t ns, double *mu, double *nu, double *beta_i, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !Xs || ns < 1 || !mu || !nu || !beta_i)
        return gpz_fail(GPZ_ERR_ARG, "gpz_predict_full: null argument");
    gpz_ctx *c = nullptr;
    if (has_nan(Xs, ns * (int64_t)desc->d))
        return gpz_fail(GPZ_ERR_UNSUPPORTED, "gpz_predict_full: the rows have missing values (NaN): group them by pattern and call gpz_predict_missing (predict.m:45-69)");
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
        rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
    for (int o = 0; o < (int)k && !rc; ++o) {
        // iSigma_w(:,:,o) is m x m (symmetric up to rounding in the reference; used as given, B[k][j] = iS(k,j))
        std::vector<double> rowmaj(m * m);
        const double *src = iSigma_w + (size_t)o * m * m;
        for (size_t a = 0; a < m; ++a)
            for (size_t b = 0; b < m; ++b) rowmaj[a * m + b] = src[a + m * b];
        if (hipMemcpy(Sd, rowmaj.data(), m * m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
            break;
        }
        launch_fill_bext(c->st, Sd, (int)m, wd + (size_t)o * m, (int)m, (int)mp, o, Bext, dgi);
        launch_tgemm(c->st, c->Phi, (int)mp, Bext, (int)mp, T, (int)np, (int)mp, nullptr, nullptr, (int)m, -1);
        launch_nu(c->st, c->Phi, T, (int)mp, (int)ns, (int)m, nud);                       // predictDiag.m:69-71
        if (hipMemcpyAsync(nu + (size_t)o * ns, nud, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
        // mu(:,o) = PHI*w(:,o) = column m+o of T                                         // predictDiag.m:65
        if (!rc && hipMemcpy2DAsync(mu + (size_t)o * ns, sizeof(double), T + m + o, mp * sizeof(double), sizeof(double),
                                    (size_t)ns, hipMemcpyDeviceToHost, c->st) != hipSuccess)
            rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
        if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = gpz_fail(GPZ_ERR_HIP, "predict: sync failed");
    }
    if (!rc) {
        // beta_i = exp(lnBeta_i)   (predictDiag.m:73): wbeta holds exp(-lnbeta) (omega = 1)
        std::vector<double> lb((size_t)ns * k);
        if (hipMemcpy2D(lb.data(), (size_t)ns * sizeof(double), c->wbeta, np * sizeof(double), (size_t)ns * sizeof(double), k,
                        hipMemcpyDeviceToHost) != hipSuccess)
            rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
        else
            for (size_t e = 0; e < (size_t)ns * k; ++e) beta_i[e] = 1.0 / lb[e];
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, (int)mp, ns, (int)m, tmp);
            if (hipMemcpy(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                rc = gpz_fail(GPZ_ERR_HIP, "predict: copy failed");
        }
    }
    (void)hipStreamSynchronize(c->st);
    free_eval_ctx(c);
    return rc;
}
namespace gpzi {

// predictNoisy (predictDiag.m:75-125, predictCov.m:70-132): inputs with noise Psi, no missing values.
}   // namespace gpzi
extern "C" int gpz_predict_noisy(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                 const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind, double *mu,
                                 double *nu, double *beta_i, double *gamma, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !Xs || !Psi || ns < 1 || !mu || !nu || !beta_i || !gamma)
        return gpz_fail(GPZ_ERR_ARG, "gpz_predict_noisy: null argument");
    if (has_nan(Xs, ns * (int64_t)desc->d))
        return gpz_fail(GPZ_ERR_UNSUPPORTED, "gpz_predict_noisy: the rows have missing values (NaN): group them by pattern and call gpz_predict_missing (predict.m:45-69)");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_noisy: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_noisy: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_noisy: kernel failed");
    free_eval_ctx(c);
    return rc;
}
namespace gpzi {

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
void pmc_model_cache_release_all() {
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
    if (Psi && psi_kind != 2 && psi_kind != 3) return gpz_fail(GPZ_ERR_ARG, "GC/VC take Psi as a d x d x n cube (fixPsi.m:22-38) or n x d variances (psi_kind 3)");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const size_t m = c->m, mp = c->mp, np = c->tr.n_pad, k = c->k;
    const int n = c->tr.n, d = c->d, de = c->de;
    int rc = 0;
    if (hipMemcpyAsync(c->theta_d, theta, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st) != hipSuccess)
        rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: kernel failed");
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
            rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
    }
    if (!rc && PHI) {
        rc = c->ar.alloc(&tmp, (size_t)ns * m);
        if (!rc) {
            launch_transpose_out(c->st, c->Phi, c->mp, ns, c->m, tmp);
            if (hipMemcpyAsync(PHI, tmp, (size_t)ns * m * sizeof(double), hipMemcpyDeviceToHost, c->st) != hipSuccess)
                rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: copy failed");
        }
    }
    if (hipStreamSynchronize(c->st) != hipSuccess && !rc) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: sync failed");
    if (!rc && hipGetLastError() != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_predict_missing: kernel failed");
    free_eval_ctx(c);
    return rc;
}

// predictMissing / predictNoisyMissing (predictDiag.m:127-297, predictCov.m:134-337) for ONE group of rows sharing a NaN pattern (the caller
// groups the rows as predict.m:45-69 does; the pattern is taken from the first row, predictDiag.m:3).
}   // namespace gpzi
extern "C" int gpz_predict_missing(const gpz_desc *desc, const double *theta, const double *w, const double *iSigma_w,
                                   const double *priors, const double *Xs, int64_t ns, const double *Psi, int32_t psi_kind,
                                   double *mu, double *nu, double *beta_i, double *gamma, double *PHI) {
    if (!desc || !theta || !w || !iSigma_w || !priors || !Xs || ns < 1 || !mu || !nu || !beta_i || !gamma)
        return gpz_fail(GPZ_ERR_ARG, "gpz_predict_missing: null argument");
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
                return gpz_fail(GPZ_ERR_ARG, "gpz_predict_missing: the rows of a group must share one NaN pattern (predict.m:45-57)");
        }
    if (nobs == d)
        return gpz_fail(GPZ_ERR_ARG, "gpz_predict_missing: no dimension is missing (use gpz_predict_full / gpz_predict_noisy)");
    if (covk)
        return predict_missing_cov(desc, flags, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
    if (d > GPZ_PM_MAXD_DIAG) {
        ObsFlags of{nullptr};
        return predict_missing_diag(desc, of, &flags, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
    }
    return predict_missing_diag(desc, obs, nullptr, theta, w, iSigma_w, priors, Xs, ns, Psi, psi_kind, mu, nu, beta_i, gamma, PHI);
}
namespace gpzi {

// prior = getPrior(X,Psi,theta,model,[])   (getPrior.m): N once, then the fixed point on the device; the convergence
// test on the m-vector (getPrior.m:18) runs on the host between iterations.
}   // namespace gpzi
extern "C" int gpz_prior(const gpz_desc *desc, const double *theta, const double *Xs, int64_t ns, const double *Psi,
                         int32_t psi_kind, double *prior, int32_t *iterations) {
    if (!desc || !theta || !Xs || ns < 1 || !prior) return gpz_fail(GPZ_ERR_ARG, "gpz_prior: null argument");
    gpz_ctx *c = nullptr;
    if (int e = make_eval_ctx(desc, Xs, ns, Psi, psi_kind, &c)) return e;
    gpz_opts_scope opts_scope(&c->opt);
    const int m = c->m;
    int rc = run_phi_only(c, theta);
    double *nd = nullptr, *pd = nullptr, *slab = nullptr, *colsum = nullptr, *hist = nullptr;
    const int nwg = ns < 1024 ? (int)ns : 1024;
    if (!rc) rc = c->ar.alloc(&nd, (size_t)c->tr.n_pad * c->mp);
    if (!rc) rc = c->ar.alloc(&pd, (size_t)m);
    if (!rc) rc = c->ar.alloc(&slab, (size_t)nwg * m);
    if (!rc) rc = c->ar.alloc(&colsum, (size_t)m);
    if (!rc) rc = c->ar.alloc(&hist, (size_t)10 * m);
    std::vector<double> pr(m, 1.0 / m), old(m);                     // getPrior.m:5
    int it = 0;
    if (!rc) {
        NormArgs a{};
        a.Phi = c->Phi; a.ld = c->mp; a.n = (int)ns; a.m = c->m; a.d = c->d; a.de = c->de; a.kind = c->kind;
        a.gen = c->gen ? 1 : 0; a.G = c->pr.G; a.Rc = c->pr.Rc; a.Mr = c->tr.Mr; a.ucnt = c->tr.ucnt;
        a.gid = c->tr.gid; a.pat = c->pat_d; a.lnS = c->lnS; a.N = nd;
        launch_phi_norm(c->st, a);
        // getPrior.m:7-20.  The iterations run on the device in batches of PRIOR_BATCH - prior <- colsum / n by a kernel, every new
        // prior kept - and the host applies the convergence test of getPrior.m:17-19 to the batch's priors IN ORDER, so the answer
        // and the iteration count are those of testing after every pass (a read-back and a synchronisation per pass were 77 of
        // the 122 us a pass cost at n = 1e5, m = 200); the passes of a batch behind the converged one are wasted.
        constexpr int PRIOR_BATCH = 10;
        hipError_t e = hipMemcpyAsync(pd, pr.data(), m * sizeof(double), hipMemcpyHostToDevice, c->st);
        std::vector<double> hh((size_t)PRIOR_BATCH * m);
        bool done = false;
        it = 1;
        while (it <= 100 && !rc && !done) {
            const int nb = 100 - it + 1 < PRIOR_BATCH ? 100 - it + 1 : PRIOR_BATCH;
            for (int b = 0; b < nb; ++b) {
                launch_prior_iter(c->st, nd, c->mp, (int)ns, m, pd, slab, nwg);
                launch_slab_sum(c->st, slab, nwg, (size_t)m, colsum);
                launch_prior_update(c->st, colsum, (double)ns, m, pd, hist + (size_t)b * m);
            }
            if (e == hipSuccess) e = hipMemcpyAsync(hh.data(), hist, (size_t)nb * m * sizeof(double), hipMemcpyDeviceToHost, c->st);
            if (e == hipSuccess) e = hipStreamSynchronize(c->st);
            if (e != hipSuccess) { rc = gpz_fail(GPZ_ERR_HIP, "gpz_prior: %s", hipGetErrorString(e)); break; }
            for (int b = 0; b < nb; ++b, ++it) {
                old = pr;
                double num = 0.0, den = 0.0;
                for (int j = 0; j < m; ++j) {
                    pr[j] = hh[(size_t)b * m + j];                         // mean(w)   getPrior.m:15
                    num += (old[j] - pr[j]) * (old[j] - pr[j]);
                    den += (old[j] + pr[j]) * (old[j] + pr[j]);
                }
                if (sqrt(num) / sqrt(den) < 1e-10) { done = true; break; } // getPrior.m:17-19
            }
        }
    }
    if (!rc) {
        memcpy(prior, pr.data(), m * sizeof(double));
        if (iterations) *iterations = it > 100 ? 100 : it;
    }
    free_eval_ctx(c);
    return rc;
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_inv_logdet(const double *Ain, int32_t m, int32_t device, double *Xi, double *logdet, int32_t *info) {
    if (!Ain || m < 1 || !Xi || !logdet) return gpz_fail(GPZ_ERR_ARG, "gpz_inv_logdet: null argument");
    gpz_ctx *c = new gpz_ctx();
    gpz_opts_scope opts_scope(&c->opt);
    c->device = device;
    c->m = m; c->k = 1; c->mq = rup(m, GPZ_CH_NB); c->mp = rup(m + 1, 16);
    auto bail = [&](int code) { c->ar.release(); delete c; return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(gpz_fail(GPZ_ERR_HIP, "hipSetDevice(%d) failed", device));
    int rc = alloc_mm(c);
    if (rc) return bail(rc);
    double *S = nullptr, *alpha0 = nullptr;
    if ((rc = c->ar.alloc(&S, (size_t)m * m))) return bail(rc);
    if ((rc = c->ar.alloc(&alpha0, (size_t)m))) return bail(rc);
    if ((rc = c->ar.alloc(&c->slab, (size_t)c->nsplit_l * c->mq * c->mq))) return bail(rc);
    if (hipMemcpy(S, Ain, (size_t)m * m * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return bail(gpz_fail(GPZ_ERR_HIP, "copy failed"));
    (void)hipMemset(alpha0, 0, (size_t)m * sizeof(double));
    (void)hipMemset(c->info, 0, 2 * sizeof(int));
    const int mq = c->mq;
    launch_build_sigma(c->st, S, m, alpha0, m, mq, c->A, mq, c->Wm, c->logdet);   // clears Wm too
    for (int k0 = 0; k0 < mq; k0 += GPZ_CH_NB) {
        launch_chol_step(c->st, c->A, c->Lm, c->Wm, mq, mq, k0, c->logdet, c->info, chol_full_inverse_fits(mq) && !c->opt.chol_rowinv_off);
    }
    if (!(chol_full_inverse_fits(mq) && !c->opt.chol_rowinv_off))
        for (int gs = GPZ_CH_NB; gs < mq; gs *= 2) launch_trtri_level(c->st, c->Lm, c->Wm, c->Tmp, mq, mq, gs);
    if (ltl_small_fits(mq) && !c->opt.syrk_small_off) launch_ltl_small(c->st, c->Wm, mq, c->Sinv);
    else {
        launch_syrk(c->st, c->Wm, mq, nullptr, mq, mq, c->nsplit_l, c->rows_per_split_l, c->nsplit_l, c->rows_per_split_l, c->slab,
                    true);
        launch_syrk_reduce(c->st, c->slab, c->nsplit_l, c->nsplit_l, mq, c->Sinv, mq);
    }
    launch_cond_flag(c->st, S, m, alpha0, c->Sinv, mq, m, c->Tmp, c->info);
    int info_h[2] = {0, 0};
    double ld = 0.0;
    hipError_t e = hipMemcpy(info_h, c->info, 2 * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(gpz_fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
    int dropped = 0;
    if (info_h[1] != 0) {
        // numerically singular or not positive definite: the truncating SVD route of inv_logdet.m:3-15
        double *out3 = c->Tmp + mq + 8;
        if (run_jacobi_pinv(c->st, S, m, nullptr, m, c->A, c->Wm, mq, c->Tmp, (unsigned long long *)(c->Tmp + mq), c->Sinv, mq,
                            c->logdet, out3) < 0)
            return bail(gpz_fail(GPZ_ERR_HIP, "gpz_inv_logdet: Jacobi SVD failed"));
        double h3[3] = {0, 0, 0};
        e = hipMemcpy(h3, out3, sizeof h3, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(gpz_fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
        dropped = m - (int)h3[1];
        info_h[0] = 0;
    }
    e = hipMemcpy2D(Xi, (size_t)m * sizeof(double), c->Sinv, (size_t)mq * sizeof(double), (size_t)m * sizeof(double), m,
                    hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&ld, c->logdet, sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(gpz_fail(GPZ_ERR_HIP, "gpz_inv_logdet: %s", hipGetErrorString(e)));
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
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_dxy(const double *X, int64_t nx, const double *Y, int64_t ny, int32_t d, int32_t device, double *D) {
    if (!X || !Y || !D || nx < 1 || ny < 1 || d < 1) return gpz_fail(GPZ_ERR_ARG, "gpz_dxy: bad argument");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_dxy: %s", hipGetErrorString(e));
    }
    ar.release();
    return rc;
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_nan_groups(const double *X, int64_t n, int32_t d, int32_t device, int32_t *group_id, int32_t *n_groups) {
    if (!X || !group_id || !n_groups || n < 1 || d < 1) return gpz_fail(GPZ_ERR_ARG, "gpz_nan_groups: bad argument");
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
        if (e != hipSuccess) rc = gpz_fail(GPZ_ERR_HIP, "gpz_nan_groups: %s", hipGetErrorString(e));
        else *n_groups = g;
    }
    ar.release();
    return rc;
}
namespace gpzi {

}   // namespace gpzi

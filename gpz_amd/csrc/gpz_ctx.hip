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
// Part 2 of 4 (gpz_ctx.h): context creation and destruction, accessors, gpz_ctx_route.
#include "gpz_ctx.h"

namespace gpzi {
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
                if (c->pats_fixed) return gpz_fail(GPZ_ERR_ARG, "row %lld has a NaN pattern that is not in the given pattern table", (long long)idx[r]);
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
    {   // column means (k_small_tail expands its moment sums about them)
        std::vector<double> mu((size_t)de, 0.0);
        for (int c_ = 0; c_ < d && !idx.empty(); ++c_) {
            double s = 0.0;
            for (size_t r = 0; r < idx.size(); ++r) s += h[(size_t)c_ * np + r];
            mu[c_] = s / (double)idx.size();
        }
        if (int e = c->ar.alloc(&rs.xmu, (size_t)de)) return e;
        HIPCHK(hipMemcpy(rs.xmu, mu.data(), (size_t)de * sizeof(double), hipMemcpyHostToDevice));
        if (need_xr && !c->gen) {   // (the evaluation's training rows)
            const bool masked = c->has_missing || any_missing;       // (setup_data has looked at every row: has_missing is final here)
            const size_t xl = masked ? 2 * (size_t)de + 2 : (size_t)de + 2;
            rs.xs_ld = (int)xl;
            std::vector<double> xs(np * xl, 0.0);
            for (size_t r = 0; r < idx.size(); ++r) {
                xs[r * xl] = 1.0;
                for (int c_ = 0; c_ < d; ++c_) {
                    const double xv = X[(size_t)c_ * n_tot + idx[r]];
                    const bool obs = xv == xv;
                    xs[r * xl + 1 + c_] = obs ? xv - mu[c_] : 0.0;
                    if (masked) xs[r * xl + 1 + de + c_] = obs ? 1.0 : 0.0;
                }
            }
            if (int e = c->ar.alloc(&rs.Xs, np * xl)) return e;
            HIPCHK(hipMemcpy(rs.Xs, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
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
            if (g < 0) return gpz_fail(GPZ_ERR_ARG, "internal: pattern table changed during the upload");
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
        // omega(selection,:) of GPz.m:48: n x 1 (one weight per row for every output) or n x k (getOmega.m:19 on a k-column Y)
        const int oc = c->desc.omega_cols > 1 ? c->desc.omega_cols : 1;
        std::vector<double> ho(np * (size_t)oc, 0.0);
        for (int o = 0; o < oc; ++o)
            for (size_t r = 0; r < idx.size(); ++r) ho[(size_t)o * np + r] = omega[(size_t)o * n_tot + idx[r]];
        if (int e = c->ar.alloc(&rs.om, np * oc)) return e;
        HIPCHK(hipMemcpy(rs.om, ho.data(), np * oc * sizeof(double), hipMemcpyHostToDevice));
        rs.om_ld = oc > 1 ? (long)np : 0;
    }
    return 0;
}

int has_nan(const double *X, int64_t count) {
    for (int64_t i = 0; i < count; ++i)
        if (X[i] != X[i]) return 1;
    return 0;
}

int setup_model(gpz_ctx *c, const gpz_desc *desc) {
    c->desc = *desc;
    c->mid = method_id_of(desc->method);
    if (c->mid < 0) return gpz_fail(GPZ_ERR_ARG, "unknown method '%.2s'", desc->method);
    if (desc->d < 1 || desc->m < 1 || desc->k < 1) return gpz_fail(GPZ_ERR_ARG, "d, m, k must be >= 1");
    if (desc->omega_cols > 1 && desc->omega_cols != desc->k)
        return gpz_fail(GPZ_ERR_ARG, "omega must be n x 1 or n x k (omega_cols = %d, k = %d)", desc->omega_cols, desc->k);
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

int alloc_mm(gpz_ctx *c) {   // m x m stage buffers
    const size_t mq2 = (size_t)c->mq * c->mq, m = c->m, k = c->k;
    if (int e = c->ar.alloc(&c->A, mq2)) return e;
    if (int e = c->ar.alloc(&c->Lm, mq2)) return e;
    if (int e = c->ar.alloc(&c->Wm, mq2)) return e;
    if (int e = c->ar.alloc(&c->Tmp, mq2)) return e;
    if (int e = c->ar.alloc(&c->Sinv, mq2)) return e;
    if (int e = c->ar.alloc(&c->Bext, (size_t)c->mp * c->mp)) return e;
    if (c->desc.dtype == GPZ_F32)
        if (int e = c->ar.alloc(&c->Bext32, (size_t)c->mp * c->mp)) return e;
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
int setup_data(gpz_ctx *c, int64_t n_tot, const double *X, const double *Y, const double *Psi, int32_t psi_kind,
                      const double *omega, const uint8_t *training, const uint8_t *validation,
                      const uint8_t *patterns, int32_t n_patterns) {
    const gpz_desc *desc = &c->desc;
    int rc = 0;
    if ((Psi != nullptr) != (psi_kind != 0)) return gpz_fail(GPZ_ERR_ARG, "Psi and psi_kind disagree");
    if (psi_kind < 0 || psi_kind > 3) return gpz_fail(GPZ_ERR_ARG, "psi_kind must be 0..3");
    c->psi_kind_in = psi_kind;
    const bool xnan = has_nan(X, n_tot * (int64_t)c->d) != 0;
    // a given pattern table means "the data set has missing values": every rank takes the general path then, also one
    // whose own rows happen to be complete (the second all-reduce carries one record block per pattern)
    const bool table = patterns && n_patterns > 0;
    if (c->kind == GPZ_KIND_COV && (Psi || xnan || table)) {
        // general path: per-pair d x d factorisations (k_gen.hip)
        if (Psi && psi_kind != 2 && psi_kind != 3)
            return gpz_fail(GPZ_ERR_ARG, "GC/VC take Psi as a d x d x n cube (fixPsi.m:22-38) or as n x d variances (psi_kind 3)");
        // the NaN-pattern table is built per rank in first-occurrence order: shards would disagree on the ids and on the
        // size of the second all-reduce, so a sharded run must be given the table of the whole data set
        if (desc->world > 1 && xnan && !table)
            return gpz_fail(GPZ_ERR_UNSUPPORTED, "row-sharded GC/VC with missing values needs the global NaN-pattern table "
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
    if (desc->dtype != GPZ_F64 && desc->dtype != GPZ_F32) return gpz_fail(GPZ_ERR_ARG, "dtype must be GPZ_F64 or GPZ_F32");
    if (Psi && c->kind == GPZ_KIND_DIAG && psi_kind != 1)
        return gpz_fail(GPZ_ERR_ARG, "diagonal kinds take Psi as n x d (fixPsi.m:42-53)");
    c->has_psi = Psi != nullptr;
    if (c->has_psi && !c->gen) c->nm = 3 * c->de;
    // one missing value anywhere (training or validation rows) switches the mask arrays on for both row sets
    c->has_missing = xnan;
    if (hipSetDevice(c->device) != hipSuccess) return gpz_fail(GPZ_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if ((rc = upload_rowset(c, c->tr, n_tot, X, Y, omega, training, true, Psi))) return rc;
    if (c->tr.n < 1 && desc->world <= 1) return gpz_fail(GPZ_ERR_ARG, "training mask selects no rows");
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
                    return gpz_fail(GPZ_ERR_HIP, "copy failed");
            }
        }
        c->psi_fast = !c->psi32 && c->has_psi && psi_fast_path_available(c->d);
        c->psi_miss = c->psi_fast && (c->has_missing || c->ngroups > 1);
        c->nrec = 3 + c->d + c->d * c->d;
        c->nm = c->ngroups * c->nrec;                 // comm2's moment segment holds the [G][m][nrec] records
        std::vector<unsigned char> hp((size_t)c->ngroups * c->d);
        for (int g = 0; g < c->ngroups; ++g) memcpy(&hp[(size_t)g * c->d], c->pats[g].data(), c->d);
        if ((rc = c->ar.alloc(&c->pat_d, hp.size()))) return rc;
        if (hipMemcpy(c->pat_d, hp.data(), hp.size(), hipMemcpyHostToDevice) != hipSuccess) return gpz_fail(GPZ_ERR_HIP, "copy failed");
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

}   // namespace gpzi
extern "C" int gpz_ctx_create(const gpz_desc *desc, int64_t n_tot, const double *X, const double *Y, const double *Psi,
                              int32_t psi_kind, const double *omega, const uint8_t *training,
                              const uint8_t *validation, gpz_ctx **out) {
    return gpz_ctx_create_sharded(desc, n_tot, X, Y, Psi, psi_kind, omega, training, validation, nullptr, 0, out);
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_ctx_create_sharded(const gpz_desc *desc, int64_t n_tot, const double *X, const double *Y,
                                      const double *Psi, int32_t psi_kind, const double *omega, const uint8_t *training,
                                      const uint8_t *validation, const uint8_t *patterns, int32_t n_patterns,
                                      gpz_ctx **out) {
    if (!desc || !X || !Y || !out || n_tot < 1) return gpz_fail(GPZ_ERR_ARG, "gpz_ctx_create: null argument");
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
            return gpz_fail(GPZ_ERR_HIP, "copy failed");
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
    // Few basis functions (m + k <= 256 columns with y's columns inside one 16-column block, no input noise, rows resident): the product with
    // [inv(SIGMA) | w], the row scalars and the moment sums are ONE kernel that keeps whole rows of T in registers (k_small.hip).
    c->small_tail = !c->gen && !c->has_psi && !c->tile_rows &&   // (dtype f32 changes nothing here: no input noise)
                    small_tail_fits(c->kind, c->de, c->m, k, c->mp, c->has_missing) && !c->opt.small_tail_off;
    if (!c->small_tail && (rc = c->ar.alloc(&c->T, npt * mp))) return bail(rc);   // (T = PHI [inv(SIGMA) | w] exists in memory only on the other routes)
    // ... and with input noise under a diagonal kind (moment sums that are not linear in row features: 1 / (1 + psi_ic gamma_jc^2)): the same
    // kernel without features, dPHI written where T would have been, the sums by k_moments_diag from that ONE matrix
    c->small_tail_dp = k == 1 && !c->gen && c->has_psi && c->kind == GPZ_KIND_DIAG && !c->tile_rows && !c->small_tail && c->de <= 100 &&   // (row entries are indexed by a byte)
                       small_tail_fits(c->kind, 0, c->m, k, c->mp, false) && !c->opt.small_tail_off;
    if (c->small_tail_dp) {
        c->st_nwg = small_tail_nwg();
        c->st_nf = 0;
        if ((rc = c->ar.alloc(&c->st_slab, (size_t)c->st_nwg * m * 2))) return bail(rc);
    }
    if (c->small_tail) {
        c->st_nwg = small_tail_nwg();
        c->st_nf = small_tail_features(c->kind, c->de, c->has_missing);
        if ((rc = c->ar.alloc(&c->st_slab, (size_t)c->st_nwg * m * (c->st_nf + 2)))) return bail(rc);
    }
    if (c->tile_rows && (rc = c->ar.alloc(&c->tile_rstats, (size_t)c->ntiles * GPZ_NS))) return bail(rc);
    // GC + Psi in fp64, 10 < d <= 32 (evaluation contexts only: prediction and getPHI contexts have no moment stage and no T)
    if (c->psi_fast && c->mid == 4 && cpsi4_available(c->d) && !c->opt.gc_minv_off) {
        // one inverse per training row, shared by the basis functions (k_cpsi4_moments<.., SHARED>)
        if ((rc = c->ar.alloc(&c->gc_minv, (size_t)(c->tr.n > 0 ? c->tr.n : 1) * cpsi4_minv_len(c->d)))) return bail(rc);
        if (!c->psi_miss && !c->opt.gc_dense_phi_off) {   // and, without missing dimensions, the dense form of the PHI build
            const size_t kp = (size_t)gcq_kpad(c->d);
            if ((rc = c->ar.alloc(&c->gcq_A, np * kp))) return bail(rc);
            if ((rc = c->ar.alloc(&c->gcq_B, kp * mp + (size_t)c->de))) return bail(rc);   // + the centre (launch_gcq_centre)
            if (hipMemset(c->gcq_A, 0, np * kp * sizeof(double)) != hipSuccess) return bail(gpz_fail(GPZ_ERR_HIP, "memset failed"));
        }
    }
    c->fused = (k == 1) || !c->gen;   // the general GC/VC path chains r1 / r2 through its records: single output only
#ifdef GPZ_DEV_SWITCHES
    // the int8-sliced T-GEMM (developer build, GPZ_TGEMM_INT8): fp64 contexts whose PHI lies in [0, 1] - no input noise on a covariance kind - with PHI resident
    if (c->opt.tgemm_int8 && c->desc.dtype != GPZ_F32 && !c->tile_rows && !(c->gen && c->has_psi) && oz_prepare_device() == 0) {
        if ((rc = c->ar.alloc(&c->oz_A, oz_a_bytes((long)np, c->mp)))) return bail(rc);
        if ((rc = c->ar.alloc(&c->oz_B, oz_b_bytes(c->mp)))) return bail(rc);
        if ((rc = c->ar.alloc(&c->oz_cs, (size_t)c->mp))) return bail(rc);
    }
#endif
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
        // few basis functions: the triangle in one workgroup's registers (k_syrk_small.hip); config 5's fp32-operand product stays on k_syrk
        c->syrk_small = syrk_small_fits(mp) && !c->psi32 && !c->opt.syrk_small_off;
        if (c->syrk_small) {
            const size_t ns = syrk_small_slab_count((int)np_k, mp);
            if (ns > need) need = ns;
        }
        size_t need_l = (size_t)c->nsplit_l * c->mq * c->mq;
        c->slab_count = need > need_l ? need : need_l;
        if ((rc = c->ar.alloc(&c->slab, c->slab_count))) return bail(rc);
    }
    {   // column-group partial sums of the PHI build: sized by the rows ONE launch sees (a row tile when streaming), and only
        // for launches small enough to split (< 1024 workgroups of >= 256 rows: below 2^20 rows)
        size_t prow = npt < 1024 * 1024 ? npt : 0;
        if (c->va.n_pad && (size_t)c->va.n_pad < 1024 * 1024 && (size_t)c->va.n_pad > prow) prow = (size_t)c->va.n_pad;
        if (prow) {
            c->phipart_groups = 16;
            c->phipart_rows = (long)prow;
            if ((rc = c->ar.alloc(&c->phipart, (size_t)c->phipart_groups * 2 * k * prow))) return bail(rc);
        }
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
        if (nc_env <= 0 && c->kind == GPZ_KIND_COV) {
            // covariance kinds (two workgroups resident per CU): whole rounds of 512 workgroups with ~2000 rows per chunk - the slab
            // costs 2 x m x nm x 8 bytes per chunk whatever the rows.  Measured at m = 1000 (moments stage, ms; workgroups 512 / 1024 /
            // 2048): 125 000 rows 0.57 / 0.60 / 0.66, 250 000 rows 1.06 / 1.10 / 1.16, 500 000 rows 2.19 / 2.11 / 2.22, 10^6 rows
            // 4.67 / 4.60 / 4.46; c3 (m = 500) 0.27 / 0.30 / -.
            long rounds = ((long)c->tr.n * ncg + 1024 * 1000 - 1) / (1024 * 1000);
            if (rounds < 1) rounds = 1;
            if (rounds > 4) rounds = 4;
            nc = (int)(512 * rounds / ncg);
            if (nc < 1) nc = 1;
        }
        // diagonal kinds, fewer than 256 basis functions: the moment kernels' workgroups are m lanes wide (64 .. 192 threads), so as many
        // more of them fill the chip
        if (nc_env <= 0 && c->kind == GPZ_KIND_DIAG && c->m < 256) nc = nc * 256 / ((c->m + 63) / 64 * 64);
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
        return bail(gpz_fail(GPZ_ERR_ALLOC, "hipHostMalloc failed"));
    *out = c;
    return GPZ_OK;
}
namespace gpzi {

}   // namespace gpzi
extern "C" void gpz_ctx_destroy(gpz_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    if (c->priv && c->priv_free) c->priv_free(c->priv);
    c->ar.release();
    if (c->out_h) (void)hipHostFree(c->out_h);
    if (c->theta_h) (void)hipHostFree(c->theta_h);
    for (auto &gs : c->gset)
        for (auto &s : gs.segs)
            if (s.exec) (void)hipGraphExecDestroy(s.exec);
    if (c->graph_st) (void)hipStreamDestroy(c->graph_st);
    for (hipEvent_t e : c->tm.pool) (void)hipEventDestroy(e);
    delete c;
}
namespace gpzi {

}   // namespace gpzi
bool gpz_ctx_allreduce_is(const gpz_ctx *c, int (*fn)(void *, void *, size_t, void *), void **user) {
    if (user) *user = c->ar_user;
    return c->ar_fn == fn;
}
int gpz_ctx_device(const gpz_ctx *c) { return c->device; }
void gpz_ctx_attach_private(gpz_ctx *c, void *priv, void (*free_fn)(void *)) {
    if (c->priv && c->priv_free) c->priv_free(c->priv);
    c->priv = priv;
    c->priv_free = free_fn;
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_ctx_set_allreduce(gpz_ctx *c, gpz_allreduce_fn fn, void *user) {
    if (!c) return gpz_fail(GPZ_ERR_ARG, "null context");
    c->ar_fn = fn;
    c->ar_user = user;
    return GPZ_OK;
}
namespace gpzi {
}   // namespace gpzi
extern "C" int64_t gpz_theta_len(const gpz_ctx *c) { return c ? c->p : -1; }
namespace gpzi {
}   // namespace gpzi
extern "C" int64_t gpz_theta_len_of(const gpz_desc *ds) {
    if (!ds || ds->d < 1 || ds->m < 1 || ds->k < 1) return -1;
    const int mid = method_id_of(ds->method);
    if (mid < 0) return -1;
    return (int64_t)ds->m * ds->d + g_dim_of(mid, ds->m, ds->d) + (int64_t)ds->m * ds->k + ds->k + (ds->heteroscedastic ? 2LL * ds->m * ds->k : 0);
}
namespace gpzi {
}   // namespace gpzi
extern "C" int64_t gpz_n_train(const gpz_ctx *c) { return c ? c->tr.n : -1; }
namespace gpzi {
}   // namespace gpzi
extern "C" int64_t gpz_n_valid(const gpz_ctx *c) { return c ? c->va.n : -1; }
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_ctx_set_pinv_mode(gpz_ctx *c, int mode) {
    if (!c || mode < -1 || mode > 1) return gpz_fail(GPZ_ERR_ARG, "gpz_ctx_set_pinv_mode: mode must be -1, 0 or 1");
    c->pinv_mode = mode;
    return GPZ_OK;
}
namespace gpzi {
}   // namespace gpzi
extern "C" int gpz_ctx_last_pinv(const gpz_ctx *c, double out[4]) {
    if (!c || !out) return gpz_fail(GPZ_ERR_ARG, "null argument");
    for (int i = 0; i < 4; ++i) out[i] = c->pinv_last[i];
    return GPZ_OK;
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_ctx_enable_timing(gpz_ctx *c, int enable) {
    if (!c) return gpz_fail(GPZ_ERR_ARG, "null context");
    c->timing = enable < 0 ? 0 : enable > 2 ? 2 : enable;   // 0 off, 1 every stage (eager launches), 2 the dominant stages (graph segments)
    c->time_rest = enable >= 3;                              // 3: as 2, plus the other segments ("rest") and the exchange points ("exchange")
    return GPZ_OK;
}
namespace gpzi {
}   // namespace gpzi
extern "C" int gpz_ctx_reset_timings(gpz_ctx *c) {
    if (!c) return gpz_fail(GPZ_ERR_ARG, "null context");
    for (auto &v : c->tm.ms) v = 0.0;
    for (auto &v : c->tm.calls) v = 0;
    return GPZ_OK;
}
namespace gpzi {
}   // namespace gpzi
extern "C" int gpz_ctx_timings(gpz_ctx *c, const char **names, double *ms, int64_t *calls, int cap) {
    if (!c) return gpz_fail(GPZ_ERR_ARG, "null context");
    const int n = (int)c->tm.names.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = c->tm.names[i];
        if (ms) ms[i] = c->tm.ms[i];
        if (calls) calls[i] = c->tm.calls[i];
    }
    return n;
}
namespace gpzi {

// Which kernels this context runs, in words, and where the evaluation graph stands: a caller that asked for dtype = f32 learns
// here whether its rows actually take the fp32 pair kernels (input noise, no missing dimension, d <= 20) or the fp64 ones, and a
// bench line can show a graph capture that failed instead of silently timing eager launches.
}   // namespace gpzi
extern "C" int gpz_ctx_route(const gpz_ctx *c, char *buf, int cap) {
    if (!c || !buf || cap <= 0) return gpz_fail(GPZ_ERR_ARG, "gpz_ctx_route: null argument");
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
    const int gstate = c->gset[c->timing == 2 ? 1 : 0].state;
    const size_t nseg = c->gset[c->timing == 2 ? 1 : 0].segs.size();
    char gsb[96];
    if (gstate == 2 && nseg > 1) snprintf(gsb, sizeof gsb, "replayed (%zu segments%s)", nseg, c->desc.world > 1 ? ", all-reduce between them" : "");
    else snprintf(gsb, sizeof gsb, "%s", gstate == 2 ? "replayed" : gstate == -1 ? "disabled (recording failed)" : c->opt.no_graph ? "disabled (GPZ_NO_GRAPH)"
                                         : c->timing == 1 ? "off (stage timing on)" : "eager (not recorded yet)");
    const char *gs = gsb;
    const bool f32mm = c->psi32 && !c->opt.f32_contractions_off;
    char rows[96];
    if (c->tile_rows) snprintf(rows, sizeof rows, "; rows: streamed, %d tiles of %d (PHI built twice per evaluation)", c->ntiles, c->tile_rows);
    else rows[0] = 0;
    return snprintf(buf, (size_t)cap, "pair/PHI kernels: %s%s; contractions: %s MFMA%s%s; evaluation graph: %s%s", phi, why,
                    f32mm ? "fp32-operand (fp64 master sums)" : "fp64",
                    c->syrk_small ? ", PHI' W PHI with the whole triangle in one workgroup (k_syrk_small)" : "",
                    c->small_tail ? ", T-GEMM + row scalars + moments in one kernel (k_small_tail: T stays in registers)" :
                    c->small_tail_dp ? ", T-GEMM + row scalars + dPHI in one kernel (k_small_tail; moment sums with input noise by k_moments_diag)" : "", gs, rows);
}
namespace gpzi {

}   // namespace gpzi

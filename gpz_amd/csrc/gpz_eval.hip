// Host side of libgpz_hip.so, part 3 of 4 (gpz_ctx.h): the evaluation pipeline - stage A (PHI build, PHI' W PHI), the m x m stage,
// the tail (T = PHI [inv(SIGMA) | w], row epilogue, moments, finish) - behind gpz_eval / gpz_eval_dev / gpz_solve / gpz_get_phi, and
// the hipGraph capture of one evaluation.
#include "gpz_ctx.h"

namespace gpzi {
// ---- pipeline stages -----------------------------------------------------------------------------
static GenRows gen_rows(const RowSet &rs) {
    GenRows r{};
    r.Xr = rs.Xr; r.gid = rs.gid; r.Psi3 = rs.Psi3; r.rows_by_group = rs.rows_by_group; r.n = rs.n; r.n_pad = rs.n_pad;
    return r;
}

// row chunks of the fp32 moment kernel: whole 64-row wave blocks, at most gen_nchunk chunks
static void psi32_chunks(const gpz_ctx *c, int *nch, int *rpc) {
    int r = (c->tr.n + c->gen_nchunk - 1) / c->gen_nchunk;
    r = rup(r > 0 ? r : 1, 64);
    *rpc = r;
    *nch = c->tr.n > 0 ? (c->tr.n + r - 1) / r : 1;
}

static int allreduce(gpz_ctx *c, double *buf, size_t count) {
    if (c->desc.world <= 1) return 0;
    if (!c->ar_fn) return gpz_fail(GPZ_ERR_COMM, "world=%d but no all-reduce hook set (gpz_ctx_set_allreduce)", c->desc.world);
    if (c->capturing) {   // an exchange point of the recording: the hook is called between this segment and the next on every replay
        if (graph_cut(c)) return gpz_fail(GPZ_ERR_HIP, "evaluation graph: cut at an exchange point failed");
        c->cap->segs.back().hook_buf = buf;
        c->cap->segs.back().hook_count = count;
        return 0;
    }
    if (c->ar_fn(c->ar_user, buf, count, (void *)c->st) != 0) return gpz_fail(GPZ_ERR_COMM, "all-reduce hook failed");
    return 0;
}

// Sharded fp32 runs: the diagonal-Psi kernels leave WHITENED moment records, the full-Psi kernels plain ones, and the
// records are summed over ranks — so every rank must run the same form.  A rank whose own rows are all diagonal
// switches to the full form (its diagonals expanded to packed triangles on the device) when any other rank needs it.
static int psi32_agree(gpz_ctx *c) {
    if (!c->psi32 || c->psi32_agreed || c->desc.world <= 1) return 0;
    const double mine = (c->tr.psi_diag && (c->va.n_pad == 0 || c->va.psi_diag)) ? 0.0 : 1.0;
    HIPCHK(hipMemcpyAsync(c->rstats, &mine, sizeof(double), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (int e = allreduce(c, c->rstats, 1)) return e;
    double total = 0.0;
    HIPCHK(hipMemcpyAsync(&total, c->rstats, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    if (total > 0.0) {
        const int D = psi32_pad_dim(c->d);
        for (RowSet *rs : {&c->tr, &c->va}) {
            if (!rs->PsiT || !rs->psi_diag) continue;
            float *full = nullptr;
            const size_t np = (size_t)rs->n_pad;
            if (int e = c->ar.alloc(&full, (size_t)D * (D + 1) / 2 * np)) return e;
            HIPCHK(hipMemsetAsync(full, 0, (size_t)D * (D + 1) / 2 * np * sizeof(float), c->st));
            for (int a = 0; a < D; ++a)   // diagonal a -> packed element (a, a)
                HIPCHK(hipMemcpyAsync(full + ((size_t)a * (a + 1) / 2 + a) * np, rs->PsiT + (size_t)a * np, np * sizeof(float),
                                      hipMemcpyDeviceToDevice, c->st));
            rs->PsiT = full;
            rs->psi_diag = 0;
        }
    }
    c->psi32_agreed = true;
    return 0;
}

// GC/VC with missing dimensions and no input noise: the rows of every NaN pattern (stored contiguously) go through the
// tuned PHI kernel with that pattern's parameter block (k_gen_pattern_params).  Launches run in row order on one
// stream: a launch zero-fills up to the end of its last 1024-row block, the next pattern's launch rewrites those rows.
static int phi_by_pattern(gpz_ctx *c, RowSet &rs, double *Phi, double *lnbeta, double *wbeta, const double *w, double *phiw,
                          bool with_y) {
    const size_t np = (size_t)rs.n_pad, mp = (size_t)c->mp;
    const int de = c->de;
    const size_t tail = np - (size_t)rs.n;   // rows past the data: zero (the slack block is never written otherwise)
    if (Phi) HIPCHK(hipMemsetAsync(Phi + (size_t)rs.n * mp, 0, tail * mp * sizeof(double), c->st));
    for (int o = 0; o < c->k; ++o) {
        HIPCHK(hipMemsetAsync(lnbeta + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
        if (wbeta) HIPCHK(hipMemsetAsync(wbeta + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
        if (phiw) HIPCHK(hipMemsetAsync(phiw + (size_t)o * np + rs.n, 0, tail * sizeof(double), c->st));
    }
    if (!rs.nwg_tab) return 0;
    // one launch over all patterns: every workgroup looks up its row range and its pattern's parameter block
    PhiArgs a{};
    a.Xc = rs.Xc; a.ldx = (long)np; a.n = rs.n; a.n_pad = rs.n_pad;
    a.m = c->m; a.mp = c->mp; a.d = de; a.k = c->k; a.kind = GPZ_KIND_COV;
    a.P = c->pr.P; a.G = c->RcP;
    a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b;
    a.omega = rs.om; a.om_ld = rs.om_ld;
    a.Y = (with_y && rs.Y) ? rs.Y : nullptr;
    a.Phi = Phi;
    a.lnbeta = lnbeta; a.wbeta = wbeta;
    a.w = w; a.phiw = phiw;
    a.part = a.n_pad <= c->phipart_rows ? c->phipart : nullptr; a.part_groups = c->phipart_groups;      // few workgroups: split the basis functions as well
    a.wgtab = rs.wgtab; a.nwg_tab = rs.nwg_tab;
    if (launch_phi(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", de);
    return 0;
}

// dP/dGamma moment records of every pattern through the tuned moment kernels (fused: single output, dPHI formed on
// the fly; plain: dPHI already in T), converted to the records k_gen_finish chains (k_gen_convert_moments).
static int moments_by_pattern(gpz_ctx *c, bool fused, double *mom) {
    const int de = c->de, nmt = de + de * (de + 1) / 2, stride = fused ? nmt + 2 : nmt;
    const size_t m = (size_t)c->m;
    if (c->mom_nchunk > 0) {   // one launch: the chunk table keeps every chunk inside one pattern's rows
        if (fused) {
            FusedMomentArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.rowscal = c->rowscal;
            a.n = c->tr.n; a.m = c->m; a.d = de; a.kind = GPZ_KIND_COV; a.P = c->pr.P; a.w = c->w;
            a.v = c->hetero ? c->pr.v : nullptr; a.nchunk = c->mom_nchunk; a.rows_per_chunk = 0; a.slab = c->gen_tslab;
            a.nm = nmt; a.chunktab = c->mom_chunktab;
            if (launch_moments_fused(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", de);
        } else {
            MomentArgs a{};
            a.dPhi = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
            a.m = c->m; a.d = de; a.kind = GPZ_KIND_COV; a.P = c->pr.P; a.nchunk = c->mom_nchunk; a.rows_per_chunk = 0;
            a.slab = c->gen_tslab; a.nm = nmt; a.chunktab = c->mom_chunktab;
            if (launch_moments(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", de);
        }
    }
    launch_slab_sum_seg(c->st, c->gen_tslab, c->mom_segtab, c->ngroups, m * stride, c->gen_frec);
    launch_gen_convert_moments(c->st, c->gen_frec, stride, fused ? 1 : 0, c->Sig, c->pat_d, c->ngroups, c->m, c->d, de, mom,
                               c->nrec);
    return 0;
}

// PHI, ln beta and omega*beta of the training row set from the unpacked parameters (getPHI.m:60-125, GPz.m:43-48).
int build_phi(gpz_ctx *c) {
    if (c->gen) {
        Stage s(c, "phi_build");
        // Sigma_j / inv(Sigma_j) / ln|Sigma_j|: everything except the whitened fp32 route (which works from the QR factor)
        const bool whitened = c->psi32 && c->tr.psi_diag && (c->va.n_pad == 0 || c->va.psi_diag);
        if (!whitened)
            launch_gen_prep(c->st, c->pr.G, c->m, c->d, c->de, c->Sig, c->iSig, c->pat_d, c->ngroups, c->lnS, c->gen_ws);
        if (!c->has_psi) {   // missing dimensions only: tuned kernels, one launch per NaN pattern
            launch_gen_pattern_params(c->st, c->Sig, c->pr.P, c->pat_d, c->ngroups, c->m, c->d, c->de, c->RcP, c->gen_ws);
            return phi_by_pattern(c, c->tr, c->Phi, c->lnbeta, c->wbeta, nullptr, nullptr, true);
        }
        if (c->psi32) {
            launch_psi32_phi(c->st, c->tr.Xr, c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m,
                             c->pr.P, c->Sig, c->pr.Rc, c->lnS, c->Phi, c->mp);
            launch_gen_fill(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->mp, c->k, c->tr.Y);
        } else if (c->psi_fast) {
            double *gcq_ctr = c->gcq_A ? c->gcq_B + (size_t)gcq_kpad(c->d) * c->mp : nullptr;   // (behind the table: d doubles)
            if (c->gcq_A) launch_gcq_centre(c->st, c->m, c->d, c->de, c->pr.P, gcq_ctr);
            if (c->gc_minv)   // GC: Sigma + Psi_i inverted once per row - for the moment kernel, and for the dense form of the PHI build
                launch_cpsi4_minv(c->st, gen_rows(c->tr), c->d, c->de, c->Sig, c->lnS, c->psi_miss ? c->pat_d : nullptr, c->gc_minv,
                                  c->gcq_A, gcq_kpad(c->d), gcq_ctr);
            if (c->gcq_A) {
                // ln PHI = -1/2 [c_ab M^-1_ab | M^-1 x | x'M^-1 x + ln|M| - ln|Sigma|] . [p_a p_b ; -2 p ; 1]: one product on the T-GEMM kernel
                // (c->T is free until the evaluation's own T-GEMM) and an exp
                launch_gcq_tab(c->st, c->m, c->d, c->de, c->mp, c->pr.P, gcq_ctr, c->gcq_B);
                launch_tgemm(c->st, c->gcq_A, gcq_kpad(c->d), c->gcq_B, c->mp, c->T, c->tr.n_pad, c->mp, nullptr, nullptr, c->m, -1, false,
                             gcq_kpad(c->d), c->mp);
                launch_gcq_exp(c->st, c->T, c->mp, c->tr.n, c->m, c->Phi);
            } else
                launch_psi_phi(c->st, gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, c->lnS, c->Phi, c->mp,
                               c->psi_miss ? c->pat_d : nullptr, c->mid == 4);
            launch_gen_fill(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->mp, c->k, c->tr.Y);
        } else {
            launch_gen_phi(c->st, gen_rows(c->tr), c->m, c->mp, c->d, c->de, c->k, c->pr.P, c->Sig, c->lnS, c->pat_d,
                           c->Phi, c->tr.Y, c->gen_ws);
        }
        launch_gen_rowdot(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          c->tr.om, nullptr, c->lnbeta, c->wbeta, nullptr, c->tr.om_ld);
    } else {
        Stage s(c, "phi_build");
        PhiArgs a{};
        a.Xc = c->tr.Xc; a.ldx = c->tr.n_pad; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om; a.om_ld = c->tr.om_ld; a.Y = c->tr.Y;
        a.Phi = c->Phi; a.lnbeta = c->lnbeta; a.wbeta = c->wbeta; a.w = nullptr; a.phiw = nullptr;
        a.Psic = c->tr.Psic; a.Mc = c->tr.Mc; a.ucnt = c->tr.ucnt;
        a.part = a.n_pad <= c->phipart_rows ? c->phipart : nullptr; a.part_groups = c->phipart_groups;
        if (launch_phi(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
    }
    return 0;
}

// Row-tile streaming: rows [r0, r0 + rows_pad) of the training set (rows of them real) as the current contents of c->Phi.
struct RowTile { long r0; int rows, rows_pad; };
static RowTile row_tile(const gpz_ctx *c, int t) {
    RowTile rt;
    rt.r0 = (long)t * c->tile_rows;
    const long left_pad = (long)c->tr.n_pad - rt.r0, left = (long)c->tr.n - rt.r0;
    rt.rows_pad = (int)(left_pad < c->tile_rows ? left_pad : c->tile_rows);
    rt.rows = (int)(left < 0 ? 0 : (left < rt.rows_pad ? left : rt.rows_pad));
    return rt;
}
static int phi_tile(gpz_ctx *c, const RowTile &rt) {
    PhiArgs a{};
    const long r0 = rt.r0;
    a.Xc = c->tr.Xc + r0; a.ldx = c->tr.n_pad; a.n = rt.rows; a.n_pad = rt.rows_pad;
    a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
    a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
    a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om ? c->tr.om + r0 : nullptr; a.om_ld = c->tr.om_ld; a.Y = c->tr.Y + r0;
    a.Phi = c->Phi; a.lnbeta = c->lnbeta + r0; a.wbeta = c->wbeta + r0; a.w = nullptr; a.phiw = nullptr;
    a.Psic = c->tr.Psic ? c->tr.Psic + r0 : nullptr; a.Mc = c->tr.Mc ? c->tr.Mc + r0 : nullptr;
    a.ucnt = c->tr.ucnt ? c->tr.ucnt + r0 : nullptr;
    a.part = a.n_pad <= c->phipart_rows ? c->phipart : nullptr; a.part_groups = c->phipart_groups;   // (row-indexed from the tile's base, stride = the tile's rows)
    if (launch_phi(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
    return 0;
}
// Stage A of a streamed evaluation: per tile PHI -> PHI' W_o PHI, summed over the tiles in comm1.
static int stage_a_tiles(gpz_ctx *c) {
    const bool f32 = c->psi32 && !c->opt.f32_contractions_off;
    for (int t = 0; t < c->ntiles; ++t) {
        const RowTile rt = row_tile(c, t);
        {
            Stage s(c, "phi_build");
            if (int e = phi_tile(c, rt)) return e;
        }
        const int nsp = (rt.rows_pad + c->rows_per_split - 1) / c->rows_per_split;
        const int nsp_d = (rt.rows_pad + c->rows_per_split_d - 1) / c->rows_per_split_d;
        for (int o = 0; o < c->k; ++o) {
            if (c->syrk_small) {
                Stage s(c, "syrk");
                launch_syrk_small(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad + rt.r0, rt.rows_pad, c->mp, c->slab,
                                  c->comm1 + (size_t)o * c->mp * c->mp, c->mp, t > 0 ? 1 : 0);
                continue;
            }
            {
                Stage s(c, "syrk");
                launch_syrk(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad + rt.r0, rt.rows_pad, c->mp, nsp, c->rows_per_split,
                            nsp_d, c->rows_per_split_d, c->slab, false, f32);
            }
            Stage s(c, "syrk_reduce");
            launch_syrk_reduce(c->st, c->slab, nsp, nsp_d, c->mp, c->comm1 + (size_t)o * c->mp * c->mp, c->mp, t > 0 ? 1 : 0);
        }
    }
    return 0;
}

// Stage A: theta -> PHI, ln beta, omega*beta, S_o = PHI' W_o PHI (incl. PHI' W_o y), sums; all-reduce #1.
int stage_a(gpz_ctx *c, const double *theta, const double *theta_dev) {
    if (theta_dev) {   // device-resident caller (gpz_eval_dev): theta never visits the host
        HIPCHK(hipMemcpyAsync(c->theta_d, theta_dev, (size_t)c->p * sizeof(double), hipMemcpyDeviceToDevice, c->st));
    } else {
        memcpy(c->theta_h, theta, (size_t)c->p * sizeof(double));
        HIPCHK(hipMemcpyAsync(c->theta_d, c->theta_h, (size_t)c->p * sizeof(double), hipMemcpyHostToDevice, c->st));
    }
    {
        Stage s(c, "unpack");
        // also clears info[0..1] and, without validation rows, the validation sums of the result block (eval_tail's layout of comm2)
        double *vsums0 = c->va.n_pad > 0 ? nullptr : c->comm2 + (size_t)c->m * c->nm + (size_t)c->k * 2 * c->mp + (size_t)c->k * 4;
        launch_unpack(c->st, c->theta_d, c->mid, c->m, c->d, c->de, c->k, c->hetero, c->pr, c->info, vsums0, gpz_ns(c->k));
        if (c->kind == GPZ_KIND_COV) launch_prep_cov(c->st, c->pr.G, c->pr.P, c->m, c->de, c->pr.Rc, c->prep_ws);
    }
    if (int e = psi32_agree(c)) return e;
    if (c->tile_rows) { if (int e = stage_a_tiles(c)) return e; }
    else if (int e = build_phi(c)) return e;
    double *sums1 = c->comm1 + (size_t)c->k * c->mp * c->mp;
    {
        Stage s(c, "row_sums");
        launch_sums1(c->st, c->tr.om, c->tr.om_ld, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), sums1);
    }
    for (int o = 0; o < c->k && !c->tile_rows; ++o) {
        if (c->syrk_small) {   // mp <= 256: product and record sum are one stage
            Stage s(c, "syrk");
            launch_syrk_small(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad, c->tr.n_pad, c->mp, c->slab,
                              c->comm1 + (size_t)o * c->mp * c->mp, c->mp, 0);
            continue;
        }
        {
            Stage s(c, "syrk");
            launch_syrk(c->st, c->Phi, c->mp, c->wbeta + (size_t)o * c->tr.n_pad, c->tr.n_pad, c->mp, c->nsplit,
                        c->rows_per_split, c->nsplit_d, c->rows_per_split_d, c->slab, false,
                        c->psi32 && !c->opt.f32_contractions_off);   // config 5: fp32-operand MFMAs, fp64 master sums
        }
        {
            Stage s(c, "syrk_reduce");
            launch_syrk_reduce(c->st, c->slab, c->nsplit, c->nsplit_d, c->mp, c->comm1 + (size_t)o * c->mp * c->mp, c->mp);
        }
    }
    {
        Stage s(c, "allreduce1");
        if (int e = allreduce(c, c->comm1, c->comm1_count)) return e;
    }
    c->phi_valid = !c->tile_rows;   // streamed: c->Phi holds the last tile only
    return 0;
}

// Stage B for one output: inv(SIGMA_o), logdet_o, w_o, dwda_o, diag; Bext = [inv | w].
static void stage_b(gpz_ctx *c, int o) {
    const int mq = c->mq, m = c->m;
    const double *S = c->comm1 + (size_t)o * c->mp * c->mp;
    // mq <= 256: every step also leaves its block row of inv(L) (k_chol.hip: chol_inverse_block); beyond, the diagonal blocks only and
    // the recursive levels behind the factorisation
    const bool rowinv = chol_full_inverse_fits(mq) && !c->opt.chol_rowinv_off;
    {
        Stage s(c, "chol");
        launch_build_sigma(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, m, mq, c->A, mq, c->Wm, c->logdet + o);   // clears Wm, logdet too
        for (int k0 = 0; k0 < mq; k0 += GPZ_CH_NB) launch_chol_step(c->st, c->A, c->Lm, c->Wm, mq, mq, k0, c->logdet + o, c->info, rowinv);
    }
    if (!rowinv) {
        Stage s(c, "trtri");
        for (int gs = GPZ_CH_NB; gs < mq; gs *= 2) launch_trtri_level(c->st, c->Lm, c->Wm, c->Tmp, mq, mq, gs);
    }
    {
        Stage s(c, "lauum");
        if (ltl_small_fits(mq) && !c->opt.syrk_small_off) launch_ltl_small(c->st, c->Wm, mq, c->Sinv);
        else {
            launch_syrk(c->st, c->Wm, mq, nullptr, mq, mq, c->nsplit_l, c->rows_per_split_l, c->nsplit_l, c->rows_per_split_l, c->slab,
                        true);
            launch_syrk_reduce(c->st, c->slab, c->nsplit_l, c->nsplit_l, mq, c->Sinv, mq);
        }
    }
    {
        Stage s(c, "solve_vectors");
        launch_post_inverse(c->st, c->Sinv, mq, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->mp, o, c->Bext,
                            c->w + (size_t)o * m, c->dwda + (size_t)o * m, c->dgi + (size_t)o * m, c->info, c->logdet);
        if (c->pinv_mode == 0) launch_cond_flag(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, c->Sinv, mq, m, c->Tmp, c->info);
    }
}

// Stage B through the rank-truncating SVD pseudo-inverse (inv_logdet.m:3-15) instead of the Cholesky inverse.
static int stage_b_pinv(gpz_ctx *c, int o) {
    const int mq = c->mq, m = c->m;
    const double *S = c->comm1 + (size_t)o * c->mp * c->mp;
    Stage s(c, "pinv_svd");
    double *sbuf = c->Tmp, *out3 = c->Tmp + mq + 8;
    unsigned long long *word = (unsigned long long *)(c->Tmp + mq);
    const int sweeps = run_jacobi_pinv(c->st, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->A, c->Wm, mq, sbuf, word, c->Sinv,
                                       mq, c->logdet + o, out3);
    if (sweeps < 0) return gpz_fail(GPZ_ERR_HIP, "pseudo-inverse (Jacobi SVD) failed: %s", hipGetErrorString(hipGetLastError()));
    double h3[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(h3, out3, sizeof h3, hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    c->pinv_last[0] = 1.0;
    c->pinv_last[1] = (o == 0) ? h3[1] : fmin(c->pinv_last[1], h3[1]);
    c->pinv_last[2] = h3[2];
    c->pinv_last[3] = (double)sweeps;
    launch_post_inverse(c->st, c->Sinv, mq, S, c->mp, c->pr.alpha + (size_t)o * m, m, c->mp, o, c->Bext,
                        c->w + (size_t)o * m, c->dwda + (size_t)o * m, c->dgi + (size_t)o * m, c->info, c->logdet);
    return 0;
}

// Everything of an evaluation after stage A (SIGMA partials reduced): solve, T-GEMM, row epilogue, moments, validation,
// all-reduce #2, finish, result copy.  pinv selects the inverse: Cholesky (false) or truncating SVD (true).
int eval_tail(gpz_ctx *c, bool pinv) {
    const size_t mp = c->mp, m = c->m, k = c->k;
    double *mom = c->comm2;
    double *cols = mom + m * c->nm;
    double *scal = cols + k * 2 * mp;
    double *vsums = scal + k * 4;
    const bool fused = c->fused;
    for (int o = 0; o < c->k; ++o) {
        if (pinv) { if (int e = stage_b_pinv(c, o)) return e; }
        else stage_b(c, o);
        if (c->tile_rows) {
            // streamed: per tile PHI again -> T -> row scalars -> moment sums into the tile's own chunks; the sums over the tiles after the walk
            const size_t oo = (size_t)o * c->tr.n_pad;
            for (int t = 0; t < c->ntiles; ++t) {
                const RowTile rt = row_tile(c, t);
                const long r0 = rt.r0;
                { Stage s(c, "phi_build"); if (int e = phi_tile(c, rt)) return e; }
                {
                    Stage s(c, "tgemm");
                    launch_tgemm(c->st, c->Phi, c->mp, c->Bext, c->mp, c->T, rt.rows_pad, c->mp, c->nupart, c->phiw + oo + r0, c->m, c->m + o,
                                 c->psi32 && !c->opt.f32_contractions_off);
                }
                {
                    Stage s(c, "row_scalars");
                    launch_row_scalars(c->st, c->nupart, c->nslots, c->phiw + oo + r0, c->tr.Y + oo + r0, c->tr.om ? c->tr.om + r0 : nullptr,
                                       c->lnbeta + oo + r0, c->wbeta + oo + r0, rt.rows_pad, rt.rows, c->rowscal + 4 * r0, c->partial, (long)o * c->tr.om_ld);
                    launch_slab_sum(c->st, c->partial, row_scalars_nwg(rt.rows), GPZ_NS, c->tile_rstats + (size_t)t * GPZ_NS);
                }
                Stage s(c, "moments");
                FusedMomentArgs a{};
                a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr + r0 * c->de; a.rowscal = c->rowscal + 4 * r0; a.n = rt.rows;
                a.m = c->m; a.d = c->de; a.kind = c->kind; a.P = c->pr.P; a.w = c->w + (size_t)o * m;
                a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
                a.rows_per_chunk = c->tile_rpc; a.nchunk = (rt.rows + c->tile_rpc - 1) / c->tile_rpc;
                a.slab = c->mom_slab + (size_t)t * c->tile_nchunk * m * (c->nm + 2); a.nm = c->nm;
                a.Psir = c->tr.Psir ? c->tr.Psir + r0 * c->de : nullptr; a.Mr = c->tr.Mr ? c->tr.Mr + r0 * c->de : nullptr; a.G2 = c->pr.G2;
                if (a.nchunk > 0 && launch_moments_fused(c->st, a))
                    return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
            }
            Stage s(c, "moments");
            launch_slab_sum(c->st, c->tile_rstats, c->ntiles, GPZ_NS, c->rstats);
            HIPCHK(hipMemcpyAsync(scal + (size_t)o * 4, c->rstats, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->st));
            const RowTile last = row_tile(c, c->ntiles - 1);
            const int nch = (c->ntiles - 1) * c->tile_nchunk + (last.rows + c->tile_rpc - 1) / c->tile_rpc;   // the last tile's chunks end the slab
            launch_slab_sum(c->st, c->mom_slab, nch, m * (c->nm + 2), c->frec);
            launch_split_fused(c->st, c->frec, c->m, c->nm, c->mp, mom, cols + (size_t)o * 2 * mp, o > 0 ? 1 : 0);
            continue;
        }
        if (c->small_tail_dp) {
            // diagonal kinds with input noise, few basis functions: T-GEMM, nu, row scalars and dPHI = -omega beta PHI o U in k_small_tail (no
            // features: the sums carry 1 / (1 + psi_ic gamma_jc^2)), dPHI in T's buffer, the moment sums from that one matrix
            {
                Stage s(c, "tail_small");
                SmallTailArgs a{};
                a.Phi = c->Phi; a.ld = c->mp; a.B = c->Bext; a.ldb = c->mp;
                a.n = c->tr.n; a.n_pad = c->tr.n_pad; a.m = c->m; a.mp = c->mp; a.d = c->de; a.kind = c->kind; a.mcol = c->m;
                a.Xs = c->tr.Xs; a.xs_ld = c->tr.xs_ld; a.missing = 0;
                a.y = c->tr.Y; a.omega = c->tr.om; a.omega1 = c->tr.om; a.lnbeta = c->lnbeta; a.wbeta = c->wbeta;
                a.w = c->w; a.v = c->hetero ? c->pr.v : c->w; a.vscale = c->hetero ? 1.0 : 0.0;
                a.phiw = c->phiw; a.slab = c->st_slab; a.partial = c->partial; a.nf = 0; a.stagger = c->opt.small_stagger;
                a.dphi = c->T; a.ldd = c->mp;
                launch_small_tail(c->st, a, c->st_nwg);
                launch_small_finish(c->st, c->st_slab, c->partial, c->st_nwg, c->m, c->de, c->kind, 0, 0, c->pr.P, c->tr.xmu, c->nm, c->mp, mom,
                                    cols, scal, 0, 1);
            }
            Stage s(c, "moments");
            MomentArgs ma{};
            ma.dPhi = c->T; ma.ld = c->mp; ma.Xr = c->tr.Xr; ma.n = c->tr.n; ma.n_pad = c->tr.n_pad; ma.m = c->m; ma.d = c->de;
            ma.kind = c->kind; ma.P = c->pr.P; ma.nchunk = c->nchunk; ma.rows_per_chunk = c->rows_per_chunk; ma.slab = c->mom_slab;
            ma.nm = c->nm; ma.Psir = c->tr.Psir; ma.Mr = c->tr.Mr; ma.G2 = c->pr.G2;
            if (launch_moments(c->st, ma)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
            launch_slab_sum(c->st, c->mom_slab, c->nchunk, m * c->nm, mom);
            continue;
        }
        if (c->small_tail) {
            // T = PHI [inv(SIGMA) | w], nu, the row scalars, dPHI and the moment sums in one kernel; T stays in registers (k_small.hip)
            Stage s(c, "tail_small");
            const size_t oo = (size_t)o * c->tr.n_pad;   // this output's columns of the k x n_pad row arrays
            SmallTailArgs a{};
            a.Phi = c->Phi; a.ld = c->mp; a.B = c->Bext; a.ldb = c->mp;
            a.n = c->tr.n; a.n_pad = c->tr.n_pad; a.m = c->m; a.mp = c->mp; a.d = c->de; a.kind = c->kind; a.mcol = c->m + o;
            a.Xs = c->tr.Xs; a.xs_ld = c->tr.xs_ld; a.missing = c->has_missing ? 1 : 0;
            a.y = c->tr.Y + oo; a.omega = c->tr.om ? c->tr.om + (size_t)o * c->tr.om_ld : nullptr; a.omega1 = c->tr.om;
            a.lnbeta = c->lnbeta + oo; a.wbeta = c->wbeta + oo;
            a.w = c->w + (size_t)o * m; a.v = c->hetero ? c->pr.v + (size_t)o * m : a.w; a.vscale = c->hetero ? 1.0 : 0.0;
            a.phiw = c->phiw + oo; a.slab = c->st_slab; a.partial = c->partial; a.nf = c->st_nf; a.stagger = c->opt.small_stagger;
            launch_small_tail(c->st, a, c->st_nwg);
            // dPHI is a sum over the outputs (GPz.m:113): the moments accumulate, the column and scalar sums are per output
            launch_small_finish(c->st, c->st_slab, c->partial, c->st_nwg, c->m, c->de, c->kind, c->st_nf, c->has_missing ? 1 : 0, c->pr.P,
                                c->tr.xmu, c->nm, c->mp, mom, cols + (size_t)o * 2 * mp, scal + (size_t)o * 4, o > 0 ? 1 : 0);
            continue;
        }
        {
            Stage s(c, "tgemm");
            // dtype f32 with the fp32 pair kernels active (config 5): fp32-operand MFMA contractions (k_gemm.hip), B = [inv(SIGMA) | w]
            // rounded to fp32 ONCE here instead of by every workgroup while it stages it (half the operand stream; same bits)
            const bool f32mm = c->psi32 && !c->opt.f32_contractions_off;
#ifdef GPZ_DEV_SWITCHES
            if (c->oz_A && !f32mm) {
                // developer build, GPZ_TGEMM_INT8: the same product as 28 exact int8 GEMMs of digit planes on the int8 matrix pipe
                if (o == 0) launch_oz_slice_a(c->st, c->Phi, c->mp, (long)c->tr.n_pad, c->m, c->mp, c->oz_A);
                launch_oz_slice_b(c->st, c->Bext, c->mp, c->mp, c->mp, c->oz_cs, c->oz_B);
                launch_oz_tgemm(c->st, c->oz_A, c->oz_B, c->oz_cs, c->Phi, c->mp, c->T, c->mp, (long)c->tr.n_pad, c->mp,
                                fused ? c->nupart : nullptr, fused ? c->phiw + (size_t)o * c->tr.n_pad : c->phiw, c->m, c->m + o);
            } else {
#else
            {
#endif
            if (f32mm && c->Bext32) launch_round_f32(c->st, c->Bext, c->Bext32, (size_t)c->mp * c->mp);
            launch_tgemm(c->st, c->Phi, c->mp, c->Bext, c->mp, c->T, c->tr.n_pad, c->mp, fused ? c->nupart : nullptr,
                         fused ? c->phiw + (size_t)o * c->tr.n_pad : c->phiw, c->m, c->m + o, f32mm, 0, 0, f32mm ? c->Bext32 : nullptr);
            }
        }
        if (fused) {
            {
                Stage s(c, "row_scalars");
                const size_t oo = (size_t)o * c->tr.n_pad;   // this output's columns of the k x n_pad row arrays
                launch_row_scalars(c->st, c->nupart, c->nslots, c->phiw + oo, c->tr.Y + oo, c->tr.om, c->lnbeta + oo,
                                   c->wbeta + oo, c->tr.n_pad, c->tr.n, c->rowscal, c->partial, (long)o * c->tr.om_ld);
                launch_slab_sum(c->st, c->partial, row_scalars_nwg(c->tr.n), GPZ_NS, c->rstats);
                HIPCHK(hipMemcpyAsync(scal + (size_t)o * 4, c->rstats, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->st));
            }
            Stage s(c, "moments");
            if (c->gen && !c->has_psi) {
                if (int e = moments_by_pattern(c, true, mom)) return e;
                continue;
            }
            if (c->gen && c->psi32) {
                int nch, rpc;
                psi32_chunks(c, &nch, &rpc);
                if (c->tr.psi_diag && psi32m_available(c->d))   // diagonal Psi: the 4 x 4-tile MFMA form (k_psi32m.hip)
                    launch_psi32m_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, c->tr.Xr,
                                          c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.n, c->m, c->pr.P, c->pr.Rc, nch, rpc,
                                          c->gen_slab);
                else
                    launch_psi32_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, c->tr.Xr,
                                         c->de, c->d, c->tr.PsiT, (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m, c->pr.P, c->Sig,
                                         c->pr.Rc, nch, rpc, c->gen_slab, c->nrec);
                launch_slab_sum(c->st, c->gen_slab, nch, m * psi32_raw_len(c->d), c->psi32_raw);
                launch_psi32_records(c->st, c->psi32_raw, c->d, c->tr.psi_diag, c->m, mom, c->nrec);
                continue;
            }
            if (c->gen && c->psi_fast) {
                int nch = c->gen_nchunk;
                if (nch > c->tr.n) nch = c->tr.n;
                if (nch < 1) nch = 1;   // a rank without training rows still writes its (zero) records
                const int rpc = (c->tr.n + nch - 1) / nch > 0 ? (c->tr.n + nch - 1) / nch : 1;
                nch = (c->tr.n + rpc - 1) / rpc;
                if (nch < 1) nch = 1;
                if (c->psi_miss) {   // one launch over all NaN patterns, one record set per pattern
                    launch_psi_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr,
                                       gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, c->mom_nchunk, 0, c->gen_slab,
                                       c->nrec, c->pat_d, c->mom_chunktab, c->gc_minv);
                    launch_slab_sum_seg(c->st, c->gen_slab, c->mom_segtab, c->ngroups, m * c->nrec, mom);
                    continue;
                }
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr,
                                   gen_rows(c->tr), c->m, c->d, c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec,
                                   nullptr, nullptr, c->gc_minv);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, mom);
                continue;
            }
            if (c->gen) {
                const GenRows gr = gen_rows(c->tr);
                for (int g = 0; g < c->ngroups; ++g) {
                    const int rb = c->tr.group_begin[g], nr = c->tr.group_begin[g + 1] - rb;
                    double *recs_g = mom + (size_t)g * m * c->nrec;
                    if (nr <= 0) { launch_zero(c->st, recs_g, m * c->nrec); continue; }
                    int nch = c->gen_nchunk;
                    if (nch > nr) nch = nr;
                    const int rpc = (nr + nch - 1) / nch;
                    nch = (nr + rpc - 1) / rpc;
                    launch_gen_moments(c->st, c->Phi, c->T, c->mp, c->rowscal, c->w, c->hetero ? c->pr.v : nullptr, gr, g, rb,
                                       nr, c->pat_d, c->m, c->d, c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, c->gen_ws);
                    launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, recs_g);
                }
                continue;
            }
            FusedMomentArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.rowscal = c->rowscal; a.n = c->tr.n; a.m = c->m;
            a.d = c->de; a.kind = c->kind; a.P = c->pr.P; a.w = c->w + (size_t)o * m;
            a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
            a.nchunk = c->nchunk; a.rows_per_chunk = c->rows_per_chunk; a.slab = c->mom_slab; a.nm = c->nm;
            a.Psir = c->tr.Psir; a.Mr = c->tr.Mr; a.G2 = c->pr.G2;
            if (launch_moments_fused(c->st, a))
                return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
            launch_slab_sum(c->st, c->mom_slab, c->nchunk, m * (c->nm + 2), c->frec);
            // dPHI is a sum over the outputs (GPz.m:113): the moments accumulate, the column sums are per output
            launch_split_fused(c->st, c->frec, c->m, c->nm, c->mp, mom, cols + (size_t)o * 2 * mp, o > 0 ? 1 : 0);
            continue;
        }
        {
            Stage s(c, "row_epilogue");
            RowArgs a{};
            a.Phi = c->Phi; a.T = c->T; a.ld = c->mp; a.n = c->tr.n; a.m = c->m; a.mp = c->mp; a.k = c->k; a.out = o;
            a.y = c->tr.Y; a.omega = c->tr.om; a.om_ld = c->tr.om_ld; a.lnbeta = c->lnbeta; a.wbeta = c->wbeta; a.ldx = c->tr.n_pad;
            a.w = c->w + (size_t)o * m; a.v = c->hetero ? c->pr.v + (size_t)o * m : nullptr;
            a.dL = c->dL; a.colslab = c->colslab; a.scal = c->scal_slab; a.nwg = c->nwg_rows;
            launch_row_epilogue(c->st, a);
            launch_colslab_reduce(c->st, c->colslab, c->scal_slab, c->nwg_rows, c->mp, cols + (size_t)o * 2 * mp,
                                  scal + (size_t)o * 4);
        }
    }
    if (!fused) {
        {
            Stage s(c, "mul_phi");
            launch_mul_phi(c->st, c->dL, c->Phi, c->T, (size_t)c->tr.n_pad * mp);
        }
        Stage s(c, "moments");
        if (c->gen && !c->has_psi) {
            if (int e = moments_by_pattern(c, false, mom)) return e;
        } else if (c->gen && c->psi32) {
            int nch, rpc;
            psi32_chunks(c, &nch, &rpc);
            if (c->tr.psi_diag && psi32m_available(c->d))
                launch_psi32m_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, c->tr.Xr, c->de, c->d, c->tr.PsiT,
                                      (long)c->tr.n_pad, c->tr.n, c->m, c->pr.P, c->pr.Rc, nch, rpc, c->gen_slab);
            else
                launch_psi32_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, c->tr.Xr, c->de, c->d, c->tr.PsiT,
                                     (long)c->tr.n_pad, c->tr.psi_diag, c->tr.n, c->m, c->pr.P, c->Sig, c->pr.Rc, nch, rpc,
                                     c->gen_slab, c->nrec);
            launch_slab_sum(c->st, c->gen_slab, nch, m * psi32_raw_len(c->d), c->psi32_raw);
            launch_psi32_records(c->st, c->psi32_raw, c->d, c->tr.psi_diag, c->m, mom, c->nrec);
        } else if (c->gen && c->psi_fast) {
            int nch = c->gen_nchunk;
            if (nch > c->tr.n) nch = c->tr.n;
            if (nch < 1) nch = 1;
            const int rpc = (c->tr.n + nch - 1) / nch > 0 ? (c->tr.n + nch - 1) / nch : 1;
            nch = (c->tr.n + rpc - 1) / rpc;
            if (nch < 1) nch = 1;
            if (c->psi_miss) {
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gen_rows(c->tr), c->m, c->d, c->de,
                                   c->pr.P, c->Sig, c->mom_nchunk, 0, c->gen_slab, c->nrec, c->pat_d, c->mom_chunktab, c->gc_minv);
                launch_slab_sum_seg(c->st, c->gen_slab, c->mom_segtab, c->ngroups, m * c->nrec, mom);
            } else {
                launch_psi_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gen_rows(c->tr), c->m, c->d, c->de,
                                   c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, nullptr, nullptr, c->gc_minv);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, mom);
            }
        } else if (c->gen) {
            const GenRows gr = gen_rows(c->tr);
            for (int g = 0; g < c->ngroups; ++g) {
                const int rb = c->tr.group_begin[g], nr = c->tr.group_begin[g + 1] - rb;
                double *recs_g = mom + (size_t)g * m * c->nrec;
                if (nr <= 0) { launch_zero(c->st, recs_g, m * c->nrec); continue; }
                int nch = c->gen_nchunk;
                if (nch > nr) nch = nr;
                const int rpc = (nr + nch - 1) / nch;
                nch = (nr + rpc - 1) / rpc;
                launch_gen_moments(c->st, c->Phi, c->T, c->mp, nullptr, nullptr, nullptr, gr, g, rb, nr, c->pat_d, c->m, c->d,
                                   c->de, c->pr.P, c->Sig, nch, rpc, c->gen_slab, c->nrec, c->gen_ws);
                launch_slab_sum(c->st, c->gen_slab, nch, m * c->nrec, recs_g);
            }
        } else {
        MomentArgs a{};
        a.dPhi = c->T; a.ld = c->mp; a.Xr = c->tr.Xr; a.n = c->tr.n; a.n_pad = c->tr.n_pad; a.m = c->m; a.d = c->de;
        a.kind = c->kind; a.P = c->pr.P; a.nchunk = c->nchunk; a.rows_per_chunk = c->rows_per_chunk;
        a.slab = c->mom_slab; a.nm = c->nm;
        a.Psir = c->tr.Psir; a.Mr = c->tr.Mr; a.G2 = c->pr.G2;
        if (launch_moments(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "moment kernel not instantiated for d=%d", c->de);
        launch_slab_sum(c->st, c->mom_slab, c->nchunk, m * c->nm, mom);
        }
    }
    const bool have_valid = c->va.n_pad > 0;
    if (have_valid && c->gen && !c->has_psi) {
        Stage s(c, "validation");
        if (int e = phi_by_pattern(c, c->va, nullptr, c->lnbeta_v, nullptr, c->w, c->phiw_v, false)) return e;
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->va.om_ld, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    } else if (have_valid && c->gen) {
        Stage s(c, "validation");
        if (c->psi32) {
            launch_psi32_phi(c->st, c->va.Xr, c->de, c->d, c->va.PsiT, (long)c->va.n_pad, c->va.psi_diag, c->va.n, c->m,
                             c->pr.P, c->Sig, c->pr.Rc, c->lnS, c->Phi_v, c->mp);
            launch_gen_fill(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->mp, c->k, nullptr);
        } else if (c->psi_fast) {
            launch_psi_phi(c->st, gen_rows(c->va), c->m, c->d, c->de, c->pr.P, c->Sig, c->lnS, c->Phi_v, c->mp,
                           c->psi_miss ? c->pat_d : nullptr, c->mid == 4);
            launch_gen_fill(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->mp, c->k, nullptr);
        } else {
            launch_gen_phi(c->st, gen_rows(c->va), c->m, c->mp, c->d, c->de, c->k, c->pr.P, c->Sig, c->lnS, c->pat_d,
                           c->Phi_v, nullptr, c->gen_ws);
        }
        launch_gen_rowdot(c->st, c->Phi_v, c->mp, c->va.n, c->va.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          nullptr, c->w, c->lnbeta_v, nullptr, c->phiw_v);
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->va.om_ld, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    } else if (have_valid) {
        Stage s(c, "validation");
        PhiArgs a{};
        a.Xc = c->va.Xc; a.ldx = c->va.n_pad; a.n = c->va.n; a.n_pad = c->va.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->va.om; a.om_ld = c->va.om_ld; a.Y = nullptr;
        a.Phi = nullptr; a.lnbeta = c->lnbeta_v; a.wbeta = nullptr; a.w = c->w; a.phiw = c->phiw_v;
        a.Psic = c->va.Psic; a.Mc = c->va.Mc; a.ucnt = c->va.ucnt;
        // few rows: the basis functions are split over workgroups here as in the training build (15 000 validation rows of c2 are 59
        // workgroups walking 208 columns each: 139 us; split, 30)
        a.part = a.n_pad <= c->phipart_rows ? c->phipart : nullptr; a.part_groups = c->phipart_groups;
        if (launch_phi(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
        launch_row_stats(c->st, c->phiw_v, c->va.Y, c->va.om, c->va.om_ld, c->lnbeta_v, c->va.n_pad, c->va.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), vsums);
    }   // (no validation rows: k_unpack zeroed vsums at the start of the evaluation)
    {
        Stage s(c, "allreduce2");
        if (int e = allreduce(c, c->comm2, c->comm2_count)) return e;
    }
    {
        Stage s(c, "finish");
        FinishArgs a{};
        a.method_id = c->mid; a.kind = c->kind; a.m = c->m; a.d = c->d; a.k = c->k; a.hetero = c->hetero;
        a.g_dim = c->g_dim; a.pr = c->pr; a.mom = mom; a.nm = c->nm; a.cols = cols; a.scal = scal;
        a.w = c->w; a.dwda = c->dwda; a.dgi = c->dgi; a.logdet = c->logdet;
        a.sums1 = c->comm1 + k * mp * mp; a.vsums = have_valid ? vsums : nullptr; a.info = c->info;
        a.out = c->out_d; a.dGfull = c->dGfull; a.p = (int)c->p; a.nmp = c->mp; a.de = c->de;
        a.psi = (c->has_psi && !c->gen) ? 1 : 0; a.gen = c->gen ? 1 : 0;
        if (c->gen && c->psi32 && c->tr.psi_diag)   // whitened records: the stable chain through R (k_psi32.hip)
            launch_psi32_finish(c->st, mom, c->m, c->d, c->de, c->pr.G, c->pr.Rc, c->mid, a.sums1, c->k, c->out_d + 1, c->dGfull,
                                c->k == 1 ? cols : nullptr, c->mp, c->nrec);
        else if (c->gen)
            launch_gen_finish(c->st, mom, c->ngroups, c->pat_d, c->m, c->d, c->de, c->pr.G, c->Sig, c->iSig, c->mid, a.sums1,
                              c->k, c->out_d + 1, c->dGfull, c->k == 1 ? cols : nullptr, c->mp, c->nrec, c->fin_part,
                              c->has_psi ? 0 : 1, c->gen_ws);
        launch_finish(c->st, a);
    }
    if (c->g_dev_out) {   // gpz_eval_dev: the gradient stays on the device, only f and the statistics block come up
        HIPCHK(hipMemcpyAsync(c->g_dev_out, c->out_d + 1, (size_t)c->p * sizeof(double), hipMemcpyDeviceToDevice, c->st));
        HIPCHK(hipMemcpyAsync(c->out_h, c->out_d, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(hipMemcpyAsync(c->out_h + 1 + c->p, c->out_d + 1 + c->p, 9 * sizeof(double), hipMemcpyDeviceToHost, c->st));
    } else {
        HIPCHK(hipMemcpyAsync(c->out_h, c->out_d, ((size_t)c->p + 10) * sizeof(double), hipMemcpyDeviceToHost, c->st));
    }
    if (c->capturing) return 0;   // being recorded into the evaluation graph: the caller synchronises after the replay
    HIPCHK(hipStreamSynchronize(c->st));
    HIPCHK(hipGetLastError());
    return 0;
}

}   // namespace gpzi
extern "C" int gpz_solve(gpz_ctx *c, const double *theta, double *w, double *iSigma_w, double *nlogML_partial) {
    if (!c || !theta || !w || !iSigma_w) return gpz_fail(GPZ_ERR_ARG, "gpz_solve: null argument");
    gpz_opts_scope opts_scope(&c->opt);
    HIPCHK(hipSetDevice(c->device));
    if (int e = stage_a(c, theta)) return e;   // (k_unpack clears the status words)
    const size_t m = c->m, mq = c->mq;
    c->pinv_last[0] = c->pinv_last[1] = c->pinv_last[2] = c->pinv_last[3] = 0.0;
    for (int o = 0; o < c->k; ++o) {
        bool pinv = c->pinv_mode == 1;
        if (!pinv) {
            stage_b(c, o);
            if (c->pinv_mode == 0) {   // see gpz_eval: take the truncating route when k_cond_flag asks for it
                int ih[2] = {0, 0};
                HIPCHK(hipMemcpyAsync(ih, c->info, sizeof ih, hipMemcpyDeviceToHost, c->st));
                HIPCHK(hipStreamSynchronize(c->st));
                if (ih[1] != 0) {
                    HIPCHK(hipMemsetAsync(c->info, 0, 2 * sizeof(int), c->st));
                    pinv = true;
                }
            }
        }
        if (pinv) { if (int e = stage_b_pinv(c, o)) return e; }
        // inv(SIGMA) is symmetric (to rounding on the SVD route): row-major == column-major
        HIPCHK(hipMemcpy2DAsync(iSigma_w + (size_t)o * m * m, m * sizeof(double), c->Sinv, mq * sizeof(double),
                                m * sizeof(double), m, hipMemcpyDeviceToHost, c->st));
    }
    HIPCHK(hipMemcpyAsync(w, c->w, m * c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    if (nlogML_partial && c->gen) {
        launch_gen_rowdot(c->st, c->Phi, c->mp, c->tr.n, c->tr.n_pad, c->m, c->k, c->hetero ? c->pr.v : nullptr, c->pr.b,
                          nullptr, c->w, c->lnbeta, nullptr, c->phiw);
        launch_row_stats(c->st, c->phiw, c->tr.Y, c->tr.om, c->tr.om_ld, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), c->rstats);
        if (int e = allreduce(c, c->rstats, gpz_ns(c->k))) return e;
        launch_solve_partial(c->st, c->pr, c->w, c->logdet, c->comm1 + (size_t)c->k * c->mp * c->mp, c->rstats, c->m,
                             c->k, c->spart);
        HIPCHK(hipMemcpyAsync(nlogML_partial, c->spart, c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    } else if (nlogML_partial) {
        PhiArgs a{};
        a.Xc = c->tr.Xc; a.ldx = c->tr.n_pad; a.n = c->tr.n; a.n_pad = c->tr.n_pad;
        a.m = c->m; a.mp = c->mp; a.d = c->de; a.k = c->k; a.kind = c->kind;
        a.P = c->pr.P; a.G = (c->kind == GPZ_KIND_COV) ? c->pr.Rc : c->pr.G2;
        a.v = c->hetero ? c->pr.v : nullptr; a.b = c->pr.b; a.omega = c->tr.om; a.om_ld = c->tr.om_ld; a.Y = nullptr;
        a.Phi = nullptr; a.lnbeta = c->lnbeta; a.wbeta = nullptr; a.w = c->w; a.phiw = c->phiw;
        a.Psic = c->tr.Psic; a.Mc = c->tr.Mc; a.ucnt = c->tr.ucnt;
        a.part = a.n_pad <= c->phipart_rows ? c->phipart : nullptr; a.part_groups = c->phipart_groups;
        if (launch_phi(c->st, a)) return gpz_fail(GPZ_ERR_UNSUPPORTED, "PHI kernel not instantiated for d=%d", c->de);
        launch_row_stats(c->st, c->phiw, c->tr.Y, c->tr.om, c->tr.om_ld, c->lnbeta, c->tr.n_pad, c->tr.n, c->k, c->partial);
        launch_slab_sum(c->st, c->partial, GPZ_SMALL_NWG, gpz_ns(c->k), c->rstats);
        if (int e = allreduce(c, c->rstats, gpz_ns(c->k))) return e;
        launch_solve_partial(c->st, c->pr, c->w, c->logdet, c->comm1 + (size_t)c->k * c->mp * c->mp, c->rstats, c->m,
                             c->k, c->spart);
        HIPCHK(hipMemcpyAsync(nlogML_partial, c->spart, c->k * sizeof(double), hipMemcpyDeviceToHost, c->st));
    }
    int info_h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(info_h, c->info, 2 * sizeof(int), hipMemcpyDeviceToHost, c->st));
    HIPCHK(hipStreamSynchronize(c->st));
    HIPCHK(hipGetLastError());
    if (c->timing || !c->tm.pending.empty()) collect_timings(c);
    if (info_h[0] != 0) {
        for (size_t e = 0; e < m * c->k; ++e) w[e] = NAN;
        for (size_t e = 0; e < m * m * c->k; ++e) iSigma_w[e] = NAN;
        if (nlogML_partial)
            for (int o = 0; o < c->k; ++o) nlogML_partial[o] = NAN;
    }
    return GPZ_OK;
}
namespace gpzi {

}   // namespace gpzi
extern "C" int gpz_get_phi(gpz_ctx *c, double *PHI) {
    if (!c || !PHI) return gpz_fail(GPZ_ERR_ARG, "gpz_get_phi: null argument");
    if (c->tile_rows) return gpz_fail(GPZ_ERR_ARG, "gpz_get_phi: this context streams PHI in row tiles (it is never whole on the device); use gpz_phi");
    if (!c->phi_valid) return gpz_fail(GPZ_ERR_ARG, "gpz_get_phi: no evaluation has been run");
    HIPCHK(hipSetDevice(c->device));
    double *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, (size_t)c->tr.n * c->m * sizeof(double)));
    launch_transpose_out(c->st, c->Phi, c->mp, c->tr.n, c->m, tmp, c->tr.orig);
    hipError_t e = hipMemcpyAsync(PHI, tmp, (size_t)c->tr.n * c->m * sizeof(double), hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipStreamSynchronize(c->st);
    (void)hipFree(tmp);
    if (e != hipSuccess) return gpz_fail(GPZ_ERR_HIP, "gpz_get_phi copy: %s", hipGetErrorString(e));
    return GPZ_OK;
}
namespace gpzi {

}   // namespace gpzi

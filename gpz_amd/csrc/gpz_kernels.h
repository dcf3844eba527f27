// Host-side launcher declarations shared by the translation units of libgpz_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gpz_options.h"

// ---- host-side helpers shared by gpz_ctx.hip and gpz_mgpu.hip ------------------------------------------------
struct gpz_ctx;
int gpz_fail(int code, const char *fmt, ...);                      // sets gpz_last_error() of the calling thread, returns code
void gpz_ctx_attach_private(gpz_ctx *c, void *priv, void (*free_fn)(void *));   // freed by gpz_ctx_destroy
bool gpz_ctx_allreduce_is(const gpz_ctx *c, int (*fn)(void *, void *, size_t, void *), void **user);   // is this the hook in place? -> its user pointer
int gpz_ctx_device(const gpz_ctx *c);

// ---- theta unpacking (getPHI.m:24-40,117,122; GPz.m:28,32,50,98-101) -------------------------
// Device parameter block produced by k_unpack from the raw theta vector.
struct GpzParams {
    double *P;       // m x d row-major  P[j*d + c]
    double *G;       // diag kinds: gamma[j*d + c];  cov kinds: Gamma_j row-major  G[j*d*d + a*d + b]
    double *G2;      // diag kinds: gamma^2 [j*d + c]                       (unused for cov kinds)
    double *Rc;      // cov kinds: [R_j packed upper | c_j = R_j p_j], R_j = QR triangular factor of Gamma_j
    double *lnAlpha; // m x k column-major (as in theta)
    double *alpha;   // exp(lnAlpha)
    double *b;       // k
    double *v;       // m x k (zeros when homoscedastic)
    double *lnTau;   // m x k
    double *tau;     // exp(lnTau)
};

void launch_unpack(hipStream_t st, const double *theta, int method_id, int m, int d, int de, int k, int hetero,
                   GpzParams pr, int *clear2 = nullptr, double *zero_p = nullptr, int zero_n = 0);   // clear2: two status words, zero_p: zero_n doubles, set to zero by the same launch

// ---- PHI build (getPHI.m:60-125) ---------------------------------------------------------------
struct PhiArgs {
    const double *Xc;    // d x ldx, column-major over rows: Xc[c*ldx + i]
    long ldx;
    int n;               // valid rows
    int n_pad;           // rows written (rows >= n are zero-filled)
    int m, mp, d, k;
    int kind;            // GPZ_KIND_*
    const double *P, *G; // see GpzParams (G = G2 for diag kinds, Rc for cov kinds)
    const double *v;     // m x k or nullptr
    const double *b;     // k
    const double *omega; // n, or k x om_ld (omega[o*om_ld + i]: an n x k omega, GPz.m:48), or nullptr (ones)
    long om_ld;          // 0: one column for every output
    const double *Y;     // k x ldx (Y[o*ldx + i]) or nullptr: written into PHI columns m..m+k-1
    double *Phi;         // n_pad x mp row-major, or nullptr (reduction-only mode)
    double *lnbeta;      // k x ldx
    double *wbeta;       // k x ldx  (omega .* exp(-lnbeta)), or nullptr
    const double *w;     // m x k: when non-null, phiw[o*ldx + i] = sum_j PHI_ij w_jo
    double *phiw;        // k x ldx
    // diagonal kinds only (nullptr = absent): input-noise variances, observed mask (1/0), missing count per row
    const double *Psic;  // d x ldx   (fixPsi.m layout n x d, zero where missing)
    const double *Mc;    // d x ldx
    const double *ucnt;  // ldx
    // scratch for the column-group split used at small row counts ([part_groups][2][k][n_pad]: the rows of THIS launch); nullptr disables it
    double *part;
    int part_groups;
    // cov kinds, rows sorted by NaN pattern: one launch over all patterns.  wgtab holds 4 ints per workgroup:
    // {first row, end of the pattern's rows, pattern index, 0}; G then points at [pattern][m][params] and n_pad / n
    // describe the whole row set.  nullptr = one parameter set for all rows.
    const int *wgtab;
    int nwg_tab;
};
int phi_cov_rows_per_wg(int de, int k);   // rows one workgroup of the cov-kind PHI kernel covers (granularity of wgtab)
bool phi_is_wide(int de, int k);          // d or k beyond the instantiated kernels: the runtime-d route of k_wide.hip
// runtime-d / any-k variants (k_wide.hip): same arguments and output layouts; -1 when the row tiles do not fit the LDS
int phi_wide_rows_per_wg();
int launch_phi_wide(hipStream_t st, const PhiArgs &a);
size_t prep_cov_ws_len(int m, int de);   // workspace (doubles) the QR needs when Gamma_j does not fit the LDS (de > 142), else 0
int launch_prep_cov_wide(hipStream_t st, const double *G, const double *P, int m, int de, double *Rc, double *ws);
void launch_prep_cov(hipStream_t st, const double *G, const double *P, int m, int de, double *Rc, double *ws = nullptr);
int launch_phi(hipStream_t st, const PhiArgs &a);   // returns 0, or -1 if d is not supported

// ---- MFMA contractions (k_gemm.hip) ------------------------------------------------------------
// off-diagonal 128-tiles: nsplit row ranges of rows_per_split rows; diagonal tiles: nsplit_d ranges of rows_per_split_d
void launch_syrk(hipStream_t st, const double *Phi, int ld, const double *wgt, int n_rows, int mp,
                 int nsplit, int rows_per_split, int nsplit_d, int rows_per_split_d, double *slab, bool tri,
                 bool f32_operands = false);
// k_syrk_small.hip: the same product for mp <= 256 columns - the whole upper triangle of 16 x 16 blocks in one workgroup's registers,
// PHI read once; sums its per-workgroup records into S (+ mirror) itself
bool syrk_small_fits(int mp);
size_t syrk_small_slab_count(int n_rows, int mp);
void launch_syrk_small(hipStream_t st, const double *Phi, int ld, const double *wgt, int n_rows, int mp, double *slab, double *S, int lds,
                       int accumulate);
bool ltl_small_fits(int mq);                                                  // inv(SIGMA) = inv(L)' inv(L) for mq <= 512 as one launch
void launch_ltl_small(hipStream_t st, const double *W, int mq, double *S);
void launch_syrk_reduce(hipStream_t st, const double *slab, int nsplit, int nsplit_d, int mp, double *S, int lds,
                        int accumulate = 0 /* S += instead of S = (row tiles of a streamed evaluation) */);
int gpz_gemm_wave_cols();   // wave columns per 128-wide tile (slots of nupart per column tile)

// ---- few basis functions: T-GEMM + row scalars + moment sums in one kernel, T never written (k_small.hip) -----------------------
struct SmallTailArgs {
    const double *Phi; int ld;              // n_pad x ld row-major (columns m .. m+k-1 hold y)
    const double *B; int ldb;               // mp x ldb: [inv(SIGMA) | w]
    int n, n_pad, m, mp, d, kind;           // d = padded dimension of Xr / P
    int mcol;                               // column of B that holds w (= column of T that is PHI w): m + the output's index
    const double *Xs; int xs_ld;            // n_pad x xs_ld rows [1 | x - mu | 0] (xs_ld = d + 2), mu = the column means (centre of the feature
                                            // expansion); diagonal kinds with missing values: [1 | (x - mu) mk | mk | 0] (xs_ld = 2 d + 2)
    int missing;                            // 1: the masked features of a diagonal kind with missing values
    const double *y, *omega, *lnbeta, *wbeta;   // n_pad each, of THIS output (omega may be nullptr)
    const double *omega1;                   // the first column of an n x k omega (omega(training) of GPz.m:236), = omega otherwise
    const double *w, *v;                    // m; without the heteroscedastic term v = w and vscale = 0 (no branch in the kernel)
    double vscale;
    double *phiw;                           // n_pad: PHI w
    double *dphi; int ldd;                  // when non-null: dPHI (n_pad x ldd, GPz.m:113) is written as well - the input-noise route of the diagonal
                                            // kinds takes its moment sums from it (k_moments_diag), with nf = 0 features here
    double *slab;                           // [nwg][m][nf + 2]: raw sums about xmu, PHI'c, PHI'dbeta
    double *partial;                        // [nwg][GPZ_NS]: sum c delta, sum omega delta^2, sum LL, sum dbeta
    int ncu;                                // compute units (set by the launcher)
    int stagger;                            // start delay of the second workgroup of a compute unit, in s_sleep(127) units of 8128 cycles (set by the launcher)
    int nf;                                 // features: 1 + 2d (diagonal kinds), 1 + d + d(d+1)/2 (covariance kinds)
};
int small_tail_features(int kind, int d, bool missing);
bool small_tail_fits(int kind, int d, int m, int k, int mp, bool missing);   // mp <= 256 columns, <= 32 features, y columns in one block
int small_tail_nwg();                            // persistent workgroups: two per compute unit
void launch_small_tail(hipStream_t st, const SmallTailArgs &a, int nwg);
void launch_small_finish(hipStream_t st, const double *slab, const double *partial, int nwg, int m, int d, int kind, int nf, int missing,
                         const double *P, const double *xmu, int nm, int mp, double *mom, double *cols,
                         double *scal, int accumulate, int cols_only = 0);   // cols_only: records without raw sums (nf = 0)   // the workgroups' records -> moments, column sums, scalar sums (one launch)
int gpz_cu_count();         // compute units of the current device (k_gemm.hip)
// nupart (optional): [gpz_gemm_wave_cols()*ceil(mp/128)][n_pad] per-wave-column partial sums of PHI.*T over columns < m; phiw: column mcol of T
void launch_tgemm(hipStream_t st, const double *Phi, int ld, const double *B, int ldb, double *T, int n_pad, int mp,
                  double *nupart, double *phiw, int m, int mcol, bool f32_operands = false, int kdim = 0, int ldt = 0,
                  const float *B32 = nullptr);   // B32 (with f32_operands): B rounded to fp32 by launch_round_f32, same leading dimension
void launch_round_f32(hipStream_t st, const double *src, float *dst, size_t n);
// ---- T = PHI * B on the int8 matrix pipe (k_oz.hip): 7 x 7 digit planes, 28 exact int32 products --------------------------------
size_t oz_a_bytes(long n_pad, int mp);
size_t oz_b_bytes(int mp);
int oz_prepare_device();   // once per device, before the first launch_oz_tgemm
void launch_oz_slice_a(hipStream_t st, const double *Phi, int ld, long n_pad, int m, int mp, char *Apl);      // PHI in [0, 1], columns < m
void launch_oz_slice_b(hipStream_t st, const double *B, int ldb, int krows, int mp, double *cs, char *Bpl);   // cs: mp column scales
void launch_oz_tgemm(hipStream_t st, const char *Apl, const char *Bpl, const double *cs, const double *Phi, int ld, double *T, int ldt,
                     long n_pad, int mp, double *nupart, double *phiw, int m, int mcol);                      // outputs as launch_tgemm's
void launch_trtri_level(hipStream_t st, const double *L, double *W, double *Tmp, int ld, int mq, int gs);

#ifndef GPZ_CH_NB
#define GPZ_CH_NB 32   // Cholesky panel width / diagonal block of the triangular inverse (64 measured 1.6x slower)
#endif
// ---- m x m factorisation pieces (k_chol.hip) -----------------------------------------------------
// A (mq x lda, mq % 32 == 0) <- S[0:m,0:m] + diag(alpha), identity on the padding.
void launch_build_sigma(hipStream_t st, const double *S, int lds, const double *alpha, int m, int mq, double *A, int lda, double *Wz = nullptr, double *logdet = nullptr);   // Wz (mq x mq) and *logdet are cleared when given
// panel + trailing update of one step in a single launch (GPZ_CH_NB == 32)
// one step: panel + trailing update, the factor into Lm, the diagonal block of inv(L) into W (W cleared beforehand; nullptr: not wanted)
bool chol_full_inverse_fits(int mq);
void launch_chol_step(hipStream_t st, double *A, double *Lm, double *W, int lda, int mq, int k0, double *logdet, int *info,
                      bool full_inverse = false);   // full_inverse: + block row k0 / 32 of inv(L) (all steps: no k_trtri_level launches)
void launch_zero(hipStream_t st, double *p, size_t count);
// Bext (mp x mp) <- [inv | w column at m | 0]; iS (m x m col-major == row-major, symmetric) copy; w, dwda, diag.
void launch_post_inverse(hipStream_t st, const double *Sinv, int ldsi, const double *S, int lds, const double *alpha,
                         int m, int mp, int out, double *Bext, double *w, double *dwda, double *dgi, int *info,
                         double *logdet);

// ---- rank-truncating pseudo-inverse, the other branch of inv_logdet.m:7-15 (k_pinv.hip) -----------
// info[1] |= 1 when SIGMA = S + diag(alpha) is so ill-conditioned that the reference's SVD might drop singular values
// (part: 128 doubles of scratch)
void launch_cond_flag(hipStream_t st, const double *S, int lds, const double *alpha, const double *Sinv, int ldsi, int m,
                      double *part, int *info);
// one-sided Jacobi SVD -> Xi = V diag(1/s) U' over s > m*eps(max s), *logdet = sum ln s (kept); out3 = [logdet, rank, max s]
// Gt, Vt: m x ld work matrices, sbuf: m doubles, word: 8-byte device word.  Returns the sweeps used, -1 on error.
int run_jacobi_pinv(hipStream_t st, const double *S, int lds, const double *alpha, int m, double *Gt, double *Vt, int ld,
                    double *sbuf, unsigned long long *word, double *Xi, int ldx, double *logdet, double *out3);

void launch_fill_bext(hipStream_t st, const double *Sinv, int ldsi, const double *w, int m, int mp, int out,
                      double *Bext, double *dgi);

// ---- row epilogue, moments, finish (k_rows.hip) --------------------------------------------------
struct RowArgs {
    const double *Phi; double *T; int ld;   // T is overwritten with dPHI (k==1) or accumulated into dL
    int n, m, mp, k, out;                   // out = output index being processed
    const double *y, *omega, *lnbeta, *wbeta; long ldx;
    long om_ld;                             // omega[out*om_ld + i]; 0 = an n x 1 omega for every output
    const double *w, *v;                    // column `out` of w (m), v (m) (v may be nullptr)
    double *dL;                             // n_pad x ld accumulator when k>1 (else nullptr)
    double *colslab;                        // [nwg][2][mp]: per-workgroup partial PHI'c, PHI'dbeta
    double *scal;                           // [nwg][4]: sum c*delta, sum omega*delta^2, sum LL, sum dbeta
    int nwg;
};
void launch_row_epilogue(hipStream_t st, const RowArgs &a);
void launch_colslab_reduce(hipStream_t st, const double *colslab, const double *scal, int nwg, int mp, double *out_cols,
                           double *out_scal);
void launch_mul_phi(hipStream_t st, const double *dL, const double *Phi, double *T, size_t count);

struct MomentArgs {
    const double *dPhi; int ld;
    const double *Xr;        // n_pad x d row-major
    int n, n_pad, m, d, kind;
    const double *P;
    int nchunk, rows_per_chunk;
    double *slab;            // [nchunk][m][nm]
    int nm;                  // moments per basis: cov d + d(d+1)/2, diag 2d (3d with Psi)
    const double *Psir, *Mr, *G2;   // diag kinds: Psi rows (n_pad x d), observed mask rows, gamma^2 (nullptr = absent)
    const int *chunktab;            // cov kinds, optional: {first row, end row} per chunk instead of chunk*rows_per_chunk
};
int launch_moments(hipStream_t st, const MomentArgs &a);

// Fused single-output path: row scalars from the T-GEMM epilogue, then moments with dPHI formed on the fly.
//   rowscal[i*4 + {0,1,2}] = omega*beta, omega*beta*delta, dbeta   (GPz.m:48,79,93)
//   partial: row_scalars_nwg(n) records of GPZ_NS doubles [sum c*delta, sum omega*delta^2, sum LL, sum dbeta, ...]
#define GPZ_ROWSCAL_MAX_NWG 1024
inline int row_scalars_nwg(int n) {   // one workgroup per 512 rows, 128 .. 1024 (the kernel is bound by load latency)
    const int w = (n + 511) / 512;
    return w < 128 ? 128 : (w > GPZ_ROWSCAL_MAX_NWG ? GPZ_ROWSCAL_MAX_NWG : w);
}
void launch_row_scalars(hipStream_t st, const double *nupart, int nslots, const double *phiw, const double *y,
                        const double *omega, const double *lnbeta, const double *wbeta, long n_pad, int n,
                        double *rowscal, double *partial, long om_off = 0);   // om_off: this output's column of an n x k omega (o * om_ld)
struct FusedMomentArgs {
    const double *Phi, *T; int ld;
    const double *Xr, *rowscal;
    int n, m, d, kind;
    const double *P, *w, *v;   // v may be nullptr
    int nchunk, rows_per_chunk;
    double *slab;              // [nchunk][m][nm + 2]: moments, then PHI'(omega beta delta), PHI'dbeta
    int nm;
    const double *Psir, *Mr, *G2;   // as in MomentArgs
    const int *chunktab;            // as in MomentArgs
};
int launch_moments_fused(hipStream_t st, const FusedMomentArgs &a);
int launch_moments_wide(hipStream_t st, const MomentArgs &a);
int launch_moments_fused_wide(hipStream_t st, const FusedMomentArgs &a);   // overwrites a.T with dPHI
// split the reduced [m][nm+2] records into mom [m][nm] and cols [2][mp]
void launch_split_fused(hipStream_t st, const double *rec, int m, int nm, int mp, double *mom, double *cols,
                        int accumulate = 0 /* add to mom instead of assigning (outputs after the first) */);
void launch_slab_sum(hipStream_t st, const double *slab, int nslab, size_t count, double *out);
// out[g][count] = sum of slabs seg[g] .. seg[g+1]-1, g < nseg (seg: nseg+1 device ints)
void launch_slab_sum_seg(hipStream_t st, const double *slab, const int *seg, int nseg, size_t count, double *out);

struct FinishArgs {
    int method_id, kind, m, d, k, hetero, g_dim;
    GpzParams pr;
    const double *mom; int nm;            // reduced moments [m][nm]
    const double *cols;                   // [k][2][mp]   PHI'(omega beta delta), PHI'dbeta
    const double *scal;                   // [k][4]
    const double *w, *dwda, *dgi;         // m x k each
    const double *logdet;                 // k
    const double *sums1;                  // [sum omega, sum_i omega_i lnbeta_io (8), n_train]       (GPZ_NS doubles)
    const double *vsums;                  // [sum omega delta^2, sum LL, per-output (8), n_valid, 0]  or nullptr
    const int *info;
    double *out;                          // [f, grad(p), stats(4), info, n, logdet0, svd-route flag, 0]   (p + 10 doubles)
    double *dGfull;                       // scratch m*d or m*d*d
    int p;
    int nmp;                              // leading dimension of cols (= mp)
    int de;                               // padded dimension of the parameter block / moments
    int psi;                              // diag kinds with input noise: moments are [A1|A2|A3]
    int gen;                              // cov kinds, general path: dP/dGamma already written by k_gen_finish
};
void launch_finish(hipStream_t st, const FinishArgs &a);

// Small row reductions; each writes GPZ_SMALL_NWG partial records of GPZ_NS doubles (sum them with launch_slab_sum).
//   row_stats: [sum omega*delta^2, sum omega*(-0.5 beta delta^2 + 0.5 ln beta), sum omega*beta*delta^2 per output (8), rows, 0]
//   sums1:     [sum omega, sum_i omega_i*lnbeta_io per output (8), 0, rows, 0]  -- rows at index 10 in both
#define GPZ_NS 12
// more than 8 outputs: the records grow by one slot per extra output, appended behind the 12 standard ones
__host__ __device__ inline int gpz_ns(int k) { return GPZ_NS + (k > 8 ? k - 8 : 0); }
__host__ __device__ inline int gpz_ns_idx(int base, int o) { return o < 8 ? base + o : GPZ_NS + (o - 8); }
#define GPZ_SMALL_NWG 128
void launch_row_stats(hipStream_t st, const double *phiw, const double *y, const double *omega, long om_ld, const double *lnbeta,
                      long ldx, int n, int k, double *partial);
void launch_sums1(hipStream_t st, const double *omega, long om_ld, const double *lnbeta, long ldx, int n, int k, double *partial);
// nlogML partial of the solve-only mode (GPz.m:81-82), one value per output.
void launch_solve_partial(hipStream_t st, GpzParams pr, const double *w, const double *logdet, const double *sums1,
                          const double *rstats, int m, int k, double *out);
// NaN-pattern grouping (getPHI.m:43-54): masks, first-occurrence unique list, ids.
size_t nan_groups_work_bytes(long n, int d);
int launch_nan_groups(hipStream_t st, const double *X, long n, int d, void *work /* nan_groups_work_bytes(n, d) */,
                      int *n_groups, int *group_id);

// ---- general covariance-kind path: input noise and/or missing dimensions (k_gen.hip) ---------------
struct GenRows {
    const double *Xr;          // n_pad x de, 0 at missing entries
    const int *gid;            // pattern id per row
    const double *Psi3;        // n_pad x d*d (Psi(:,:,i) column-major) or nullptr
    const int *rows_by_group;  // row indices sorted by pattern id
    int n, n_pad;
};
// ws: runtime-d workspace of gen_rt_threads(d) * gen_ws_per_thread(d) doubles, needed (non-null) when d > 20; the small
// matrices of the general path live there instead of per-thread scratch and the kernels run as grid-stride loops
size_t gen_ws_per_thread(int d);
int gen_rt_threads(int d);   // size of the grid-stride thread pool of the runtime-d kernels (workspace = threads x per-thread doubles)
void launch_gen_prep(hipStream_t st, const double *G, int m, int d, int de, double *Sig, double *iSig,
                     const unsigned char *pat, int ngroups, double *lnS, double *ws = nullptr);
// missing dimensions without input noise: per-pattern parameter block for the tuned PHI kernel ([R~ | c~], layout of
// k_prep_cov) and conversion of the tuned moment sums of a pattern into the records k_gen_finish consumes
void launch_gen_pattern_params(hipStream_t st, const double *Sig, const double *P, const unsigned char *pat, int G, int m,
                               int d, int de, double *RcAll /* G blocks of m*(nt+de) */, double *ws = nullptr);
void launch_gen_convert_moments(hipStream_t st, const double *frecAll /* [G][m][stride] */, int stride, int has_r,
                                const double *Sig, const unsigned char *pat, int G, int m, int d, int de,
                                double *recsAll /* [G][m][nrec] */, int nrec);
void launch_gen_phi(hipStream_t st, const GenRows &r, int m, int mp, int d, int de, int k, const double *P,
                    const double *Sig, const double *lnS, const unsigned char *pat, double *Phi, const double *Y,
                    double *ws = nullptr);
void launch_gen_fill(hipStream_t st, double *Phi, int ld, int n, int n_pad, int m, int mp, int k, const double *Y);
// register-resident variants for Psi without missing dimensions, 2 <= d <= 10 (k_psi.hip); return -1 outside that range
// fp64 GC/VC + Psi for 10 < d <= 64 (k_cpsi.hip): same arguments and records as launch_psi_phi / launch_psi_moments
bool cpsi_available(int d);
int launch_cpsi_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                    const double *lnS, double *Phi, int ld, const unsigned char *pat);
int launch_cpsi_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                        const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                        int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                        const int *chunktab);
// the same for 10 < d <= 32 with four pairs per wave (k_cpsi4.hip)
bool cpsi4_available(int d);
int launch_cpsi4_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                     const double *lnS, double *Phi, int ld, const unsigned char *pat, bool shared = false /* GC: one covariance */);
int launch_cpsi4_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                         const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                         int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                         const int *chunktab, const double *minv = nullptr);
size_t cpsi4_minv_len(int d);   // doubles per row of the GC inverse table (k_cpsi4_minv)
int launch_cpsi4_minv(hipStream_t st, const GenRows &r, int d, int de, const double *Sig, const double *lnS, const unsigned char *pat,
                      double *minv, double *qA = nullptr, int lda = 0, const double *ctr = nullptr);   // ctr (with qA): the centre c of launch_gcq_centre
// GC + Psi without missing dimensions, dense form of the PHI build: PHI = exp(-1/2 A * B), A (n x gcq_kpad(d)) from launch_cpsi4_minv(.., qA),
// B (gcq_kpad(d) x ldb) from launch_gcq_tab, the product on launch_tgemm
int gcq_kpad(int d);
void launch_gcq_centre(hipStream_t st, int m, int d, int de, const double *P, double *ctr);   // c = mean basis centre: both factors work on x - c, p - c
void launch_gcq_tab(hipStream_t st, int m, int d, int de, int ldb, const double *P, const double *ctr, double *B);
void launch_gcq_exp(hipStream_t st, const double *Q, int ld, int n, int m, double *Phi);
// 32 < d <= 48, rows without missing values (k_cpsi4w.hip)
bool cpsi4w_available(int d);
int launch_cpsi4w_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                      const double *lnS, double *Phi, int ld);
int launch_cpsi4w_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                          const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                          int nchunk, int rows_per_chunk, double *slab, int nrec, const int *chunktab);
int launch_cpsi4w_predict_noisy(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                                const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                                long pairs_per_chunk, double *part, bool shared);
int launch_cpsi4_predict_noisy(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                               const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                               long pairs_per_chunk, double *part, bool shared /* GC: one covariance for every pair */);
// prediction with missing values, GC/VC, 10 < d <= 32: the record sums on 4 x 4 MFMA tiles (k_pmc4.hip); arguments as pmc_sum (k_pmiss_cov.hip)
bool pmc4_available(int d);
bool launch_pmc4_sum(hipStream_t st, int d, bool noisy, int nrows, int row0, int m, int ld, long R, int nchunk, const double *tab,
                     int ntab, int nw, const double *Pio, const double *XhT, const double *PsT, double *Phi, long ldx,
                     double *part);
bool psi_fast_path_available(int d);
// pat (observed flags [G][d]) non-null: rows carry missing dimensions (r.gid = pattern per row, lnS = [G][m]);
// chunktab (optional): {first row, end row} per moment chunk
int launch_psi_phi(hipStream_t st, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                   const double *lnS, double *Phi, int ld, const unsigned char *pat, bool shared = false /* GC: every basis function has the same covariance */);
int launch_psi_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                       const double *v, const GenRows &r, int m, int d, int de, const double *P, const double *Sig,
                       int nchunk, int rows_per_chunk, double *slab, int nrec, const unsigned char *pat,
                       const int *chunktab, const double *gc_minv = nullptr);   // gc_minv: GC, 10 < d <= 32: the per-row inverse table (launch_cpsi4_minv)
// fp32 per-pair kernels for Psi without missing dimensions, d <= 20 (k_psi32.hip).  PsiT: packed lower triangles of
// Psi_i, element-major [e][ldp] (diag != 0: only the D diagonals), D = psi32_pad_dim(d).
int psi32_pad_dim(int d);
int psi32_raw_len(int d);   // doubles per (chunk, basis) the moment kernel writes: 3 + D + D(D+1)/2
// chunk-summed raw sums [m][psi32_raw_len] -> records [m][nrec]: the layout of k_gen_moments (diag = 0, consumed by
// k_gen_finish) or whitened records (diag = 1, consumed by launch_psi32_finish: dP = R'a, dGamma = -Q C~' R^-T)
void launch_psi32_records(hipStream_t st, const double *raw, int d, int diag, int m, double *recs, int nrec);
void launch_psi32_finish(hipStream_t st, const double *recs, int m, int d, int de, const double *Gam, const double *Rc,
                         int method_id, const double *sums1, int k, double *grad, double *dGfull, double *cols, int mp,
                         int nrec);
int launch_psi32_phi(hipStream_t st, const double *Xr, int de, int d, const float *PsiT, long ldp, int diag, int n, int m,
                     const double *P, const double *Sig, const double *Rc, const double *lnS, double *Phi, int ld);
// the same sums for DIAGONAL Psi on the matrix pipe (k_psi32m.hip): sixteen pairs per wave on v_mfma_f32_4x4x1_16b_f32; writes the raw
// layout of launch_psi32_moments (rows_per_chunk any positive number)
bool psi32m_available(int d);
int launch_psi32m_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                          const double *v, const double *Xr, int de, int d, const float *PsiT, long ldp, int n, int m,
                          const double *P, const double *Rc, int nchunk, int rows_per_chunk, double *slab);
int launch_psi32_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                         const double *v, const double *Xr, int de, int d, const float *PsiT, long ldp, int diag, int n, int m,
                         const double *P, const double *Sig, const double *Rc, int nchunk, int rows_per_chunk, double *slab,
                         int nrec);
void launch_gen_rowdot(hipStream_t st, const double *Phi, int ld, int n, long ldx, int m, int k, const double *v,
                       const double *b, const double *omega, const double *w, double *lnbeta, double *wbeta,
                       double *phiw, long om_ld = 0);   // omega[o*om_ld + i]
void launch_gen_moments(hipStream_t st, const double *Phi, const double *T, int ld, const double *rowscal, const double *w,
                        const double *v, const GenRows &r, int g, int row_begin, int nrows, const unsigned char *pat, int m,
                        int d, int de, const double *P, const double *Sig, int nchunk, int rows_per_chunk, double *slab,
                        int nrec, double *ws = nullptr);
void launch_gen_finish(hipStream_t st, const double *recs, int G, const unsigned char *pat, int m, int d, int de,
                       const double *Gam, const double *Sig, const double *iSig, int method_id, const double *sums1, int k,
                       double *grad, double *dGfull, double *cols, int mp, int nrec,
                       double *part /* G*m*(d + d*d + 2) doubles of scratch */,
                       int raw /* records hold plain moment sums (no input noise): see k_gen_convert_moments */,
                       double *ws = nullptr);

// prediction with input noise (predictDiag.m:75-125 / predictCov.m:70-132) and the getPrior iteration (getPrior.m:7-20)
void launch_pair_table(hipStream_t st, int kind, int m, int d, int de, const double *P, const double *G, const double *Sig,
                       const double *iSig, double *tab, int rec, double *ws = nullptr);
int launch_predict_noisy_cov(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psi3,
                             const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                             long pairs_per_chunk, double *part,
                             int flags = 0 /* bit 0: every Psi_i diagonal, bit 1: one covariance for all basis functions (GC) */);   // k_psi.hip: 2 <= d <= 10, else -1
int launch_predict_noisy_diag(hipStream_t st, int n, long ldx, int m, int d, int de, int k, const double *Xr, const double *Psir,
                              const double *tab, int rec, const double *w, const double *v, const double *iS, int nchunk,
                              long pairs_per_chunk, double *part);   // k_psi.hip: d <= 20, k <= 8, else -1
void launch_predict_noisy(hipStream_t st, int kind, int n, long ldx, int m, int d, int de, int k, const double *Xr,
                          const double *Psir, const double *Psi3, const double *tab, int rec, const double *w,
                          const double *v, const double *iS, int nchunk, long pairs_per_chunk, double *part,
                          double *ws = nullptr, int flags = 0);
void launch_predict_noisy_final(hipStream_t st, const double *sums, long ldx, int n, int k, const double *mu,
                                const double *lnbeta, const double *b, double *gamma, double *nu, double *beta_i);
void launch_prior_update(hipStream_t st, const double *colsum, double ns, int m, double *prior, double *keep);
void launch_prior_iter(hipStream_t st, const double *N, int ld, int n, int m, const double *prior, double *colslab,
                       int nwg);

// prediction with missing dimensions, diagonal kinds (predictDiag.m:127-297; k_pmiss.hip).  obs: bit c set = dimension c observed
// (ObsMask: 256 bits, passed by value; the covariance kinds keep a 64-bit mask, d <= 64).
struct ObsMask {
    unsigned long long w[4];
};
#define GPZ_PM_MAXD 256        // bits of the mask
#define GPZ_PM_MAXD_DIAG 144   // what the pair-table kernel's LDS tile (d KB per 64 pairs) allows
static inline __host__ __device__ bool obs_bit(const ObsMask &o, int c) { return (o.w[c >> 6] >> (c & 63)) & 1ull; }
struct ObsFlags {               // any width: one byte per dimension in DEVICE memory (1 = observed); the kernels' wide instantiations
    const unsigned char *f;
};
static inline __device__ bool obs_bit(const ObsFlags &o, int c) { return o.f[c] != 0; }
void launch_pm_no(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int m, int ld, int d,
                  ObsMask obs, const double *P, const double *G, const double *priors, double *No, double *Pio);
void launch_pm_no(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int m, int ld, int d,
                  ObsFlags obs, const double *P, const double *G, const double *priors, double *No, double *Pio);
void launch_pm_pio(hipStream_t st, const double *No, int ld, int n, int m, const double *priors, double *Pio);
// covariance kinds (predictCov.m:134-337; k_pmiss_cov.hip): see launch_pmc for the work buffers
int pmc_rec_len(int d, unsigned long long obs);
void launch_pmc(hipStream_t st, unsigned long long obs, int n, long ldx, int m, int ld, int d, int de, int k, const double *Xr,
                const double *Psi3, const double *P, const double *Sig, const double *iSig, const double *priors,
                const double *w, const double *v, const double *iS, int rows_blk, double *rec, double *tab, double *Ex,
                double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                double *work2 = nullptr /* m * (d(d+1)/2 + d*d + d + 1) doubles: enables the register-resident route */,
                bool tab_ready = false /* the pair table (pattern-independent) is already in `tab`: kept from the previous group */);
// the same for 32 < d <= 64: scratch-resident kernels only (k_pmiss_cov64.hip); work2 is not used
void launch_pmc_wide(hipStream_t st, unsigned long long obs, int n, long ldx, int m, int ld, int d, int de, int k, const double *Xr,
                     const double *Psi3, const double *P, const double *Sig, const double *iSig, const double *priors,
                     const double *w, const double *v, const double *iS, int rows_blk, double *rec, double *tab, double *Ex,
                     double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                     double *work2, bool tab_ready);
bool pmc_fast(int d, int k);   // 2 <= d <= 10, k <= 8: the register-resident kernels (needs rows_blk <= 64)
void launch_pm_nij(hipStream_t st, int m, int ld, int d, int de, ObsMask obs, const double *P, const double *G, double *B);
void launch_pm_phi(hipStream_t st, const double *No, const double *T1, int ld, int n, long n_pad, int m, int d, int de,
                   const double *G, double *Phi);
void launch_pm_pairtab(hipStream_t st, long q0, long npairs, int m, int ld, int width, int d, int de, int k, ObsMask obs,
                       int has_psi, const double *P, const double *G, const double *w, const double *v, const double *iS,
                       double *B, double *rec, int nrec);
void launch_pm_accum(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int ld, int d, int k,
                     ObsMask obs, int npq, const double *T2, const double *rec, int nrec,
                     double *sums /* [nsplit][3k][n_pad] */, int nsplit = 1);
int pm_accum_splits(int n);
// the same four for inputs of any width (d > GPZ_PM_MAXD_DIAG): pattern as device flags, no LDS tiles
void launch_pm_nij(hipStream_t st, int m, int ld, int d, int de, ObsFlags obs, const double *P, const double *G, double *B);
void launch_pm_pairtab(hipStream_t st, long q0, long npairs, int m, int ld, int width, int d, int de, int k, ObsFlags obs,
                       int has_psi, const double *P, const double *G, const double *w, const double *v, const double *iS,
                       double *B, double *rec, int nrec);
void launch_pm_accum(hipStream_t st, const double *Xr, const double *Psir, int de, int n, long n_pad, int ld, int d, int k,
                     ObsFlags obs, int npq, const double *T2, const double *rec, int nrec, double *sums, int nsplit);
// covariance kinds at d > 64 (k_pmiss_covg.hip): the per-thread d x d temporaries live in a workspace in device memory.
//   pat_dev: 3 d ints (observed | missing | unshuffle), filled by the launcher;  ws: ws_threads * pmg_ws_per_thread(d) doubles.
// Work buffers as for launch_pmc (rows_blk = 1 is enough: one launch per row and item range).
size_t pmg_ws_per_thread(int d);
void launch_pmc_generic(hipStream_t st, const unsigned char *obs_host, int n, long ldx, int m, int ld, int d, int de, int k,
                        const double *Xr, const double *Psi3, const double *P, const double *Sig, const double *iSig,
                        const double *priors, const double *w, const double *v, const double *iS, double *rec, double *tab, double *Ex,
                        double *Pio, double *Xhat, double *Phat, int nchunk, long pairs_per_chunk, double *part, double *Phi,
                        bool tab_ready, int *pat_dev, double *ws, long ws_threads);

// N = PHI .* exp(-1/2 ln|Sigma_oo| - 1/2 |o| ln 2pi + 1/2 |u| ln 2)   (getPHI.m:77,87,98,105,114)
struct NormArgs {
    const double *Phi; int ld; int n, m, d, de, kind, gen;
    const double *G;        // diag kinds: gamma [m][de]
    const double *Rc;       // cov kinds, tuned path: QR factor records
    const double *Mr, *ucnt;// diag kinds with missing values (row-major mask, missing count) or nullptr
    const int *gid; const unsigned char *pat; const double *lnS;   // general cov path
    double *N;              // n_pad x ld row-major
};
void launch_phi_norm(hipStream_t st, const NormArgs &a);

// misc
void launch_dxy(hipStream_t st, const double *X, long nx, const double *Y, long ny, int d, double *D);
void launch_transpose_out(hipStream_t st, const double *src, int ld, long n, int m, double *dst /* n x m col-major */,
                          const int *perm = nullptr /* source row i -> destination row perm[i] */);
void launch_nu(hipStream_t st, const double *Phi, const double *T, int ld, int n, int m, double *nu);

// Run-time options of libgpz_hip.so.
//
// The environment is read in ONE place (gpz_options_load, gpz_options.hip).  A context loads its options when it is created and
// keeps them: every evaluation of that context, its captured hipGraph and gpz_ctx_route agree on what runs.  Entry points that
// take no context load a snapshot per call.  Code below the entry points (launchers in the k_*.hip files) asks gpz_opts() for
// the options of the call in progress on this thread.
//
// User-facing switches (README.md):
//   GPZ_NO_GRAPH=1                    eager launches instead of one hipGraph replay per evaluation
//   GPZ_GRAPH_DEBUG=1                 report a failed graph capture on stderr (the evaluation falls back to eager launches)
//   GPZ_ROW_TILE=<rows>               stream PHI / T in row tiles of this size even when they would fit the device
//   GPZ_CACHE_CAP_MB=<MB>             cap of the released-buffer cache per device (0: none); read once per process
//   GPZ_PSI32_MFMA=1                  config-5 moment sums on 4 x 4 MFMA tiles (k_psi32m.hip; measured slower, opt-in)
//   GPZ_PREDICT_MIN_ROWS_PER_BLOCK=<rows>  smallest NaN-pattern group gpz_mgpu_predict splits over devices (default 4096)
// Developer A/B switches exist only in builds with -DGPZ_DEV_SWITCHES (./build.sh --dev -> libgpz_hip_dev.so, loaded by the tests that
// compare routes through GPZ_HIP_LIB); in the release library their fields keep the defaults below.
#pragma once

struct gpz_options {
    bool no_graph = false;
    bool graph_debug = false;
    long row_tile = 0;
    long cache_cap_mb = -1;   // -1: the built-in default
    bool psi32_mfma = false;
    long predict_min_rows_per_block = 4096;
    // ---- developer switches (GPZ_DEV_SWITCHES) ----
    bool f32_contractions_off = false;   // GPZ_F32_CONTRACTIONS_OFF  config 5: fp64-operand MFMA contractions beside the fp32 pair kernels
    bool gc_minv_off = false;            // GPZ_GC_MINV_OFF           GC + Psi: no per-row inverse of Sigma + Psi_i
    bool gc_dense_phi_off = false;       // GPZ_GC_DENSE_PHI_OFF      GC + Psi: no dense-product PHI build
    bool tgemm_no_split = false;         // GPZ_TGEMM_NO_SPLIT        k_tgemm: no column pieces in the last round
    bool phi_diag_no_split = false;      // GPZ_PHI_DIAG_NO_SPLIT     diagonal-kind PHI build: no split over basis functions
    bool pmc_scratch = false;            // GPZ_PMC_SCRATCH           predictMissing GC/VC: scratch-resident kernels
    bool pmc_prep_scratch = false;       // GPZ_PMC_PREP_SCRATCH
    bool pmc_no_model_cache = false;     // GPZ_PMC_NO_MODEL_CACHE    predictMissing GC/VC: rebuild the model tables per call
    bool round_phi32 = false;            // GPZ_EXPERIMENT_ROUND_PHI32  (tools/f32_operand_experiment.py)
    bool cpsi_off = false;               // GPZ_CPSI_OFF              fp64 GC/VC + Psi: general kernels instead of 16 x 16 tiles
    bool cpsi4_off = false;              // GPZ_CPSI4_OFF             ... instead of 4 x 4 tiles
    bool tgemm_int8 = false;             // GPZ_TGEMM_INT8            T = PHI [inv(SIGMA) | w] as 28 int8 products of digit planes (k_oz.hip: measured, not faster)
    int syrk_wgs = 0, syrk_s1 = 0, syrk_s2 = 0;   // GPZ_SYRK_WGS / _S1 / _S2   row-split tuning of k_syrk
    int mom_nc = 0;                      // GPZ_MOM_NC                chunk count of the moment kernels
    int small_stagger = 0;               // GPZ_SMALL_STAGGER         k_small_tail: start delay of a compute unit's second workgroup (x 8128 cycles; 0 = default)
    bool chol_rowinv_off = false;        // GPZ_CHOL_ROWINV_OFF       m + k <= 256: k_trtri_level launches behind the factorisation instead of inv(L) row by row inside it
    bool syrk_small_off = false;         // GPZ_SYRK_SMALL_OFF        m + k <= 256: k_syrk's 128 x 128 tiles instead of k_syrk_small
    bool small_tail_off = false;         // GPZ_SMALL_TAIL_OFF        m + k <= 256: k_tgemm + k_row_scalars + k_moments_fused instead of k_small_tail
    int debug_fail_cut = 0;              // GPZ_DEBUG_FAIL_CUT=k      test hook: the k-th segment cut of a graph recording fails after its hipStreamEndCapture
};

gpz_options gpz_options_load();   // from the environment

// options of the call in progress on this thread: the context's (or the per-call snapshot of a context-free entry point); a
// process-wide snapshot, loaded once, when no entry point is on the stack
const gpz_options &gpz_opts();
struct gpz_opts_scope {
    const gpz_options *prev;
    explicit gpz_opts_scope(const gpz_options *o);
    ~gpz_opts_scope();
    gpz_opts_scope(const gpz_opts_scope &) = delete;
    gpz_opts_scope &operator=(const gpz_opts_scope &) = delete;
};

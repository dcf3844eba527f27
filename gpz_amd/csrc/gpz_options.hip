// The one place libgpz_hip.so reads the environment (see gpz_options.h).
#include <stdlib.h>
#include "gpz_options.h"

static const char *env(const char *name) { return getenv(name); }
static bool env_set(const char *name) { return env(name) != nullptr; }
static long env_long(const char *name, long dflt) {
    const char *e = env(name);
    return e ? atol(e) : dflt;
}

gpz_options gpz_options_load() {
    gpz_options o;
    o.no_graph = env_set("GPZ_NO_GRAPH");
    o.graph_debug = env_set("GPZ_GRAPH_DEBUG");
    o.row_tile = env_long("GPZ_ROW_TILE", 0);
    o.cache_cap_mb = env_long("GPZ_CACHE_CAP_MB", -1);
    {
        const char *e = env("GPZ_PSI32_MFMA");
        o.psi32_mfma = e && e[0] == '1';
    }
    o.predict_min_rows_per_block = env_long("GPZ_PREDICT_MIN_ROWS_PER_BLOCK", 4096);
    if (o.predict_min_rows_per_block < 1) o.predict_min_rows_per_block = 1;
#ifdef GPZ_DEV_SWITCHES
    o.f32_contractions_off = env_set("GPZ_F32_CONTRACTIONS_OFF");
    o.gc_minv_off = env_set("GPZ_GC_MINV_OFF");
    o.gc_dense_phi_off = env_set("GPZ_GC_DENSE_PHI_OFF");
    o.tgemm_no_split = env_set("GPZ_TGEMM_NO_SPLIT");
    o.phi_diag_no_split = env_set("GPZ_PHI_DIAG_NO_SPLIT");
    o.pmc_scratch = env_set("GPZ_PMC_SCRATCH");
    o.pmc_prep_scratch = env_set("GPZ_PMC_PREP_SCRATCH");
    o.pmc_no_model_cache = env_set("GPZ_PMC_NO_MODEL_CACHE");
    o.round_phi32 = env_set("GPZ_EXPERIMENT_ROUND_PHI32");
    o.cpsi_off = env_set("GPZ_CPSI_OFF");
    o.cpsi4_off = env_set("GPZ_CPSI4_OFF");
    o.tgemm_int8 = env_set("GPZ_TGEMM_INT8");
    o.syrk_wgs = (int)env_long("GPZ_SYRK_WGS", 0);
    o.syrk_s1 = (int)env_long("GPZ_SYRK_S1", 0);
    o.syrk_s2 = (int)env_long("GPZ_SYRK_S2", 0);
    o.mom_nc = (int)env_long("GPZ_MOM_NC", 0);
    o.debug_fail_cut = (int)env_long("GPZ_DEBUG_FAIL_CUT", 0);
    o.small_tail_off = env_set("GPZ_SMALL_TAIL_OFF");
    o.syrk_small_off = env_set("GPZ_SYRK_SMALL_OFF");
    o.chol_rowinv_off = env_set("GPZ_CHOL_ROWINV_OFF");
    o.small_stagger = (int)env_long("GPZ_SMALL_STAGGER", 0);
#endif
    return o;
}

static thread_local const gpz_options *t_opts = nullptr;
const gpz_options &gpz_opts() {
    if (t_opts) return *t_opts;
    static const gpz_options process = gpz_options_load();
    return process;
}
gpz_opts_scope::gpz_opts_scope(const gpz_options *o) : prev(t_opts) { t_opts = o; }
gpz_opts_scope::~gpz_opts_scope() { t_opts = prev; }

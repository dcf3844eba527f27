// predictMissing / predictNoisyMissing for GC/VC at 32 < d <= 64 (predictCov.m:134-337): the scratch-resident kernels of
// k_pmiss_cov.hip compiled with 64-wide temporaries under their own names (launch_pmc_wide).  The reference is generic in d; this is
// the slow, correct route for inputs the register / MFMA routes of that file do not cover.
#define GDM 64
#define PMC_WIDE 1
#include "k_pmiss_cov.hip"

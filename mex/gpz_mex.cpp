// gpz_mex.cpp — MEX gateway between MATLAB and libgpz_hip.so (pure marshalling, no arithmetic).
//
// Build on a machine that has MATLAB (not available in the build image, so this file is source only):
//     mex -R2018a gpz_mex.cpp -I../include -L../gpz_amd/lib -lgpz_hip
// Same gateway convention as the reference's own MEX files (minFunc_2012/minFunc/mex/lbfgsProdC.c:7).
//
//   gpz_mex('create', model, X, Y, Psi, omega, training, validation)   once per closure (train.m:40)
//   [f, g, stats] = gpz_mex('eval', theta)                             GPz.m nargout<=2
//   [w, iSigma_w, part] = gpz_mex('solve', theta)                      GPz.m nargout>2 (GPz.m:84-87)
//   PHI = gpz_mex('phi')                                               5th output of GPz.m:1
//   [PHI, lnBeta_i, N] = gpz_mex('getphi', model, theta, X, Psi)       getPHI.m:1 (rows already selected)
//   [mu,nu,beta_i,gamma,PHI] = gpz_mex('predict', model, theta, w, iSigma_w, priors, X, Psi)
//                                                  one NaN-pattern group of predict.m:60-69: predictFull / predictNoisy /
//                                                  predictMissing / predictNoisyMissing by what X and Psi contain
//   prior = gpz_mex('prior', model, theta, X, Psi)                     getPrior.m:1
//   [Xi, logdet] = gpz_mex('inv_logdet', A)                            inv_logdet.m:1
//   D = gpz_mex('dxy', X, Y)                                           Dxy.m:1
//   gpz_mex('pinv_mode', mode)                                         branch of inv_logdet.m:7-12 (0 auto, 1 always, -1 never)
//   gpz_mex('reset')
// model.dtype = 'f32' (optional field) selects the fp32 per-pair factorisations of GC/VC with input noise.
#include <string.h>
#include "mex.h"
#include "gpz_hip.h"

static gpz_ctx *g_ctx = NULL;
static int g_m = 0, g_k = 0;

static void cleanup(void) {
    if (g_ctx) { gpz_ctx_destroy(g_ctx); g_ctx = NULL; }
}
static const double *opt(const mxArray *a) { return (a && !mxIsEmpty(a)) ? mxGetPr(a) : NULL; }
static const uint8_t *optmask(const mxArray *a) {
    if (!a || mxIsEmpty(a)) return NULL;
    if (!mxIsLogical(a)) mexErrMsgIdAndTxt("gpz:type", "masks must be logical");
    return (const uint8_t *)mxGetLogicals(a);
}
static double field(const mxArray *s, const char *name) {
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("gpz:model", "model.%s missing", name);
    return mxGetScalar(f);
}

static gpz_desc desc_of(const mxArray *model) {
    gpz_desc d;
    memset(&d, 0, sizeof d);
    d.d = (int32_t)field(model, "d"); d.m = (int32_t)field(model, "m"); d.k = (int32_t)field(model, "k");
    d.heteroscedastic = (int32_t)field(model, "heteroscedastic");
    mxGetString(mxGetField(model, 0, "method"), d.method, sizeof d.method);
    d.world = 1;
    const mxArray *dt = mxGetField(model, 0, "dtype");
    char buf[8] = "";
    if (dt && !mxGetString(dt, buf, sizeof buf) && !strcmp(buf, "f32")) d.dtype = GPZ_F32;
    return d;
}
static int psi_kind_of(const mxArray *Psi) {   /* fixPsi.m layouts: [] / n x d / d x d x n */
    return (!Psi || mxIsEmpty(Psi)) ? 0 : (mxGetNumberOfDimensions(Psi) == 3 ? 2 : 1);
}
static int has_nan(const mxArray *X) {
    const double *x = mxGetPr(X);
    for (size_t i = 0, n = mxGetNumberOfElements(X); i < n; ++i)
        if (mxIsNaN(x[i])) return 1;
    return 0;
}
#define CHECK(call, id) do { if (call) mexErrMsgIdAndTxt(id, "%s", gpz_last_error()); } while (0)

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[16];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("gpz:usage", "first argument: command");
    if (!strcmp(cmd, "reset")) { cleanup(); return; }
    /* ---- stand-alone entries (no context) ---- */
    if (!strcmp(cmd, "getphi")) {
        if (nrhs != 5) mexErrMsgIdAndTxt("gpz:usage", "getphi needs model,theta,X,Psi");
        gpz_desc d = desc_of(prhs[1]);
        const mwSize ns = mxGetM(prhs[3]);
        plhs[0] = mxCreateDoubleMatrix(ns, d.m, mxREAL);
        mxArray *lb = mxCreateDoubleMatrix(ns, d.k, mxREAL), *N = nlhs > 2 ? mxCreateDoubleMatrix(ns, d.m, mxREAL) : NULL;
        CHECK(gpz_phi(&d, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)ns, opt(prhs[4]), psi_kind_of(prhs[4]),
                      mxGetPr(plhs[0]), mxGetPr(lb), N ? mxGetPr(N) : NULL), "gpz:getphi");
        if (nlhs > 1) plhs[1] = lb; else mxDestroyArray(lb);
        if (nlhs > 2) plhs[2] = N;
        return;
    }
    if (!strcmp(cmd, "predict")) {
        if (nrhs != 8) mexErrMsgIdAndTxt("gpz:usage", "predict needs model,theta,w,iSigma_w,priors,X,Psi");
        gpz_desc d = desc_of(prhs[1]);
        const mxArray *X = prhs[6], *Psi = prhs[7];
        const mwSize ns = mxGetM(X);
        mxArray *o[5];
        for (int q = 0; q < 4; ++q) o[q] = mxCreateDoubleMatrix(ns, d.k, mxREAL);   /* mu nu beta_i gamma (gamma = 0 for predictFull) */
        o[4] = mxCreateDoubleMatrix(ns, d.m, mxREAL);
        const double *th = mxGetPr(prhs[2]), *w = mxGetPr(prhs[3]), *iS = mxGetPr(prhs[4]);
        if (has_nan(X))
            CHECK(gpz_predict_missing(&d, th, w, iS, mxGetPr(prhs[5]), mxGetPr(X), (int64_t)ns, opt(Psi), psi_kind_of(Psi),
                                      mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]), mxGetPr(o[3]), mxGetPr(o[4])), "gpz:predict");
        else if (psi_kind_of(Psi))
            CHECK(gpz_predict_noisy(&d, th, w, iS, mxGetPr(X), (int64_t)ns, mxGetPr(Psi), psi_kind_of(Psi), mxGetPr(o[0]),
                                    mxGetPr(o[1]), mxGetPr(o[2]), mxGetPr(o[3]), mxGetPr(o[4])), "gpz:predict");
        else
            CHECK(gpz_predict_full(&d, th, w, iS, mxGetPr(X), (int64_t)ns, mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]),
                                   mxGetPr(o[4])), "gpz:predict");
        for (int q = 0; q < 5; ++q)
            if (q < nlhs || q == 0) plhs[q] = o[q]; else mxDestroyArray(o[q]);
        return;
    }
    if (!strcmp(cmd, "prior")) {
        if (nrhs != 5) mexErrMsgIdAndTxt("gpz:usage", "prior needs model,theta,X,Psi");
        gpz_desc d = desc_of(prhs[1]);
        plhs[0] = mxCreateDoubleMatrix(1, d.m, mxREAL);
        CHECK(gpz_prior(&d, mxGetPr(prhs[2]), mxGetPr(prhs[3]), (int64_t)mxGetM(prhs[3]), opt(prhs[4]), psi_kind_of(prhs[4]),
                        mxGetPr(plhs[0]), NULL), "gpz:prior");
        return;
    }
    if (!strcmp(cmd, "inv_logdet")) {
        const mwSize m = mxGetM(prhs[1]);
        double ld = 0.0;
        plhs[0] = mxCreateDoubleMatrix(m, m, mxREAL);
        CHECK(gpz_inv_logdet(mxGetPr(prhs[1]), (int32_t)m, 0, mxGetPr(plhs[0]), &ld, NULL), "gpz:inv_logdet");
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(ld);
        return;
    }
    if (!strcmp(cmd, "dxy")) {
        plhs[0] = mxCreateDoubleMatrix(mxGetM(prhs[1]), mxGetM(prhs[2]), mxREAL);
        CHECK(gpz_dxy(mxGetPr(prhs[1]), (int64_t)mxGetM(prhs[1]), mxGetPr(prhs[2]), (int64_t)mxGetM(prhs[2]),
                      (int32_t)mxGetN(prhs[1]), 0, mxGetPr(plhs[0])), "gpz:dxy");
        return;
    }
    if (!strcmp(cmd, "create")) {
        if (nrhs != 8) mexErrMsgIdAndTxt("gpz:usage", "create needs model,X,Y,Psi,omega,training,validation");
        cleanup();
        gpz_desc d = desc_of(prhs[1]);
        const mxArray *Psi = prhs[4];
        int psi_kind = psi_kind_of(Psi);
        if (gpz_ctx_create(&d, (int64_t)mxGetM(prhs[2]), mxGetPr(prhs[2]), mxGetPr(prhs[3]), opt(Psi), psi_kind,
                           opt(prhs[5]), optmask(prhs[6]), optmask(prhs[7]), &g_ctx))
            mexErrMsgIdAndTxt("gpz:create", "%s", gpz_last_error());
        g_m = d.m; g_k = d.k;
        mexLock();
        mexAtExit(cleanup);
        return;
    }
    if (!g_ctx) mexErrMsgIdAndTxt("gpz:state", "call gpz_mex('create', ...) first");
    if (!strcmp(cmd, "eval")) {
        mwSize p = (mwSize)gpz_theta_len(g_ctx);
        if (nrhs != 2 || !mxIsDouble(prhs[1]) || mxGetNumberOfElements(prhs[1]) != p)
            mexErrMsgIdAndTxt("gpz:theta", "theta must be a double vector of %d elements", (int)p);
        double f;
        plhs[1 < nlhs ? 1 : 0] = NULL;
        mxArray *g = mxCreateDoubleMatrix(p, 1, mxREAL);
        mxArray *st = mxCreateDoubleMatrix(4, 1, mxREAL);
        mxGetPr(st)[2] = mxGetNaN(); mxGetPr(st)[3] = mxGetNaN();
        if (gpz_eval(g_ctx, mxGetPr(prhs[1]), &f, mxGetPr(g), mxGetPr(st), NULL))
            mexErrMsgIdAndTxt("gpz:eval", "%s", gpz_last_error());
        plhs[0] = mxCreateDoubleScalar(f);
        if (nlhs > 1) plhs[1] = g; else mxDestroyArray(g);
        if (nlhs > 2) plhs[2] = st; else mxDestroyArray(st);
        return;
    }
    if (!strcmp(cmd, "solve")) {
        mwSize dims[3] = {(mwSize)g_m, (mwSize)g_m, (mwSize)g_k};
        plhs[0] = mxCreateDoubleMatrix(g_m, g_k, mxREAL);
        mxArray *iS = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL);
        mxArray *part = mxCreateDoubleMatrix(1, g_k, mxREAL);
        if (gpz_solve(g_ctx, mxGetPr(prhs[1]), mxGetPr(plhs[0]), mxGetPr(iS), mxGetPr(part)))
            mexErrMsgIdAndTxt("gpz:solve", "%s", gpz_last_error());
        if (nlhs > 1) plhs[1] = iS; else mxDestroyArray(iS);
        if (nlhs > 2) plhs[2] = part; else mxDestroyArray(part);
        return;
    }
    if (!strcmp(cmd, "pinv_mode")) {
        CHECK(gpz_ctx_set_pinv_mode(g_ctx, (int)mxGetScalar(prhs[1])), "gpz:pinv_mode");
        return;
    }
    if (!strcmp(cmd, "phi")) {
        plhs[0] = mxCreateDoubleMatrix((mwSize)gpz_n_train(g_ctx), g_m, mxREAL);
        if (gpz_get_phi(g_ctx, mxGetPr(plhs[0]))) mexErrMsgIdAndTxt("gpz:phi", "%s", gpz_last_error());
        return;
    }
    mexErrMsgIdAndTxt("gpz:usage", "unknown command '%s'", cmd);
}

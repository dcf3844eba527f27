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
//   gpz_mex('reset')
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

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    char cmd[16];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("gpz:usage", "first argument: command");
    if (!strcmp(cmd, "reset")) { cleanup(); return; }
    if (!strcmp(cmd, "create")) {
        if (nrhs != 8) mexErrMsgIdAndTxt("gpz:usage", "create needs model,X,Y,Psi,omega,training,validation");
        cleanup();
        gpz_desc d;
        memset(&d, 0, sizeof d);
        d.d = (int32_t)field(prhs[1], "d"); d.m = (int32_t)field(prhs[1], "m"); d.k = (int32_t)field(prhs[1], "k");
        d.heteroscedastic = (int32_t)field(prhs[1], "heteroscedastic");
        mxGetString(mxGetField(prhs[1], 0, "method"), d.method, sizeof d.method);
        d.world = 1;
        const mxArray *Psi = prhs[4];
        int psi_kind = mxIsEmpty(Psi) ? 0 : (mxGetNumberOfDimensions(Psi) == 3 ? 2 : 1);   /* fixPsi.m layouts */
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
    if (!strcmp(cmd, "phi")) {
        plhs[0] = mxCreateDoubleMatrix((mwSize)gpz_n_train(g_ctx), g_m, mxREAL);
        if (gpz_get_phi(g_ctx, mxGetPr(plhs[0]))) mexErrMsgIdAndTxt("gpz:phi", "%s", gpz_last_error());
        return;
    }
    mexErrMsgIdAndTxt("gpz:usage", "unknown command '%s'", cmd);
}

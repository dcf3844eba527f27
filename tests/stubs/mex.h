/* Declarations-only stand-in for MATLAB's mex.h: just the part of the MEX C API that mex/gpz_mex.cpp uses, with the
 * documented MATLAB signatures (R2017b "separate complex" API).  TEST INFRASTRUCTURE: it exists so that the gateway —
 * our own file — can be compiled in an image without MATLAB (SURVEY.md 8b "testable twin"), and, together with
 * mex_runtime.cpp, executed against libgpz_hip.so on the GPU box.  It is not used to build anything of the reference. */
#ifndef GPZ_TEST_MEX_H
#define GPZ_TEST_MEX_H
#include <stddef.h>
#include <stdbool.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS,
               mxDOUBLE_CLASS, mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS,
               mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS } mxClassID;

double *mxGetPr(const mxArray *a);
void *mxGetData(const mxArray *a);
mxLogical *mxGetLogicals(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
size_t mxGetElementSize(const mxArray *a);
bool mxIsEmpty(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsComplex(const mxArray *a);
bool mxIsLogical(const mxArray *a);
bool mxIsStruct(const mxArray *a);
bool mxIsNaN(double v);
double mxGetNaN(void);
mxArray *mxGetField(const mxArray *s, mwIndex index, const char *name);
double mxGetScalar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity flag);
void mxDestroyArray(mxArray *a);

void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
void mexLock(void);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#ifdef __cplusplus
}
#endif
#endif

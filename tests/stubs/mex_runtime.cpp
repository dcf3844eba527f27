// A small stand-in for the MATLAB MEX runtime: the functions tests/stubs/mex.h declares, implemented on plain heap
// arrays, plus a C driver (mexrt_*) that lets the Python tests build argument lists and call the gateway's mexFunction.
// TEST INFRASTRUCTURE: it lets mex/gpz_mex.cpp — our own file — be compiled, linked against libgpz_hip.so and EXECUTED
// in an image without MATLAB (tests/test_mex_gateway.py).  Nothing of the reference is built with it.
// mexErrMsgIdAndTxt leaves mexFunction the way MATLAB's does (it does not return): here by throwing, caught in mexrt_call.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    std::vector<mwSize> dims{0, 0};
    void *data = nullptr;
    size_t esize = 8;
    std::map<std::string, mxArray *> fields;   // struct (1 x 1 only)
    std::string chars;                         // char row vector
};
struct MexError {
    std::string id, msg;
};
static std::string g_last_id, g_last_msg;
static void (*g_atexit)(void) = nullptr;
static int g_locks = 0;

static size_t numel(const mxArray *a) {
    if (a->cls == mxCHAR_CLASS) return a->chars.size();
    size_t n = 1;
    for (mwSize d : a->dims) n *= d;
    return n;
}
static mxArray *make(mxClassID cls, size_t esize, const std::vector<mwSize> &dims) {
    mxArray *a = new mxArray_tag();
    a->cls = cls;
    a->esize = esize;
    a->dims = dims;
    size_t n = numel(a);
    a->data = n ? calloc(n, esize) : nullptr;
    return a;
}

extern "C" {
double *mxGetPr(const mxArray *a) { return a->cls == mxDOUBLE_CLASS ? (double *)a->data : nullptr; }
void *mxGetData(const mxArray *a) { return a->data; }
mxLogical *mxGetLogicals(const mxArray *a) { return a->cls == mxLOGICAL_CLASS ? (mxLogical *)a->data : nullptr; }
size_t mxGetM(const mxArray *a) { return a->cls == mxCHAR_CLASS ? (a->chars.empty() ? 0 : 1) : a->dims[0]; }
size_t mxGetN(const mxArray *a) {          // product of the trailing dimensions, like MATLAB's
    if (a->cls == mxCHAR_CLASS) return a->chars.size();
    size_t n = 1;
    for (size_t q = 1; q < a->dims.size(); ++q) n *= a->dims[q];
    return n;
}
size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->cls == mxCHAR_CLASS ? 2 : a->dims.size(); }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims.data(); }
size_t mxGetElementSize(const mxArray *a) { return a->cls == mxCHAR_CLASS ? 2 : a->esize; }
bool mxIsEmpty(const mxArray *a) { return a->cls == mxSTRUCT_CLASS ? false : numel(a) == 0; }
bool mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsComplex(const mxArray *) { return false; }
bool mxIsLogical(const mxArray *a) { return a->cls == mxLOGICAL_CLASS; }
bool mxIsStruct(const mxArray *a) { return a->cls == mxSTRUCT_CLASS; }
bool mxIsNaN(double v) { return v != v; }
double mxGetNaN(void) { return NAN; }
mxArray *mxGetField(const mxArray *s, mwIndex index, const char *name) {
    if (s->cls != mxSTRUCT_CLASS || index != 0) return nullptr;
    auto it = s->fields.find(name);
    return it == s->fields.end() ? nullptr : it->second;
}
double mxGetScalar(const mxArray *a) {
    if (a->cls == mxDOUBLE_CLASS && numel(a)) return ((double *)a->data)[0];
    if (a->cls == mxLOGICAL_CLASS && numel(a)) return ((mxLogical *)a->data)[0] ? 1.0 : 0.0;
    return 0.0;
}
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {   // 0 on success, 1 on failure / truncation (MATLAB's contract)
    if (!a || a->cls != mxCHAR_CLASS || !buflen) return 1;
    const size_t n = a->chars.size() < buflen - 1 ? a->chars.size() : buflen - 1;
    memcpy(buf, a->chars.data(), n);
    buf[n] = 0;
    return a->chars.size() > buflen - 1 ? 1 : 0;
}
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity) { return make(mxDOUBLE_CLASS, 8, {m, n}); }
mxArray *mxCreateDoubleScalar(double v) {
    mxArray *a = make(mxDOUBLE_CLASS, 8, {1, 1});
    ((double *)a->data)[0] = v;
    return a;
}
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity) {
    const size_t es = cls == mxDOUBLE_CLASS ? 8 : cls == mxSINGLE_CLASS ? 4 : 1;
    return make(cls, es, std::vector<mwSize>(dims, dims + ndim));
}
void mxDestroyArray(mxArray *a) {
    if (!a) return;
    for (auto &kv : a->fields) mxDestroyArray(kv.second);
    free(a->data);
    delete a;
}
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw MexError{id, buf};
}
void mexLock(void) { ++g_locks; }
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }

// ---- driver used by the Python tests (ctypes) ----------------------------------------------------------------------
mxArray *mexrt_double(const double *src, int ndim, const size_t *dims) {     // column-major copy
    mxArray *a = make(mxDOUBLE_CLASS, 8, std::vector<mwSize>(dims, dims + ndim));
    if (src && numel(a)) memcpy(a->data, src, numel(a) * 8);
    return a;
}
mxArray *mexrt_logical(const unsigned char *src, size_t n) {
    mxArray *a = make(mxLOGICAL_CLASS, sizeof(mxLogical), {n, n ? (mwSize)1 : (mwSize)0});
    for (size_t q = 0; q < n; ++q) ((mxLogical *)a->data)[q] = src[q] != 0;
    return a;
}
mxArray *mexrt_string(const char *s) {
    mxArray *a = new mxArray_tag();
    a->cls = mxCHAR_CLASS;
    a->chars = s;
    return a;
}
mxArray *mexrt_struct(void) {
    mxArray *a = new mxArray_tag();
    a->cls = mxSTRUCT_CLASS;
    a->dims = {1, 1};
    return a;
}
void mexrt_set_field(mxArray *s, const char *name, mxArray *v) {             // the struct owns v
    auto it = s->fields.find(name);
    if (it != s->fields.end()) mxDestroyArray(it->second);
    s->fields[name] = v;
}
int mexrt_ndim(const mxArray *a) { return (int)a->dims.size(); }
size_t mexrt_dim(const mxArray *a, int q) { return a->dims[q]; }
int mexrt_call(int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {   // 0, or 1 with mexrt_last_error / _id set
    try {
        mexFunction(nlhs, plhs, nrhs, prhs);
        return 0;
    } catch (const MexError &e) {
        g_last_id = e.id;
        g_last_msg = e.msg;
        return 1;
    }
}
const char *mexrt_last_error(void) { return g_last_msg.c_str(); }
const char *mexrt_last_id(void) { return g_last_id.c_str(); }
int mexrt_locks(void) { return g_locks; }
void mexrt_unload(void) {                                                     // what "clear mex" does
    if (g_atexit) g_atexit();
    g_atexit = nullptr;
}
}

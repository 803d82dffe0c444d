// mex_mock.cpp -- TEST INFRASTRUCTURE: a small in-process stand-in for the part of the MATLAB runtime's C API that
// qmri_pnp_recon_poc_amd/mex/qmri_mex.cpp uses (tests/stubs/mex.h declares it from MathWorks' published documentation).  It lets the
// gateway -- our own code, unchanged -- be compiled, LINKED against libqmri.so and driven command by command from Python (ctypes:
// tests/mexmock.py): the exact call sequence a MATLAB session would make, with MATLAB's memory layouts (column-major, interleaved complex).
// It is not MathWorks' runtime: no workspace, no garbage collection of temporaries (arrays live until mxDestroyArray), errors are C++
// exceptions caught at the mock's entry point.
#include "mex.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    bool cplx = false;
    std::vector<mwSize> dims;
    void* data = nullptr;
    size_t bytes = 0;
    std::vector<std::string> fields;            // struct arrays (1 x 1 here): names and values
    std::vector<mxArray*> values;
    bool is_struct = false;
};

namespace {
struct MexError { std::string id, msg; };
std::string g_err_id, g_err_msg;
void (*g_atexit)(void) = nullptr;
bool g_locked = false;

size_t elem_size(mxClassID c) {
    switch (c) { case mxDOUBLE_CLASS: return 8; case mxSINGLE_CLASS: return 4; case mxINT32_CLASS: return 4; case mxCHAR_CLASS: return 2; default: return 1; }
}
size_t numel(const mxArray* a) { size_t n = 1; for (mwSize d : a->dims) n *= d; return a->dims.empty() ? 0 : n; }
mxArray* make(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag) {
    mxArray* a = new mxArray_tag();
    a->cls = cls; a->cplx = (flag == mxCOMPLEX);
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->bytes = numel(a) * elem_size(cls) * (a->cplx ? 2 : 1);
    a->data = a->bytes ? std::calloc(a->bytes, 1) : nullptr;
    return a;
}
}  // namespace

extern "C" {

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char buf[2048];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    throw MexError{id ? id : "", buf};
}
int mexAtExit(void (*fn)(void)) { g_atexit = fn; return 0; }
void mexLock(void) { g_locked = true; }
void mexUnlock(void) { g_locked = false; }
bool mexIsLocked(void) { return g_locked; }
void mexMakeArrayPersistent(mxArray*) {}          // (nothing is collected here anyway)

mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag) { return make(ndim, dims, classid, flag); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag) { const mwSize d[2] = {m, n}; return make(2, d, classid, flag); }
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, flag); }
mxArray* mxCreateDoubleScalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); *(double*)a->data = v; return a; }
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** names) {
    const mwSize d[2] = {m, n};
    mxArray* a = make(2, d, mxUNKNOWN_CLASS, mxREAL);
    a->is_struct = true;
    for (int i = 0; i < nfields; ++i) { a->fields.push_back(names[i]); a->values.push_back(nullptr); }
    return a;
}
void mxDestroyArray(mxArray* a) {
    if (!a) return;
    for (mxArray* v : a->values) mxDestroyArray(v);
    std::free(a->data);
    delete a;
}
mxArray* mxDuplicateArray(const mxArray* in) {
    mxArray* a = new mxArray_tag();
    a->cls = in->cls; a->cplx = in->cplx; a->dims = in->dims; a->bytes = in->bytes; a->is_struct = in->is_struct; a->fields = in->fields;
    a->data = in->bytes ? std::malloc(in->bytes) : nullptr;
    if (in->bytes) std::memcpy(a->data, in->data, in->bytes);
    for (const mxArray* v : in->values) a->values.push_back(v ? mxDuplicateArray(v) : nullptr);
    return a;
}
mxComplexDouble* mxGetComplexDoubles(const mxArray* a) { return (a->cls == mxDOUBLE_CLASS && a->cplx) ? (mxComplexDouble*)a->data : nullptr; }
mxDouble* mxGetDoubles(const mxArray* a) { return (a->cls == mxDOUBLE_CLASS && !a->cplx) ? (mxDouble*)a->data : nullptr; }
void* mxGetData(const mxArray* a) { return a->data; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims.data(); }
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->dims.size(); }
size_t mxGetNumberOfElements(const mxArray* a) { return numel(a); }
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) { size_t n = 1; for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i]; return n; }
void mxSetM(mxArray* a, mwSize m) { a->dims[0] = m; }                   // (shrinks the view; the allocation stays)
mxArray* mxGetField(const mxArray* a, mwIndex, const char* name) {
    if (!a->is_struct) return nullptr;
    for (size_t i = 0; i < a->fields.size(); ++i) if (a->fields[i] == name) return a->values[i];
    return nullptr;
}
void mxSetFieldByNumber(mxArray* a, mwIndex, int k, mxArray* v) { if (a->is_struct && k >= 0 && (size_t)k < a->values.size()) { mxDestroyArray(a->values[k]); a->values[k] = v; } }
double mxGetScalar(const mxArray* a) {
    if (!a->data || !numel(a)) return 0.0;
    switch (a->cls) {
        case mxDOUBLE_CLASS: return *(const double*)a->data;
        case mxSINGLE_CLASS: return *(const float*)a->data;
        case mxINT32_CLASS: return *(const int32_t*)a->data;
        case mxCHAR_CLASS: return *(const uint16_t*)a->data;
        default: return 0.0;
    }
}
int mxGetString(const mxArray* a, char* str, mwSize len) {
    if (a->cls != mxCHAR_CLASS || len == 0) return 1;
    const size_t n = numel(a);
    const uint16_t* c = (const uint16_t*)a->data;
    size_t i = 0;
    for (; i < n && i + 1 < len; ++i) str[i] = (char)c[i];
    str[i] = 0;
    return n + 1 > len ? 1 : 0;
}
bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
bool mxIsComplex(const mxArray* a) { return a->cplx; }
bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsSingle(const mxArray* a) { return a->cls == mxSINGLE_CLASS; }
bool mxIsInt32(const mxArray* a) { return a->cls == mxINT32_CLASS; }
bool mxIsStruct(const mxArray* a) { return a->is_struct; }
bool mxIsEmpty(const mxArray* a) { return numel(a) == 0; }

// ---- the mock's own entry points (tests/mexmock.py) ----------------------------------------------------------------------------------------
mxArray* mock_string(const char* s) {
    const mwSize d[2] = {1, (mwSize)std::strlen(s)};
    mxArray* a = make(2, d, mxCHAR_CLASS, mxREAL);
    for (size_t i = 0; i < d[1]; ++i) ((uint16_t*)a->data)[i] = (unsigned char)s[i];
    return a;
}
mxArray* mock_struct(int n, const char** names, const double* values) {     // 1 x 1 struct of double scalars
    mxArray* a = mxCreateStructMatrix(1, 1, n, names);
    for (int i = 0; i < n; ++i) mxSetFieldByNumber(a, 0, i, mxCreateDoubleScalar(values[i]));
    return a;
}
int mock_class(const mxArray* a) { return (int)a->cls; }
int mock_is_struct(const mxArray* a) { return a->is_struct ? 1 : 0; }
int mock_nfields(const mxArray* a) { return (int)a->fields.size(); }
const char* mock_field_name(const mxArray* a, int k) { return a->fields[k].c_str(); }
mxArray* mock_field_value(const mxArray* a, int k) { return a->values[k]; }
// what MATLAB's `out = qmri_mex(...)` does around mexFunction: 0 = returned, 1 = an error was raised (id / message below)
int mock_call(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {
    g_err_id.clear(); g_err_msg.clear();
    for (int i = 0; i < (nlhs > 0 ? nlhs : 1); ++i) plhs[i] = nullptr;
    try {
        mexFunction(nlhs, plhs, nrhs, prhs);
    } catch (const MexError& e) {
        g_err_id = e.id; g_err_msg = e.msg;
        return 1;
    }
    return 0;
}
const char* mock_error_id(void) { return g_err_id.c_str(); }
const char* mock_error_msg(void) { return g_err_msg.c_str(); }
void mock_exit(void) { if (g_atexit) g_atexit(); }                      // MATLAB quitting / `clear mex`

}  // extern "C"

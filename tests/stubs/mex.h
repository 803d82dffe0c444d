/*
 * mex.h -- STUB, test infrastructure only.  NOT MathWorks' header and not part of any build that produces a MEX file.
 *
 * The build image has no MATLAB (no mex.h), so qmri_pnp_recon_poc_amd/mex/qmri_mex.cpp could never go through a compiler.
 * This file declares -- from MathWorks' published C Matrix API / MEX API documentation (R2018a interleaved-complex API) -- the
 * handful of types and functions the shim uses, so that (1) tests/test_host_logic.py::test_mex_shim_compiles_against_stub_header can
 * run `g++ -fsyntax-only` on it, and (2) tests/cpp/mex_mock.cpp can IMPLEMENT them as a small in-process stand-in for the MATLAB runtime,
 * under which the gateway is compiled, linked against libqmri.so and driven command by command (tests/test_mex_mock.py on the CPU,
 * tests/test_gpu_mex.py on the GPU box).  Test infrastructure for OUR gateway only; a real build uses MATLAB's own mex.h (`mex -R2018a`).
 */
#ifndef QMRI_TEST_STUB_MEX_H
#define QMRI_TEST_STUB_MEX_H
#include <stddef.h>
#include <stdint.h>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef struct { double real, imag; } mxComplexDouble;
typedef double mxDouble;
typedef enum { mxUNKNOWN_CLASS = 0, mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

#ifdef __cplusplus
extern "C" {
#endif
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
void mexLock(void);
void mexUnlock(void);
bool mexIsLocked(void);

mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray* mxCreateDoubleScalar(double value);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID classid, mxComplexity flag);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID classid, mxComplexity flag);
mxArray* mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char** fieldnames);
void mxDestroyArray(mxArray* pm);
mxArray* mxDuplicateArray(const mxArray* in);
void mexMakeArrayPersistent(mxArray* pm);
mxComplexDouble* mxGetComplexDoubles(const mxArray* pa);
mxDouble* mxGetDoubles(const mxArray* pa);
void* mxGetData(const mxArray* pm);
const mwSize* mxGetDimensions(const mxArray* pm);
mxArray* mxGetField(const mxArray* pm, mwIndex index, const char* fieldname);
size_t mxGetM(const mxArray* pm);
size_t mxGetN(const mxArray* pm);
mwSize mxGetNumberOfDimensions(const mxArray* pm);
size_t mxGetNumberOfElements(const mxArray* pm);
double mxGetScalar(const mxArray* pm);
int mxGetString(const mxArray* pm, char* str, mwSize strlen);
bool mxIsChar(const mxArray* pm);
bool mxIsComplex(const mxArray* pm);
bool mxIsDouble(const mxArray* pm);
bool mxIsSingle(const mxArray* pm);
bool mxIsInt32(const mxArray* pm);
bool mxIsStruct(const mxArray* pm);
bool mxIsEmpty(const mxArray* pm);
void mxSetFieldByNumber(mxArray* pm, mwIndex index, int fieldnumber, mxArray* pvalue);
void mxSetM(mxArray* pm, mwSize m);
#ifdef __cplusplus
}
#endif
#endif

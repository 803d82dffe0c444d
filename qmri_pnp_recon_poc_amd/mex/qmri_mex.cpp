// qmri_mex.cpp -- MATLAB gateway for libqmri.so (compile-gated: needs mex.h, which is absent from the build image).
//
//   mex -R2018a qmri_mex.cpp -I../../include -L.. -lqmri        (interleaved-complex API: mxComplexDouble == (re,im) doubles)
//
// One mexFunction with a leading command string; a persistent context is created on first use and released by
// mexAtExit.  Every libqmri status != 0 becomes mexErrMsgIdAndTxt('qmri:<code>', qmri_last_error(ctx)), which is how
// the reference's plugins report errors (MATLAB exceptions, denoiseImage_PnP_ADMM.m:123-135).
// The MATLAB wrappers in ../matlab give these commands the reference's own signatures.
//
// Batches and devices (round 5).  The reference's denoiser handle takes H x W x C x N batches (denoiseImage_PnP_ADMM.m:13-17) and north_star
// shards a slice batch over the GPUs of a node.  The gateway therefore
//   * keeps persistent copies of what defines the plans (V / frame_ptr / kidx, the denoiser's weights and shape, the dictionary), so that
//   * a call that brings more slices than the current plan holds re-plans by itself: 'denoise' with a 4-D array of N > max_batch slices,
//     'pnp_admm' with a measurement MATRIX (m x S) -- the plan then grows to min(S, 15) slices per launch;
//   * 'device' selects the GPU of the single-context commands;
//   * 'recon_batch' hands a whole slice stack to qmri_recon_batch: one worker (host thread + context) per entry of `devs`, slices_per_launch
//     slices advanced together on each (k_conv6p, batched LSQR), x and the T1 / T2 / PD maps of every slice back.
// tests/cpp/mex_mock.cpp is a small stand-in for the MATLAB runtime's C API under which this file is compiled, LINKED against libqmri.so and
// driven command by command on the GPU box (tests/test_gpu_mex.py); with MATLAB's own mex.h nothing here changes.
#include "mex.h"
#include "qmri.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static qmri_ctx* g_ctx = nullptr;
static int g_device = 0;

// what the current plans were made from (persistent mxArrays: they survive the call that brought them)
struct OperatorSpec { mxArray* V = nullptr; mxArray* fp = nullptr; mxArray* kidx = nullptr; int N = 0, M = 0, max_batch = 0; };
struct DenoiserSpec { mxArray* w = nullptr; qmri_net_desc d{}; int H = 0, W = 0, max_batch = 0; };
struct DictSpec { mxArray* D = nullptr; mxArray* normD = nullptr; mxArray* lut = nullptr; };
static OperatorSpec g_op;
static DenoiserSpec g_net;
static DictSpec g_dict;
static const int DEFAULT_SLICES_PER_LAUNCH = 15;       // what a measurement matrix grows the plans to (the batched kernels' design point)

static void drop(mxArray*& a) { if (a) { mxDestroyArray(a); a = nullptr; } }
static mxArray* keep(const mxArray* a) { mxArray* c = mxDuplicateArray(a); mexMakeArrayPersistent(c); return c; }

static void cleanup() {
    if (g_ctx) { qmri_destroy(g_ctx); g_ctx = nullptr; }
    drop(g_op.V); drop(g_op.fp); drop(g_op.kidx); g_op = OperatorSpec();
    drop(g_net.w); g_net = DenoiserSpec();
    drop(g_dict.D); drop(g_dict.normD); drop(g_dict.lut);
}

static void check(int st) {
    if (st == QMRI_OK) return;
    char id[32];
    snprintf(id, sizeof id, "qmri:err%d", -st);
    mexErrMsgIdAndTxt(id, "%s", qmri_last_error(g_ctx));
}

static qmri_ctx* ctx() {
    if (!g_ctx) {
        int st = qmri_create(g_device, &g_ctx);
        if (st != QMRI_OK) mexErrMsgIdAndTxt("qmri:create", "%s", qmri_last_error(nullptr));
    }
    static bool registered = false;
    if (!registered) { mexAtExit(cleanup); mexLock(); registered = true; }
    return g_ctx;
}

static double scalar_field(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    return f ? mxGetScalar(f) : dflt;
}

static void need(int nrhs, int n, const char* usage) {
    if (nrhs < n) mexErrMsgIdAndTxt("qmri:usage", "%s", usage);
}
// The C ABI takes plain pointers: what a MATLAB array must be and hold is checked HERE, before the library reads it (a wrong class or a short
// array would otherwise be read past its end).  Errors are MATLAB exceptions with an identifier, like the reference's validateInputImage.
static void want(bool ok, const char* id, const char* msg) {
    if (!ok) mexErrMsgIdAndTxt(id, "%s", msg);
}
static bool is_cdouble(const mxArray* a) { return mxIsDouble(a) && mxIsComplex(a); }
// a MATLAB scalar that is about to become an int / size_t: real, finite, integer-valued and inside [lo, hi] BEFORE the cast (a NaN or a negative
// double cast to an integer type is undefined behaviour)
static int int_arg(const mxArray* a, double lo, double hi, const char* id, const char* msg) {
    want(a && mxIsDouble(a) && !mxIsComplex(a) && mxGetNumberOfElements(a) == 1, id, msg);
    const double v = mxGetScalar(a);
    want(std::isfinite(v) && v == std::floor(v) && v >= lo && v <= hi, id, msg);
    return (int)v;
}
static size_t image_numel() {                                       // N * M * s of the planned operator
    want(g_op.V != nullptr, "qmri:state", "no operator: call qmri_mex('set_operator', ...) (qmri_make_F) first");
    return (size_t)g_op.N * (size_t)g_op.M * mxGetN(g_op.V);
}
static size_t dims_numel(const mxArray* d) {                        // the [N M s] argument
    want(mxIsDouble(d) && !mxIsComplex(d) && mxGetNumberOfElements(d) == 3, "qmri:size", "the size argument must be [N M s]");
    const double* v = mxGetDoubles(d);
    for (int i = 0; i < 3; ++i) want(std::isfinite(v[i]) && v[i] == std::floor(v[i]) && v[i] >= 1 && v[i] <= 1e6, "qmri:size", "[N M s] must hold positive integers");
    const size_t n = (size_t)v[0] * (size_t)v[1] * (size_t)v[2];
    want(n == image_numel(), "qmri:size", "[N M s] does not match the operator (N x M grid, s = columns of V)");
    return n;
}
static size_t operator_m() {
    int m = 0;
    check(qmri_operator_m(ctx(), &m));
    return (size_t)m;
}

// (re-)make the plans from the kept specifications
static void plan_operator(int max_batch) {
    const int T = (int)mxGetM(g_op.V), s = (int)mxGetN(g_op.V);
    check(qmri_set_operator(ctx(), g_op.N, g_op.M, s, T, mxGetDoubles(g_op.V), (const int32_t*)mxGetData(g_op.fp), (const int32_t*)mxGetData(g_op.kidx), max_batch));
    g_op.max_batch = max_batch;
}
static void plan_denoiser(int max_batch) {
    check(qmri_set_denoiser(ctx(), &g_net.d, (const float*)mxGetData(g_net.w), mxGetNumberOfElements(g_net.w) * 4, g_net.H, g_net.W, max_batch));
    g_net.max_batch = max_batch;
}
static void plan_dictionary() {
    check(qmri_set_dictionary(ctx(), (int)mxGetM(g_dict.D), (int)mxGetN(g_dict.D), (int)mxGetN(g_dict.lut), (const float*)mxGetData(g_dict.D),
                              (const float*)mxGetData(g_dict.normD), (const float*)mxGetData(g_dict.lut)));
}
// a call brings B slices: grow the plans that hold fewer
static void reserve(int B, bool op, bool net) {
    if (op && g_op.V && B > g_op.max_batch) plan_operator(B);
    if (net && g_net.w && B > g_net.max_batch) plan_denoiser(B);
}

static qmri_admm_params admm_params(const mxArray* P, bool want_diag) {
    qmri_admm_params p;
    p.gamma = scalar_field(P, "gamma", 0.05);
    p.iters = (int)scalar_field(P, "iter", 100);
    p.cg_tol = scalar_field(P, "cg_tol", 1e-4);
    p.cg_maxit = 100;                                               // literal in PnP_ADMM.m:102
    p.solver = (int)scalar_field(P, "solver", QMRI_SOLVER_LSQR);
    p.denoiser_type = (int)scalar_field(P, "multi_level", 0);
    p.noise_std = scalar_field(P, "noise_std", 0.01);
    p.want_diag = want_diag ? 1 : 0;
    return p;
}

static void set_denoiser_from(const mxArray* w, const qmri_net_desc& d, int H, int W, int max_batch) {
    if (!mxIsSingle(w) || mxIsComplex(w)) mexErrMsgIdAndTxt("qmri:set_denoiser:type", "the weights must be a real single vector");
    drop(g_net.w);
    g_net.w = keep(w); g_net.d = d; g_net.H = H; g_net.W = W;
    plan_denoiser(std::max(1, max_batch));
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("qmri:usage", "first argument must be a command string");
    char cmd[64];
    mxGetString(prhs[0], cmd, sizeof cmd);
    const std::string c(cmd);

    if (c == "device") {                             // qmri_mex('device', d): the GPU of the single-context commands (plans are re-made on it)
        need(nrhs, 2, "qmri_mex('device', d)");
        const int d = int_arg(prhs[1], 0, 1023, "qmri:device", "the device id must be a non-negative integer");
        if (d != g_device || !g_ctx) {
            // try the new device first: a failed qmri_create must leave the gateway on the device it had (with its context and plans)
            qmri_ctx* fresh = nullptr;
            if (qmri_create(d, &fresh) != QMRI_OK) mexErrMsgIdAndTxt("qmri:create", "%s", qmri_last_error(nullptr));
            if (g_ctx) qmri_destroy(g_ctx);
            g_ctx = fresh; g_device = d;
            (void)ctx();                                            // (registers the exit hook and the lock on first use)
            if (g_op.V) plan_operator(std::max(1, g_op.max_batch));
            if (g_net.w) plan_denoiser(std::max(1, g_net.max_batch));
            if (g_dict.D) plan_dictionary();
        }
        if (nlhs > 0) plhs[0] = mxCreateDoubleScalar((double)g_device);
    } else if (c == "set_operator") {                // qmri_mex('set_operator', N, M, V, frame_ptr(int32), kidx(int32) [, max_batch])
        need(nrhs, 6, "qmri_mex('set_operator', N, M, V, frame_ptr, kidx [, max_batch])");
        if (!mxIsDouble(prhs[3]) || mxIsComplex(prhs[3])) mexErrMsgIdAndTxt("qmri:set_operator:type", "V must be a real double T x s matrix");
        want(mxIsInt32(prhs[4]) && mxIsInt32(prhs[5]), "qmri:set_operator:type", "frame_ptr and kidx must be int32 (as 'build_spiral' / 'build_epi' return them)");
        {
            const size_t T = mxGetM(prhs[3]);
            want(mxGetNumberOfElements(prhs[4]) == T + 1, "qmri:set_operator:size", "frame_ptr must have T + 1 entries (T = rows of V)");
            const int32_t total = ((const int32_t*)mxGetData(prhs[4]))[T];
            want(total >= 0 && mxGetNumberOfElements(prhs[5]) == (size_t)total, "qmri:set_operator:size", "kidx must have frame_ptr(end) entries");
        }
        const int Nn = int_arg(prhs[1], 1, 65536, "qmri:set_operator:size", "N must be a positive integer");
        const int Mm = int_arg(prhs[2], 1, 65536, "qmri:set_operator:size", "M must be a positive integer");
        const int mb = nrhs > 6 ? int_arg(prhs[6], 1, 4096, "qmri:set_operator:size", "max_batch must be a positive integer") : 1;
        drop(g_op.V); drop(g_op.fp); drop(g_op.kidx);
        g_op.N = Nn; g_op.M = Mm;
        g_op.V = keep(prhs[3]); g_op.fp = keep(prhs[4]); g_op.kidx = keep(prhs[5]);
        plan_operator(mb);
    } else if (c == "build_spiral" || c == "build_epi") {   // [frame_ptr, kidx] = qmri_mex('build_spiral', N, S, T)   (host integer code: no GPU needed)
        need(nrhs, 4, "[frame_ptr, kidx] = qmri_mex('build_spiral', N, S, T) | qmri_mex('build_epi', N, M, pct, T)");
        const int N = (int)mxGetScalar(prhs[1]);
        int m = 0, st;
        if (c == "build_spiral") {
            const int S = (int)mxGetScalar(prhs[2]), T = (int)mxGetScalar(prhs[3]);
            plhs[0] = mxCreateNumericMatrix(T + 1, 1, mxINT32_CLASS, mxREAL);
            mxArray* k = mxCreateNumericMatrix((size_t)S * T, 1, mxINT32_CLASS, mxREAL);
            st = qmri_build_spiral(nullptr, N, S, T, (int32_t*)mxGetData(plhs[0]), (int32_t*)mxGetData(k), S * T, &m);
            mxSetM(k, m);
            plhs[1] = k;
        } else {
            need(nrhs, 5, "[frame_ptr, kidx] = qmri_mex('build_epi', N, M, pct, T)");
            const int M = (int)mxGetScalar(prhs[2]);
            const double pct = mxGetScalar(prhs[3]);
            const int T = (int)mxGetScalar(prhs[4]);
            const int cap = N * M * T;
            plhs[0] = mxCreateNumericMatrix(T + 1, 1, mxINT32_CLASS, mxREAL);
            mxArray* k = mxCreateNumericMatrix((size_t)cap, 1, mxINT32_CLASS, mxREAL);
            st = qmri_build_epi(nullptr, N, M, pct, T, (int32_t*)mxGetData(plhs[0]), (int32_t*)mxGetData(k), cap, &m);
            mxSetM(k, m);
            plhs[1] = k;
        }
        if (st != QMRI_OK) mexErrMsgIdAndTxt("qmri:mask", "%s", qmri_last_error(nullptr));
    } else if (c == "forward") {                     // y = qmri_mex('forward', x)   (F.forward, main_recon_tsmis_FFT.m:228)
        need(nrhs, 2, "y = qmri_mex('forward', x)");
        want(mxIsDouble(prhs[1]) && mxGetNumberOfElements(prhs[1]) == image_numel(), "qmri:forward:size", "x must be a double N x M x s array");
        plhs[0] = mxCreateDoubleMatrix(operator_m(), 1, mxCOMPLEX);
        const bool cx = mxIsComplex(prhs[1]);
        check(qmri_forward(ctx(), cx ? (const void*)mxGetComplexDoubles(prhs[1]) : (const void*)mxGetDoubles(prhs[1]), cx,
                           mxGetComplexDoubles(plhs[0])));
    } else if (c == "adjoint") {                     // x = qmri_mex('adjoint', y, [N M s])   (F.adjoint, :229)
        need(nrhs, 3, "x = qmri_mex('adjoint', y, [N M s])");
        (void)dims_numel(prhs[2]);
        want(is_cdouble(prhs[1]) && mxGetNumberOfElements(prhs[1]) == operator_m(), "qmri:adjoint:size", "y must be a complex double vector with one entry per sample");
        const double* d = mxGetDoubles(prhs[2]);
        const mwSize dims[3] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2]};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        check(qmri_adjoint(ctx(), mxGetComplexDoubles(prhs[1]), mxGetComplexDoubles(plhs[0])));
    } else if (c == "set_denoiser") {                // qmri_mex('set_denoiser', weights(single), in_nc, out_nc, nc(1x4), nb, residual_noise, H, W [, max_batch])
        need(nrhs, 9, "qmri_mex('set_denoiser', weights, in_nc, out_nc, nc, nb, residual_noise, H, W [, max_batch])");
        qmri_net_desc d;
        d.arch = QMRI_ARCH_UNETRES;
        d.in_nc = (int)mxGetScalar(prhs[2]); d.out_nc = (int)mxGetScalar(prhs[3]);
        want(mxIsDouble(prhs[4]) && mxGetNumberOfElements(prhs[4]) == 4, "qmri:set_denoiser:size", "nc must hold the four channel counts");
        const double* nc = mxGetDoubles(prhs[4]);
        for (int i = 0; i < 4; ++i) d.nc[i] = (int)nc[i];
        d.nb = (int)mxGetScalar(prhs[5]); d.residual_noise = (int)mxGetScalar(prhs[6]);
        set_denoiser_from(prhs[1], d, (int)mxGetScalar(prhs[7]), (int)mxGetScalar(prhs[8]), nrhs > 9 ? (int)mxGetScalar(prhs[9]) : 1);
    } else if (c == "load_onnx") {                   // [in_nc, out_nc] = qmri_mex('load_onnx', denoiser_path, residual_noise, H, W [, max_batch])
        // the weight-loading half of `Net = importONNXNetwork(denoiser_path, ...)` (main_recon_tsmis_FFT.m:138) + set_denoiser
        need(nrhs, 5, "[in_nc, out_nc] = qmri_mex('load_onnx', denoiser_path, residual_noise, H, W [, max_batch])");
        char path[4096];
        if (mxGetString(prhs[1], path, sizeof path)) mexErrMsgIdAndTxt("qmri:usage", "denoiser_path must be a char vector");
        qmri_net_desc d;
        size_t n = 0;
        if (qmri_onnx_read_unetres(path, &d, nullptr, 0, &n) != QMRI_OK) mexErrMsgIdAndTxt("qmri:onnx", "%s", qmri_last_error(nullptr));
        mxArray* w = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        if (qmri_onnx_read_unetres(path, &d, (float*)mxGetData(w), n, &n) != QMRI_OK) mexErrMsgIdAndTxt("qmri:onnx", "%s", qmri_last_error(nullptr));
        d.residual_noise = (int)mxGetScalar(prhs[2]);
        set_denoiser_from(w, d, (int)mxGetScalar(prhs[3]), (int)mxGetScalar(prhs[4]), nrhs > 5 ? (int)mxGetScalar(prhs[5]) : 1);
        mxDestroyArray(w);
        plhs[0] = mxCreateDoubleScalar((double)d.in_nc);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)d.out_nc);
    } else if (c == "denoise") {                     // I = qmri_mex('denoise', A, out_nc)   (param.net, :164): A is H x W x C or H x W x C x N (denoiseImage_PnP_ADMM.m:13-17)
        need(nrhs, 3, "I = qmri_mex('denoise', A, out_nc)");
        const mwSize* dm = mxGetDimensions(prhs[1]);
        const int nd = (int)mxGetNumberOfDimensions(prhs[1]);
        if (mxIsComplex(prhs[1]) || !mxIsDouble(prhs[1]) || nd > 4)
            mexErrMsgIdAndTxt("images:denoiseImage:invalidImageFormat", "A must be a real double H x W x C (x N) array");
        const int H = (int)dm[0], W = (int)dm[1], C = nd > 2 ? (int)dm[2] : 1, B = nd > 3 ? (int)dm[3] : 1;
        // the library writes H * W * (the PLAN's out_nc) * B doubles: the output array is sized from the plan, and what the caller says is checked
        // against it here (an out_nc below the plan's would otherwise be a write past the end of a MATLAB array)
        want(g_net.w != nullptr, "qmri:state", "no denoiser: call qmri_mex('set_denoiser' | 'load_onnx', ...) (qmri_make_net) first");
        want(H == g_net.H && W == g_net.W && C == g_net.d.in_nc, "qmri:denoise:size", "A must be H x W x in_nc (x N) as given to set_denoiser");
        want(int_arg(prhs[2], 1, 1 << 20, "qmri:denoise:size", "out_nc must be a positive integer") == g_net.d.out_nc, "qmri:denoise:size",
             "out_nc does not match the denoiser's output channels");
        want(B >= 1, "qmri:denoise:size", "A holds no slice");
        reserve(B, false, true);                                    // a batch larger than the plan: re-planned, as the reference's handle takes any N
        const mwSize od[4] = {(mwSize)H, (mwSize)W, (mwSize)g_net.d.out_nc, (mwSize)B};
        plhs[0] = mxCreateNumericArray(B > 1 ? 4 : 3, od, mxDOUBLE_CLASS, mxREAL);
        check(qmri_denoise(ctx(), mxGetDoubles(prhs[1]), H, W, C, B, mxGetDoubles(plhs[0])));
    } else if (c == "pnp_admm") {                    // [x, diag, lsqr_iters] = qmri_mex('pnp_admm', y, param_struct, X0, gt, [N M s])
        // y: m x 1 -> x is N x M x s;  y: m x S (a slice stack) -> x is N x M x s x S, diag 2 x iter x S, lsqr_iters iter x S: the slices advance together
        // through the batched kernels, slices_per_launch = min(S, 15) at a time, on the current device (X0 / gt: N x M x s x S or empty)
        need(nrhs, 6, "[x, diag, lsqr_iters] = qmri_mex('pnp_admm', y, param, X0, gt, [N M s])");
        want(mxIsStruct(prhs[2]), "qmri:pnp_admm:type", "param must be a struct");
        const qmri_admm_params p = admm_params(prhs[2], nlhs > 1);
        const double* d = mxGetDoubles(prhs[5]);
        const size_t S = mxGetN(prhs[1]), m = mxGetM(prhs[1]), n = dims_numel(prhs[5]);
        want(is_cdouble(prhs[1]) && S >= 1 && m == operator_m(), "qmri:pnp_admm:size", "y must be complex double, one column of m samples per slice");
        want(g_net.w != nullptr, "qmri:state", "no denoiser: call qmri_mex('set_denoiser' | 'load_onnx', ...) (qmri_make_net) first");
        want((mxIsEmpty(prhs[3]) || is_cdouble(prhs[3])) && (mxIsEmpty(prhs[4]) || is_cdouble(prhs[4])), "qmri:pnp_admm:type", "X0 and gt_tsmi must be complex double or empty");
        const int it = p.iters > 0 ? p.iters : 1;
        const mwSize dims[4] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2], (mwSize)S};
        plhs[0] = mxCreateNumericArray(S > 1 ? 4 : 3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        const mwSize ddims[3] = {2, (mwSize)it, (mwSize)S};
        mxArray* diag = mxCreateNumericArray(S > 1 ? 3 : 2, ddims, mxDOUBLE_CLASS, mxREAL);
        mxArray* li = mxCreateNumericMatrix(it, S, mxINT32_CLASS, mxREAL);
        const mxComplexDouble* y = mxGetComplexDoubles(prhs[1]);
        const mxComplexDouble* x0 = mxIsEmpty(prhs[3]) ? nullptr : mxGetComplexDoubles(prhs[3]);
        const mxComplexDouble* gt = mxIsEmpty(prhs[4]) ? nullptr : mxGetComplexDoubles(prhs[4]);
        if (x0 && mxGetNumberOfElements(prhs[3]) != n * S) mexErrMsgIdAndTxt("qmri:pnp_admm:size", "X0 must hold N x M x s values per slice");
        if (gt && mxGetNumberOfElements(prhs[4]) != n * S) mexErrMsgIdAndTxt("qmri:pnp_admm:size", "gt_tsmi must hold N x M x s values per slice");
        if (S == 1) {
            check(qmri_pnp_admm(ctx(), y, &p, x0, gt, mxGetComplexDoubles(plhs[0]), p.want_diag ? mxGetDoubles(diag) : nullptr, (int32_t*)mxGetData(li)));
        } else {
            // several slices on this device: the plans grow to the launch size, then qmri_pnp_admm_batch walks the stack
            const int spl = (int)std::min<size_t>(S, DEFAULT_SLICES_PER_LAUNCH);
            reserve(spl, true, true);
            check(qmri_pnp_admm_batch(ctx(), (int)S, spl, y, &p, x0, gt, mxGetComplexDoubles(plhs[0]), p.want_diag ? mxGetDoubles(diag) : nullptr,
                                      (int32_t*)mxGetData(li)));
        }
        if (nlhs > 1) plhs[1] = diag; else mxDestroyArray(diag);
        if (nlhs > 2) plhs[2] = li; else mxDestroyArray(li);
    } else if (c == "recon_batch") {                 // [X, qmap, pd] = qmri_mex('recon_batch', Y(m x S), param_struct, devs, slices_per_launch, [N M s])
        // north_star's batch path: S independent slices sharded over the GPUs in `devs` (one worker = host thread + context per entry; an id may
        // repeat), slices_per_launch advanced together on each; 100 PnP-ADMM iterations + dictionary match per slice (the match only when a dictionary
        // is set and maps are asked for).  Uses the operator / denoiser / dictionary given to 'set_operator' / 'set_denoiser' / 'set_dictionary'.
        need(nrhs, 6, "[X, qmap, pd] = qmri_mex('recon_batch', Y, param, devs, slices_per_launch, [N M s])");
        if (!g_op.V || !g_net.w) mexErrMsgIdAndTxt("qmri:recon_batch:state", "set_operator and set_denoiser (or load_onnx) must come first");
        const size_t S = mxGetN(prhs[1]);
        (void)dims_numel(prhs[5]);
        want(mxIsStruct(prhs[2]), "qmri:recon_batch:type", "param must be a struct");
        want(is_cdouble(prhs[1]) && S >= 1 && mxGetM(prhs[1]) == operator_m(), "qmri:recon_batch:size", "Y must be complex double, one column of m samples per slice");
        want(mxIsDouble(prhs[3]) && !mxIsComplex(prhs[3]), "qmri:recon_batch:devs", "devs must be a double vector of device ids");
        const double* d = mxGetDoubles(prhs[5]);
        std::vector<int> devs(mxGetNumberOfElements(prhs[3]));
        for (size_t i = 0; i < devs.size(); ++i) devs[i] = (int)mxGetDoubles(prhs[3])[i];
        if (devs.empty()) mexErrMsgIdAndTxt("qmri:recon_batch:devs", "devs must name at least one device");
        const bool maps = nlhs > 1 && g_dict.D;
        qmri_problem pb;
        std::memset(&pb, 0, sizeof pb);
        pb.N = g_op.N; pb.M = g_op.M; pb.T = (int)mxGetM(g_op.V); pb.s = (int)mxGetN(g_op.V);
        pb.V = mxGetDoubles(g_op.V); pb.frame_ptr = (const int32_t*)mxGetData(g_op.fp); pb.kidx = (const int32_t*)mxGetData(g_op.kidx);
        pb.net = &g_net.d; pb.weights = (const float*)mxGetData(g_net.w); pb.weights_nbytes = mxGetNumberOfElements(g_net.w) * 4;
        if (maps) {
            pb.K = (int)mxGetM(g_dict.D); pb.Q = (int)mxGetN(g_dict.lut);
            pb.D = (const float*)mxGetData(g_dict.D); pb.normD = (const float*)mxGetData(g_dict.normD); pb.lut = (const float*)mxGetData(g_dict.lut);
        }
        pb.admm = admm_params(prhs[2], false);
        pb.slices_per_launch = std::max(1, (int)mxGetScalar(prhs[4]));
        const mwSize xd[4] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2], (mwSize)S};
        plhs[0] = mxCreateNumericArray(4, xd, mxDOUBLE_CLASS, mxCOMPLEX);
        mxArray *qm = nullptr, *pd = nullptr;
        if (maps) {
            const mwSize qd[4] = {(mwSize)d[0], (mwSize)d[1], (mwSize)pb.Q, (mwSize)S};
            const mwSize pdd[3] = {(mwSize)d[0], (mwSize)d[1], (mwSize)S};
            qm = mxCreateNumericArray(4, qd, mxSINGLE_CLASS, mxREAL);
            pd = mxCreateNumericArray(3, pdd, mxSINGLE_CLASS, mxCOMPLEX);
        }
        char err[1024] = "";
        const int st = qmri_recon_batch((int)devs.size(), devs.data(), (int)S, &pb, mxGetComplexDoubles(prhs[1]), mxGetComplexDoubles(plhs[0]),
                                        qm ? (float*)mxGetData(qm) : nullptr, pd ? (float*)mxGetData(pd) : nullptr, err, sizeof err);
        if (st != QMRI_OK) {
            char id[32];
            snprintf(id, sizeof id, "qmri:err%d", -st);
            mexErrMsgIdAndTxt(id, "%s", err);
        }
        if (nlhs > 1) plhs[1] = qm ? qm : mxCreateNumericMatrix(0, 0, mxSINGLE_CLASS, mxREAL);
        if (nlhs > 2) plhs[2] = pd ? pd : mxCreateNumericMatrix(0, 0, mxSINGLE_CLASS, mxCOMPLEX); else if (pd) mxDestroyArray(pd);
    } else if (c == "lrtv") {                        // [x, info] = qmri_mex('lrtv', y, param_struct, [N M s])   (FISTA_deep, main_recon_tsmis_FFT.m:273-282)
        need(nrhs, 4, "[x, info] = qmri_mex('lrtv', y, param, [N M s])");
        const mxArray* P = prhs[2];
        want(mxIsStruct(P), "qmri:lrtv:type", "param must be a struct");
        (void)dims_numel(prhs[3]);
        want(is_cdouble(prhs[1]) && mxGetNumberOfElements(prhs[1]) == operator_m(), "qmri:lrtv:size", "y must be a complex double vector with one entry per sample");
        qmri_lrtv_params p;
        p.K = scalar_field(P, "K", 4e-5);
        p.iters = (int)scalar_field(P, "iter", 200);
        p.step = scalar_field(P, "step", 0.0);
        p.tol = scalar_field(P, "tol", 1e-4);
        p.backtrack = (int)scalar_field(P, "backtrack", 1);
        p.prox_tol = 0.0; p.prox_maxit = 0;                        // prox_tv defaults (prox_tv.m:99-101)
        const double* d = mxGetDoubles(prhs[3]);
        const mwSize dims[3] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2]};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        qmri_lrtv_info info;
        check(qmri_lrtv(ctx(), mxGetComplexDoubles(prhs[1]), &p, mxGetComplexDoubles(plhs[0]), &info));
        if (nlhs > 1) {
            const char* names[] = {"iters", "halvings", "step", "obj", "prox_calls", "prox_iters_total"};
            plhs[1] = mxCreateStructMatrix(1, 1, 6, names);
            const double v[6] = {(double)info.iters, (double)info.halvings, info.step, info.obj, (double)info.prox_calls, (double)info.prox_iters_total};
            for (int i = 0; i < 6; ++i) mxSetFieldByNumber(plhs[1], 0, i, mxCreateDoubleScalar(v[i]));
        }
    } else if (c == "set_dictionary") {              // qmri_mex('set_dictionary', D(single KxS), normD(single), lut(single KxQ))
        // a complex mxArray holds interleaved (re,im) pairs: reading it as K x s reals would match against garbage atoms.
        // mrf_dtm_hip.m passes real(D) after checking that imag(D) is zero; anything else is refused here.
        for (int a = 1; a <= 3; ++a)
            if (nrhs <= a || mxIsComplex(prhs[a]) || !mxIsSingle(prhs[a]))
                mexErrMsgIdAndTxt("qmri:set_dictionary:type", "D, normD and lut must be real single arrays (argument %d is not)", a);
        if (mxGetNumberOfElements(prhs[2]) != mxGetM(prhs[1]) || mxGetM(prhs[3]) != mxGetM(prhs[1]))
            mexErrMsgIdAndTxt("qmri:set_dictionary:size", "normD must have K elements and lut K rows (K = rows of D)");
        drop(g_dict.D); drop(g_dict.normD); drop(g_dict.lut);
        g_dict.D = keep(prhs[1]); g_dict.normD = keep(prhs[2]); g_dict.lut = keep(prhs[3]);
        plan_dictionary();
    } else if (c == "dict_match") {                  // [qmap, pd, mt, dm, xfit] = qmri_mex('dict_match', X(Npix x s complex double), Q)
        need(nrhs, 3, "[qmap, pd, mt, dm, xfit] = qmri_mex('dict_match', X, Q)");
        want(g_dict.D != nullptr, "qmri:state", "no dictionary: call qmri_mex('set_dictionary', ...) (mrf_dtm_hip(dict, [], [])) first");
        want(is_cdouble(prhs[1]) && mxGetN(prhs[1]) == mxGetN(g_dict.D), "qmri:dict_match:size", "X must be complex double, Npix x s (s = columns of dict.D)");
        want((size_t)mxGetScalar(prhs[2]) == mxGetN(g_dict.lut), "qmri:dict_match:size", "Q must be the number of columns of dict.lut");
        const int npix = (int)mxGetM(prhs[1]), Q = (int)mxGetScalar(prhs[2]);
        mxArray* xfit = (nlhs > 4) ? mxCreateNumericMatrix(npix, mxGetN(prhs[1]), mxSINGLE_CLASS, mxCOMPLEX) : nullptr;   // out.Xfit, mrf_dtm_cpu.m:129-134
        plhs[0] = mxCreateNumericMatrix(npix, Q, mxSINGLE_CLASS, mxREAL);
        mxArray* pd = mxCreateNumericMatrix(npix, 1, mxSINGLE_CLASS, mxCOMPLEX);
        mxArray* mt = mxCreateNumericMatrix(npix, 1, mxSINGLE_CLASS, mxREAL);
        mxArray* dm = mxCreateNumericMatrix(npix, 1, mxINT32_CLASS, mxREAL);
        check(qmri_dict_match_xfit(ctx(), mxGetComplexDoubles(prhs[1]), npix, (float*)mxGetData(plhs[0]), (float*)mxGetData(pd),
                                   (float*)mxGetData(mt), (int32_t*)mxGetData(dm), xfit ? (float*)mxGetData(xfit) : nullptr));
        if (nlhs > 1) plhs[1] = pd; else mxDestroyArray(pd);
        if (nlhs > 2) plhs[2] = mt; else mxDestroyArray(mt);
        if (nlhs > 3) plhs[3] = dm; else mxDestroyArray(dm);
        if (nlhs > 4) plhs[4] = xfit;
    } else if (c == "health") {                      // h = qmri_mex('health'): qmri_get_health of the gateway's context as a struct
        qmri_health h;
        check(qmri_get_health(ctx(), &h));
        const char* names[] = {"denoiser_scheme", "denoiser_fallbacks", "resident_armed", "resident_timeouts", "lsqr_one_launch", "lsqr_timeouts",
                               "repeated_calls", "last_call_wall_ms", "last_call_stage_ms", "set_denoiser_ms"};
        plhs[0] = mxCreateStructMatrix(1, 1, 10, names);
        const double v[8] = {(double)h.denoiser_scheme, (double)h.denoiser_fallbacks, (double)h.resident_armed, (double)h.resident_timeouts,
                             (double)h.lsqr_one_launch, (double)h.lsqr_timeouts, (double)h.repeated_calls, h.last_call_wall_ms};
        for (int i = 0; i < 8; ++i) mxSetFieldByNumber(plhs[0], 0, i, mxCreateDoubleScalar(v[i]));
        mxArray* st = mxCreateDoubleMatrix(1, 4, mxREAL);           // x-update, denoiser, elementwise, diagnostics
        for (int i = 0; i < 4; ++i) mxGetDoubles(st)[i] = h.last_call_stage_ms[i];
        mxSetFieldByNumber(plhs[0], 0, 8, st);
        mxArray* sd = mxCreateDoubleMatrix(1, 3, mxREAL);           // pack + upload, tensors, calibration probe
        for (int i = 0; i < 3; ++i) mxGetDoubles(sd)[i] = h.set_denoiser_ms[i];
        mxSetFieldByNumber(plhs[0], 0, 9, sd);
    } else if (c == "release") {
        cleanup();
        if (mexIsLocked()) mexUnlock();
    } else {
        mexErrMsgIdAndTxt("qmri:usage", "unknown command '%s'", cmd);
    }
}

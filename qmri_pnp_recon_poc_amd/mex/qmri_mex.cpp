// qmri_mex.cpp -- MATLAB gateway for libqmri.so (compile-gated: needs mex.h, which is absent from the build image).
//
//   mex -R2018a qmri_mex.cpp -I../../include -L.. -lqmri        (interleaved-complex API: mxComplexDouble == (re,im) doubles)
//
// One mexFunction with a leading command string; a persistent context is created on first use and released by
// mexAtExit.  Every libqmri status != 0 becomes mexErrMsgIdAndTxt('qmri:<code>', qmri_last_error(ctx)), which is how
// the reference's plugins report errors (MATLAB exceptions, denoiseImage_PnP_ADMM.m:123-135).
// The MATLAB wrappers in ../matlab give these commands the reference's own signatures.
#include "mex.h"
#include "qmri.h"

#include <cstring>
#include <string>

static qmri_ctx* g_ctx = nullptr;

static void cleanup() { if (g_ctx) { qmri_destroy(g_ctx); g_ctx = nullptr; } }

static void check(int st) {
    if (st == QMRI_OK) return;
    char id[32];
    snprintf(id, sizeof id, "qmri:err%d", -st);
    mexErrMsgIdAndTxt(id, "%s", qmri_last_error(g_ctx));
}

static qmri_ctx* ctx() {
    if (!g_ctx) {
        int st = qmri_create(0, &g_ctx);
        if (st != QMRI_OK) mexErrMsgIdAndTxt("qmri:create", "%s", qmri_last_error(nullptr));
        mexAtExit(cleanup);
        mexLock();
    }
    return g_ctx;
}

static double scalar_field(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = mxGetField(s, 0, name);
    return f ? mxGetScalar(f) : dflt;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("qmri:usage", "first argument must be a command string");
    char cmd[64];
    mxGetString(prhs[0], cmd, sizeof cmd);
    const std::string c(cmd);

    if (c == "set_operator") {                       // qmri_mex('set_operator', N, M, V, frame_ptr(int32), kidx(int32))
        const int N = (int)mxGetScalar(prhs[1]), M = (int)mxGetScalar(prhs[2]);
        const mxArray* V = prhs[3];
        const int T = (int)mxGetM(V), s = (int)mxGetN(V);
        check(qmri_set_operator(ctx(), N, M, s, T, mxGetDoubles(V), (const int32_t*)mxGetData(prhs[4]),
                                (const int32_t*)mxGetData(prhs[5]), 1));
    } else if (c == "build_spiral" || c == "build_epi") {   // [frame_ptr, kidx] = qmri_mex('build_spiral', N, S, T)
        const int N = (int)mxGetScalar(prhs[1]);
        int m = 0;
        if (c == "build_spiral") {
            const int S = (int)mxGetScalar(prhs[2]), T = (int)mxGetScalar(prhs[3]);
            plhs[0] = mxCreateNumericMatrix(T + 1, 1, mxINT32_CLASS, mxREAL);
            mxArray* k = mxCreateNumericMatrix((size_t)S * T, 1, mxINT32_CLASS, mxREAL);
            check(qmri_build_spiral(ctx(), N, S, T, (int32_t*)mxGetData(plhs[0]), (int32_t*)mxGetData(k), S * T, &m));
            mxSetM(k, m);
            plhs[1] = k;
        } else {
            const int M = (int)mxGetScalar(prhs[2]);
            const double pct = mxGetScalar(prhs[3]);
            const int T = (int)mxGetScalar(prhs[4]);
            const int cap = N * M * T;
            plhs[0] = mxCreateNumericMatrix(T + 1, 1, mxINT32_CLASS, mxREAL);
            mxArray* k = mxCreateNumericMatrix((size_t)cap, 1, mxINT32_CLASS, mxREAL);
            check(qmri_build_epi(ctx(), N, M, pct, T, (int32_t*)mxGetData(plhs[0]), (int32_t*)mxGetData(k), cap, &m));
            mxSetM(k, m);
            plhs[1] = k;
        }
    } else if (c == "forward") {                     // y = qmri_mex('forward', x)   (F.forward, main_recon_tsmis_FFT.m:228)
        int m = 0;
        check(qmri_operator_m(ctx(), &m));
        plhs[0] = mxCreateDoubleMatrix(m, 1, mxCOMPLEX);
        const bool cx = mxIsComplex(prhs[1]);
        check(qmri_forward(ctx(), cx ? (const void*)mxGetComplexDoubles(prhs[1]) : (const void*)mxGetDoubles(prhs[1]), cx,
                           mxGetComplexDoubles(plhs[0])));
    } else if (c == "adjoint") {                     // x = qmri_mex('adjoint', y, [N M s])   (F.adjoint, :229)
        const double* d = mxGetDoubles(prhs[2]);
        const mwSize dims[3] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2]};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        check(qmri_adjoint(ctx(), mxGetComplexDoubles(prhs[1]), mxGetComplexDoubles(plhs[0])));
    } else if (c == "set_denoiser") {                // qmri_mex('set_denoiser', weights(single), in_nc, out_nc, nc(1x4), nb, residual_noise, H, W)
        qmri_net_desc d;
        d.arch = QMRI_ARCH_UNETRES;
        d.in_nc = (int)mxGetScalar(prhs[2]); d.out_nc = (int)mxGetScalar(prhs[3]);
        const double* nc = mxGetDoubles(prhs[4]);
        for (int i = 0; i < 4; ++i) d.nc[i] = (int)nc[i];
        d.nb = (int)mxGetScalar(prhs[5]); d.residual_noise = (int)mxGetScalar(prhs[6]);
        check(qmri_set_denoiser(ctx(), &d, (const float*)mxGetData(prhs[1]), mxGetNumberOfElements(prhs[1]) * 4,
                                (int)mxGetScalar(prhs[7]), (int)mxGetScalar(prhs[8]), 1));
    } else if (c == "load_onnx") {                   // [in_nc, out_nc] = qmri_mex('load_onnx', denoiser_path, residual_noise, H, W)
        // the weight-loading half of `Net = importONNXNetwork(denoiser_path, ...)` (main_recon_tsmis_FFT.m:138) + set_denoiser
        char path[4096];
        if (mxGetString(prhs[1], path, sizeof path)) mexErrMsgIdAndTxt("qmri:usage", "denoiser_path must be a char vector");
        qmri_net_desc d;
        size_t n = 0;
        if (qmri_onnx_read_unetres(path, &d, nullptr, 0, &n) != QMRI_OK) mexErrMsgIdAndTxt("qmri:onnx", "%s", qmri_last_error(nullptr));
        mxArray* w = mxCreateNumericMatrix(n, 1, mxSINGLE_CLASS, mxREAL);
        if (qmri_onnx_read_unetres(path, &d, (float*)mxGetData(w), n, &n) != QMRI_OK) mexErrMsgIdAndTxt("qmri:onnx", "%s", qmri_last_error(nullptr));
        d.residual_noise = (int)mxGetScalar(prhs[2]);
        check(qmri_set_denoiser(ctx(), &d, (const float*)mxGetData(w), n * 4, (int)mxGetScalar(prhs[3]), (int)mxGetScalar(prhs[4]), 1));
        mxDestroyArray(w);
        plhs[0] = mxCreateDoubleScalar((double)d.in_nc);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)d.out_nc);
    } else if (c == "denoise") {                     // I = qmri_mex('denoise', A, out_nc)   (param.net, :164)
        const mwSize* dm = mxGetDimensions(prhs[1]);
        const int nd = (int)mxGetNumberOfDimensions(prhs[1]);
        if (mxIsComplex(prhs[1]) || !mxIsDouble(prhs[1]) || nd > 4)
            mexErrMsgIdAndTxt("images:denoiseImage:invalidImageFormat", "A must be a real double H x W x C (x N) array");
        const int H = (int)dm[0], W = (int)dm[1], C = nd > 2 ? (int)dm[2] : 1, B = nd > 3 ? (int)dm[3] : 1;
        const mwSize od[4] = {(mwSize)H, (mwSize)W, (mwSize)mxGetScalar(prhs[2]), (mwSize)B};
        plhs[0] = mxCreateNumericArray(4, od, mxDOUBLE_CLASS, mxREAL);
        check(qmri_denoise(ctx(), mxGetDoubles(prhs[1]), H, W, C, B, mxGetDoubles(plhs[0])));
    } else if (c == "pnp_admm") {                    // [x, diag, lsqr_iters] = qmri_mex('pnp_admm', y, param_struct, X0, gt, [N M s])
        const mxArray* P = prhs[2];
        qmri_admm_params p;
        p.gamma = scalar_field(P, "gamma", 0.05);
        p.iters = (int)scalar_field(P, "iter", 100);
        p.cg_tol = scalar_field(P, "cg_tol", 1e-4);
        p.cg_maxit = 100;                                           // literal in PnP_ADMM.m:102
        p.solver = (int)scalar_field(P, "solver", QMRI_SOLVER_LSQR);
        p.denoiser_type = (int)scalar_field(P, "multi_level", 0);
        p.noise_std = scalar_field(P, "noise_std", 0.01);
        p.want_diag = nlhs > 1;
        const double* d = mxGetDoubles(prhs[5]);
        const mwSize dims[3] = {(mwSize)d[0], (mwSize)d[1], (mwSize)d[2]};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        mxArray* diag = mxCreateDoubleMatrix(2, p.iters > 0 ? p.iters : 1, mxREAL);
        mxArray* li = mxCreateNumericMatrix(p.iters > 0 ? p.iters : 1, 1, mxINT32_CLASS, mxREAL);
        const void* x0 = mxIsEmpty(prhs[3]) ? nullptr : (const void*)mxGetComplexDoubles(prhs[3]);
        const void* gt = mxIsEmpty(prhs[4]) ? nullptr : (const void*)mxGetComplexDoubles(prhs[4]);
        check(qmri_pnp_admm(ctx(), mxGetComplexDoubles(prhs[1]), &p, x0, gt, mxGetComplexDoubles(plhs[0]),
                            p.want_diag ? mxGetDoubles(diag) : nullptr, (int32_t*)mxGetData(li)));
        if (nlhs > 1) plhs[1] = diag; else mxDestroyArray(diag);
        if (nlhs > 2) plhs[2] = li; else mxDestroyArray(li);
    } else if (c == "lrtv") {                        // [x, info] = qmri_mex('lrtv', y, param_struct, [N M s])   (FISTA_deep, main_recon_tsmis_FFT.m:273-282)
        const mxArray* P = prhs[2];
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
        check(qmri_set_dictionary(ctx(), (int)mxGetM(prhs[1]), (int)mxGetN(prhs[1]), (int)mxGetN(prhs[3]),
                                  (const float*)mxGetData(prhs[1]), (const float*)mxGetData(prhs[2]), (const float*)mxGetData(prhs[3])));
    } else if (c == "dict_match") {                  // [qmap, pd, mt, dm, xfit] = qmri_mex('dict_match', X(Npix x s complex double), Q)
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
    } else if (c == "release") {
        cleanup();
        if (mexIsLocked()) mexUnlock();
    } else {
        mexErrMsgIdAndTxt("qmri:usage", "unknown command '%s'", cmd);
    }
}

// mc_kernels.hip -- multi-coil x-update of PnP-ADMM: an EXTENSION (BASELINE.json configs[4] names a "complex-valued multi-coil forward op").
//
// The reference simulates a single coil (README.md:63) and has no multi-coil operator, so nothing here has a reference counterpart and nothing can pin
// it: PARITY UNPINNED, stated in include/qmri.h and in the oracle's restatement.  What it generalises is the one line PnP_ADMM.m:102
//     x = lsqr(@afun, [y; sqrt(r) z], cg_tol, cg_iter, [], [], x0),   afun: B = [A; sqrt(r) I]  (PnP_ADMM.m:153-171)
// with A replaced by the SENSE operator of api_core.cpp (qmri_forward_mc / qmri_adjoint_mc):  A_mc x = [A (C_j . x)]_j,  A_mc' y = sum_j conj(C_j) . A' y_j.
// Coil maps act in image space, so A_mc'A_mc is no longer block-diagonal in k-space and the k-space iteration of kslsqr_kernels.hip does not apply:
// this is the image-domain LSQR (the recurrences, stop rules and their order exactly as oracle/orc_lsqr.c restates MATLAB's lsqr), two batched
// transforms per coil chunk and iteration, the scalars on the host (three small device -> host copies per iteration: an extension, not a tuned path).
// Sums are 256 block partials added on the host in block order: run-to-run reproducible.
#include <cfloat>
#include <cmath>
#include <vector>
#include "qmri_internal.h"

namespace {
constexpr int MT = 256, MB = 256;     // threads per block, blocks per reduction

__device__ __forceinline__ double mc_block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < MT / 64; ++w) t += sh[w];
    __syncthreads();
    return t;
}
// u = a - alpha * u   (a nullable: u = -alpha * u), partial |u|^2 -> part[blockIdx.x]      (doubles: a complex vector as 2n reals)
__global__ __launch_bounds__(MT) void k_mc_lin(size_t n, const double* __restrict__ a, double sa, double alpha, double* __restrict__ u, double* __restrict__ part) {
    __shared__ double sh[MT / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)MB * MT) {
        const double v = (a ? a[i] * sa : 0.0) - alpha * u[i];
        u[i] = v;
        acc += v * v;
    }
    const double t = mc_block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// v = (t + ub * sr) - beta * v, partial |v|^2
__global__ __launch_bounds__(MT) void k_mc_vupd(size_t n, const double* __restrict__ t, const double* __restrict__ ub, double sr, double beta, double* __restrict__ v,
                                                double* __restrict__ part) {
    __shared__ double sh[MT / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)MB * MT) {
        const double w = (t[i] + ub[i] * sr) - beta * v[i];
        v[i] = w;
        acc += w * w;
    }
    const double tt = mc_block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tt;
}
__global__ __launch_bounds__(MT) void k_mc_scale(size_t n, double s, double* __restrict__ u) {
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)gridDim.x * MT) u[i] *= s;
}
// d = (v - thet d) / rho ; partials |d|^2 and |x|^2
__global__ __launch_bounds__(MT) void k_mc_dupd(size_t n, const double* __restrict__ v, double thet, double rho, double* __restrict__ d, const double* __restrict__ x,
                                                double* __restrict__ part) {
    __shared__ double sh[MT / 64];
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)MB * MT) {
        const double w = (v[i] - thet * d[i]) / rho;
        d[i] = w;
        a += w * w; b += x[i] * x[i];
    }
    const double ta = mc_block_sum(a, sh);
    const double tb = mc_block_sum(b, sh);
    if (threadIdx.x == 0) { part[blockIdx.x] = ta; part[MB + blockIdx.x] = tb; }
}
__global__ __launch_bounds__(MT) void k_mc_axpy(size_t n, double phi, const double* __restrict__ d, double* __restrict__ x) {
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)gridDim.x * MT) x[i] += phi * d[i];
}
__global__ __launch_bounds__(MT) void k_mc_sq(size_t n, const double* __restrict__ a, double* __restrict__ part) {
    __shared__ double sh[MT / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * MT + threadIdx.x; i < n; i += (size_t)MB * MT) acc += a[i] * a[i];
    const double t = mc_block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

struct McBuf { double2 *ut = nullptr, *ub = nullptr, *v = nullptr, *d = nullptr, *t = nullptr, *scr = nullptr; double* part = nullptr; };

int mc_sum(qmri_ctx* ctx, const double* d_part, int cnt, double* out) {          // block partials -> host, added in block order
    std::vector<double> h((size_t)cnt);
    QMRI_HIP(ctx, hipMemcpyAsync(h.data(), d_part, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (double v : h) s += v;
    *out = s;
    return QMRI_OK;
}
// tm[j] = A (C_j . x) for all coils, in chunks of the operator's max_batch (qmri_forward_mc's device half); then u1 = tm * st - alpha * u1 chunk by chunk
int mc_forward_lin(qmri_ctx* ctx, const McBuf& b, const double2* x, double st, double alpha, double2* u1, double* sumsq) {
    OpHost& o = ctx->op;
    const size_t n = (size_t)o.N * o.M * o.s, plane = (size_t)o.N * o.M;
    double tot = 0.0;
    for (int j0 = 0; j0 < o.ncoil; j0 += o.maxB) {
        const int cnt = std::min(o.maxB, o.ncoil - j0);
        QMRI_TRY(ew_launch_coil_mul(ctx, n, plane, cnt, x, o.d_coils + (size_t)j0 * plane, b.scr));
        QMRI_TRY(dc_launch_fwd(ctx, qmri_opdev(ctx), o.ls, DC_PLAIN, cnt, b.scr, o.d_tmp, o.d_ya, nullptr));
        const size_t len = (size_t)2 * cnt * o.m;
        k_mc_lin<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(len, (const double*)o.d_ya, st, alpha, (double*)(u1 + (size_t)j0 * o.m), b.part);
        QMRI_HIP(ctx, hipGetLastError());
        double s = 0.0;
        QMRI_TRY(mc_sum(ctx, b.part, MB, &s));
        tot += s;
    }
    *sumsq = tot;
    return QMRI_OK;
}
// t = A_mc' u1
int mc_adjoint(qmri_ctx* ctx, const McBuf& b, const double2* u1, double2* t) {
    OpHost& o = ctx->op;
    const size_t n = (size_t)o.N * o.M * o.s, plane = (size_t)o.N * o.M;
    for (int j0 = 0; j0 < o.ncoil; j0 += o.maxB) {
        const int cnt = std::min(o.maxB, o.ncoil - j0);
        QMRI_TRY(dc_launch_adj(ctx, qmri_opdev(ctx), cnt, u1 + (size_t)j0 * o.m, o.d_tmp, b.scr));
        QMRI_TRY(ew_launch_coil_sum(ctx, n, plane, cnt, b.scr, o.d_coils + (size_t)j0 * plane, t, j0 > 0));
    }
    return QMRI_OK;
}
}  // namespace

// LSQR on [A_mc; sqrt(r) I] x = [y; sqrt(r) z] from x0 = d_x (device, overwritten with the solution).  d_y: [ncoil][m] in the ABI's frame-major order.
// Recurrences and stop rules in the order of oracle/orc_lsqr.c (MATLAB's lsqr as documented; flags 0 converged, 1 maxit, 3 stagnation).
int qmri_lsqr_mc_dev(qmri_ctx* ctx, const double2* d_y, const double2* d_z, double r, double tol, int maxit, double2* d_x, int32_t* iters_out, int32_t* flag_out) {
    OpHost& o = ctx->op;
    if (!o.ncoil) { qmri_set_error(ctx, "no coil maps set: call qmri_set_coils first"); return QMRI_ERR_STATE; }
    const size_t n = (size_t)o.N * o.M * o.s, n2 = 2 * n, mtot = (size_t)o.ncoil * o.m;
    McBuf b;
    void* owned[7] = {};
    auto alloc = [&](void** p, size_t bytes, int slot) { if (hipMalloc(p, bytes) != hipSuccess) return false; owned[slot] = *p; return true; };
    int rc = QMRI_OK;
    if (!alloc((void**)&b.ut, mtot * sizeof(double2), 0) || !alloc((void**)&b.ub, n * sizeof(double2), 1) || !alloc((void**)&b.v, n * sizeof(double2), 2) ||
        !alloc((void**)&b.d, n * sizeof(double2), 3) || !alloc((void**)&b.t, n * sizeof(double2), 4) || !alloc((void**)&b.scr, (size_t)o.maxB * n * sizeof(double2), 5) ||
        !alloc((void**)&b.part, (size_t)2 * MB * sizeof(double), 6)) {
        qmri_set_error(ctx, "hipMalloc failed in the multi-coil x-update");
        rc = QMRI_ERR_NOMEM;
    }
    int iter = maxit, flag = 1;
    do {
        if (rc != QMRI_OK) break;
        const double sr = std::sqrt(r);
        double sy = 0.0, sz = 0.0, s1 = 0.0, s2 = 0.0;
#define MC_TRY(x) { rc = (x); if (rc != QMRI_OK) break; }
#define MC_LAUNCH(...) { __VA_ARGS__; if (hipGetLastError() != hipSuccess) { qmri_set_error(ctx, "kernel launch failed in the multi-coil x-update"); rc = QMRI_ERR_HIP; break; } }
        // n2b = norm([y; sqrt(r) z])
        MC_LAUNCH((k_mc_sq<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(2 * mtot, (const double*)d_y, b.part)));
        MC_TRY(mc_sum(ctx, b.part, MB, &sy));
        MC_LAUNCH((k_mc_sq<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)d_z, b.part)));
        MC_TRY(mc_sum(ctx, b.part, MB, &sz));
        const double n2b = std::sqrt(sy + r * sz), tolb = tol * n2b;
        // u = b - B x0:  u1 = y - A_mc x0  (k_mc_lin with u1 := y first),  u2 = sqrt(r) z - sqrt(r) x0
        if (hipMemcpyAsync(b.ut, d_y, mtot * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(b.ub, d_x, n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(b.d, 0, n * sizeof(double2), ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        MC_TRY(mc_forward_lin(ctx, b, d_x, -1.0, -1.0, b.ut, &s1));                      // u1 = -(A x0) + y
        MC_LAUNCH((k_mc_lin<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)d_z, sr, sr, (double*)b.ub, b.part)));   // u2 = z sr - sr x0
        MC_TRY(mc_sum(ctx, b.part, MB, &s2));
        double beta = std::sqrt(s1 + s2), normr = beta;
        if (beta != 0.0) {
            MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(2 * mtot, 1.0 / beta, (double*)b.ut)));
            MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, 1.0 / beta, (double*)b.ub)));
        }
        double c = 1.0, s = 0.0, phibar = beta;
        // v = B' u
        MC_TRY(mc_adjoint(ctx, b, b.ut, b.t));
        if (hipMemsetAsync(b.v, 0, n * sizeof(double2), ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        MC_LAUNCH((k_mc_vupd<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)b.t, (const double*)b.ub, sr, 0.0, (double*)b.v, b.part)));
        MC_TRY(mc_sum(ctx, b.part, MB, &s1));
        double alpha = std::sqrt(s1);
        if (alpha != 0.0) MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, 1.0 / alpha, (double*)b.v)));
        double normar = alpha * beta;
        if (normar == 0.0 || n2b == 0.0) { flag = 0; iter = 0; break; }
        double norma = 0.0;
        int stag = 0;
        for (int ii = 1; ii <= maxit; ++ii) {
            // u = B v - alpha u
            MC_TRY(mc_forward_lin(ctx, b, b.v, 1.0, alpha, b.ut, &s1));
            MC_LAUNCH((k_mc_lin<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)b.v, sr, alpha, (double*)b.ub, b.part)));
            MC_TRY(mc_sum(ctx, b.part, MB, &s2));
            beta = std::sqrt(s1 + s2);
            MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(2 * mtot, 1.0 / beta, (double*)b.ut)));
            MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, 1.0 / beta, (double*)b.ub)));
            norma = std::sqrt(norma * norma + alpha * alpha + beta * beta);
            const double thet = -s * alpha, rhot = c * alpha, rho = std::sqrt(rhot * rhot + beta * beta);
            c = rhot / rho;
            s = -beta / rho;
            const double phi = c * phibar;
            if (phi == 0.0) stag = 1;
            phibar = s * phibar;
            MC_LAUNCH((k_mc_dupd<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)b.v, thet, rho, (double*)b.d, (const double*)d_x, b.part)));
            double sd = 0.0, sx = 0.0;
            {
                std::vector<double> h((size_t)2 * MB);
                if (hipMemcpyAsync(h.data(), b.part, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
                for (int k = 0; k < MB; ++k) { sd += h[k]; sx += h[MB + k]; }
            }
            if (std::fabs(phi) * std::sqrt(sd) < DBL_EPSILON * std::sqrt(sx)) stag++; else stag = 0;
            if (normar / (norma * normr) <= tol) { flag = 0; iter = ii - 1; break; }
            if (normr <= tolb) { flag = 0; iter = ii - 1; break; }
            if (stag >= 3) { flag = 3; iter = ii - 1; break; }
            MC_LAUNCH((k_mc_axpy<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, phi, (const double*)b.d, (double*)d_x)));
            normr = std::fabs(s) * normr;
            // v = B' u - beta v
            MC_TRY(mc_adjoint(ctx, b, b.ut, b.t));
            MC_LAUNCH((k_mc_vupd<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, (const double*)b.t, (const double*)b.ub, sr, beta, (double*)b.v, b.part)));
            MC_TRY(mc_sum(ctx, b.part, MB, &s1));
            alpha = std::sqrt(s1);
            MC_LAUNCH((k_mc_scale<<<dim3(MB), dim3(MT), 0, ctx->stream>>>(n2, 1.0 / alpha, (double*)b.v)));
            normar = alpha * std::fabs(s * phi);
        }
#undef MC_TRY
#undef MC_LAUNCH
    } while (0);
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : owned) if (p) (void)hipFree(p);
    if (rc == QMRI_ERR_HIP && ctx->err.empty()) qmri_set_error(ctx, "HIP failure in the multi-coil x-update");
    if (rc != QMRI_OK) return rc;
    if (iters_out) *iters_out = iter;
    if (flag_out) *flag_out = flag;
    return QMRI_OK;
}

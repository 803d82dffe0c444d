// ew_kernels.hip -- the elementwise stages of the PnP-ADMM loop (gfx950), all HBM/L2-bound streaming kernels.
//
// Reference semantics (PnP_ADMM.m):
//   :115-118  v = real(x + uold)
//   :121,174-184  norm_zero_to_one: global min / max over the whole N x M x s stack, v = (v - min)/(max - min)
//   :132      multi_level: cat(3, v, noise_map), noise_map = constant plane (build_noise_map.m:19)
//   :138,187-192  undo_norm_zero_to_one: v = v*range + min
//   :144      uold = uold + x - v
//   :106-109  the two printed diagnostics
// and the casts of denoiseImage_PnP_ADMM.m:72-77 (double -> single) / :111-115 (single -> double), :99-104
// (residual_noise).  fp64 outside the network, fp32 inside, exactly as the reference.
#include "qmri_internal.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
    return r;
}

__device__ __forceinline__ void block_minmax(double& lo, double& hi, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off, 64));
        hi = fmax(hi, __shfl_down(hi, off, 64));
    }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sh[2 * wid] = lo; sh[2 * wid + 1] = hi; }
    __syncthreads();
    lo = sh[0]; hi = sh[1];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) { lo = fmin(lo, sh[2 * i]); hi = fmax(hi, sh[2 * i + 1]); }
}

// pass 1: per-block min / max of real(x + u)
__global__ __launch_bounds__(NT) void k_minmax(size_t n, const double2* __restrict__ x, const double2* __restrict__ u,
                                                double* __restrict__ mm) {
    __shared__ double sh[2 * NT / 64];
    const int b = blockIdx.y;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    double lo = INFINITY, hi = -INFINITY;
    for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const double v = x[(size_t)b * n + i].x + u[(size_t)b * n + i].x;
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    block_minmax(lo, hi, sh);
    if (threadIdx.x == 0) {
        mm[((size_t)b * gridDim.x + blockIdx.x) * 2] = lo;
        mm[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = hi;
    }
}

// pass 2: reduce the partials (min/max are order independent), normalise, cast to single, append noise map
__global__ __launch_bounds__(NT) void k_normalise(size_t n, size_t plane, int in_nc, int multi_level, double noise_std,
                                                   const double2* __restrict__ x, const double2* __restrict__ u,
                                                   const double* __restrict__ mm, int nblk, double* __restrict__ norm,
                                                   float* __restrict__ in32) {
    __shared__ double sh[2 * NT / 64];
    const int b = blockIdx.y;
    double lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += NT) {
        lo = fmin(lo, mm[((size_t)b * nblk + i) * 2]);
        hi = fmax(hi, mm[((size_t)b * nblk + i) * 2 + 1]);
    }
    block_minmax(lo, hi, sh);
    const double range = hi - lo;                       // no zero-range guard, as PnP_ADMM.m:174-184
    if (blockIdx.x == 0 && threadIdx.x == 0) { norm[2 * b] = lo; norm[2 * b + 1] = range; }
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    float* dst = in32 + (size_t)b * in_nc * plane;
    for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const double v = x[(size_t)b * n + i].x + u[(size_t)b * n + i].x;
        dst[i] = (float)((v - lo) / range);
    }
    if (multi_level) {
        const size_t c0 = (size_t)blockIdx.x * ((plane + gridDim.x - 1) / gridDim.x);
        const size_t c1 = (c0 + (plane + gridDim.x - 1) / gridDim.x < plane) ? c0 + (plane + gridDim.x - 1) / gridDim.x : plane;
        for (size_t i = c0 + threadIdx.x; i < c1; i += NT) dst[n + i] = (float)noise_std;
    }
}

// v = double(I)*range + min ;  uold = uold + x - v     (I = CNN output, or input - CNN output)
__global__ __launch_bounds__(NT) void k_unnormalise_dual(size_t n, size_t in_stride, const float* __restrict__ out32,
                                                          const float* __restrict__ in32, int residual_noise,
                                                          const double* __restrict__ norm, const double2* __restrict__ x,
                                                          double2* __restrict__ u, double2* __restrict__ v) {
    const int b = blockIdx.y;
    const double lo = norm[2 * b], range = norm[2 * b + 1];
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    float I = out32[(size_t)b * n + i];
    if (residual_noise) I = in32[(size_t)b * in_stride + i] - I;
    const double vv = (double)I * range + lo;
    const double2 xv = x[(size_t)b * n + i];
    double2 uv = u[(size_t)b * n + i];
    uv.x = uv.x + xv.x - vv;
    uv.y = uv.y + xv.y - 0.0;
    u[(size_t)b * n + i] = uv;
    v[(size_t)b * n + i] = make_double2(vv, 0.0);
}

// ||gt - x||^2 partials, then the final diagnostics for this ADMM iteration
__global__ __launch_bounds__(NT) void k_diffnorm(size_t n, const double2* __restrict__ a, const double2* __restrict__ c,
                                                  double* __restrict__ part) {
    __shared__ double sh[NT / 64];
    const int b = blockIdx.y;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    double acc = 0.0, ref = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const double2 p = a[(size_t)b * n + i], q = c[(size_t)b * n + i];
        const double dx = p.x - q.x, dy = p.y - q.y;
        acc += dx * dx + dy * dy;
        ref += p.x * p.x + p.y * p.y;
    }
    const double t0 = block_sum(acc, sh);
    const double t1 = block_sum(ref, sh);
    if (threadIdx.x == 0) {
        part[((size_t)b * gridDim.x + blockIdx.x) * 2] = t0;
        part[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = t1;
    }
}

__global__ __launch_bounds__(NT) void k_diag_final(const LsqrState* __restrict__ st, const double* __restrict__ py, int npy,
                                                    const double* __restrict__ pg, int npg, int have_gt,
                                                    double* __restrict__ diag, int iters_total, int it) {
    __shared__ double sh[NT / 64];
    const int b = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < npy; i += NT) a += py[(size_t)b * npy + i];
    const double ry = block_sum(a, sh);
    double g = 0.0, r = 0.0;
    for (int i = threadIdx.x; i < npg; i += NT) { g += pg[((size_t)b * npg + i) * 2]; r += pg[((size_t)b * npg + i) * 2 + 1]; }
    const double dg = block_sum(g, sh);
    const double rg = block_sum(r, sh);
    if (threadIdx.x == 0) {
        diag[((size_t)b * iters_total + it) * 2] = sqrt(ry) / sqrt(st[b].ny2);                         // PnP_ADMM.m:106
        diag[((size_t)b * iters_total + it) * 2 + 1] = have_gt ? sqrt(dg) / sqrt(rg) : NAN;            // PnP_ADMM.m:107
    }
}

__global__ __launch_bounds__(NT) void k_cast_d2f(size_t count, const double* __restrict__ in, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < count) out[i] = (float)in[i];
}

__global__ __launch_bounds__(NT) void k_denoise_out(size_t plane, int out_nc, int in_nc, const float* __restrict__ out32,
                                                     const float* __restrict__ in32, int residual_noise,
                                                     double* __restrict__ out) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= plane * out_nc) return;
    float I = out32[(size_t)b * out_nc * plane + i];
    if (residual_noise) I = in32[(size_t)b * in_nc * plane + i] - I;      // denoiseImage_PnP_ADMM.m:101
    out[(size_t)b * out_nc * plane + i] = (double)I;
}

__global__ __launch_bounds__(NT) void k_real_to_complex(size_t count, const double* __restrict__ in, double2* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < count) out[i] = make_double2(in[i], 0.0);
}

}  // namespace

int ew_launch_minmax_normalise(qmri_ctx* ctx, int B, size_t n, size_t plane, int in_nc, int multi_level, double noise_std,
                               const double2* x, const double2* u, double* mm, double* norm, int nblk, float* in32) {
    k_minmax<<<dim3(nblk, B), dim3(NT), 0, ctx->stream>>>(n, x, u, mm);
    k_normalise<<<dim3(nblk, B), dim3(NT), 0, ctx->stream>>>(n, plane, in_nc, multi_level, noise_std, x, u, mm, nblk, norm, in32);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_unnormalise_dual(qmri_ctx* ctx, int B, size_t n, const float* out32, const float* in32, int residual_noise,
                               const double* norm, const double2* x, double2* u, double2* v) {
    // in32 holds in_nc planes per slice; its slice stride is passed through ctx->net
    const size_t in_stride = (size_t)ctx->net.desc.in_nc * ctx->net.H * ctx->net.W;
    k_unnormalise_dual<<<dim3((unsigned)((n + NT - 1) / NT), B), dim3(NT), 0, ctx->stream>>>(n, in_stride, out32, in32,
                                                                                             residual_noise, norm, x, u, v);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_diag(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* x, const double2* gt,
                   double* pd, double* diag_slot, int iters_total, int it) {
    // pd layout: [B][N] data-fidelity partials (written by k_fwd_w<DC_DIAG>) followed by [B][nblk_z][2] gt partials
    const size_t n = (size_t)op.s * op.N * op.M;
    double* pg = pd + (size_t)B * op.N;
    if (gt) k_diffnorm<<<dim3(ls.nblk_z, B), dim3(NT), 0, ctx->stream>>>(n, gt, x, pg);
    k_diag_final<<<dim3(B), dim3(NT), 0, ctx->stream>>>(ls.st, pd, op.N, pg, ls.nblk_z, gt ? 1 : 0, diag_slot, iters_total, it);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_cast(qmri_ctx* ctx, size_t count, const double* in, float* out) {
    k_cast_d2f<<<dim3((unsigned)((count + NT - 1) / NT)), dim3(NT), 0, ctx->stream>>>(count, in, out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_denoise_out(qmri_ctx* ctx, size_t plane, int out_nc, int in_nc, int B, const float* out32, const float* in32,
                          int residual_noise, double* out) {
    k_denoise_out<<<dim3((unsigned)((plane * out_nc + NT - 1) / NT), B), dim3(NT), 0, ctx->stream>>>(plane, out_nc, in_nc, out32,
                                                                                                   in32, residual_noise, out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_real_to_complex(qmri_ctx* ctx, size_t count, const double* in, double2* out) {
    k_real_to_complex<<<dim3((unsigned)((count + NT - 1) / NT)), dim3(NT), 0, ctx->stream>>>(count, in, out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

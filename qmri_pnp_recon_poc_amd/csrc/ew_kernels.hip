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
// write-through store (see dc_device.h: the output goes to another launch; nothing dirty is left for the kernel boundary)
__device__ __forceinline__ void st_wt(double2* p, double2 v) {
    typedef double d2v_ __attribute__((ext_vector_type(2)));
    const d2v_ t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
}  // namespace

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

// pass 2: reduce the partials (min/max are order independent), normalise, cast to single, append noise map.
// The workgroup's share of x and u is requested BEFORE the partials are reduced (the loads do not depend on min / max): one memory latency
// for the launch instead of two plus one per loop iteration; shares beyond NRQ * NT elements per workgroup fall back to the plain loop.
constexpr int NRQ = 16;
__global__ __launch_bounds__(NT) void k_normalise(size_t n, int plane, int H, int s, int multi_level, double noise_std,
                                                   const double2* __restrict__ x, const double2* __restrict__ u,
                                                   const double* __restrict__ mm, int nblk, double* __restrict__ norm,
                                                   float* __restrict__ in32, int php, int pplane, size_t pbs) {
    __shared__ double sh[2 * NT / 64];
    const int b = blockIdx.y;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    double xr[NRQ], ur[NRQ];
#pragma unroll
    for (int q = 0; q < NRQ; ++q) {
        const size_t i = i0 + threadIdx.x + (size_t)NT * q;
        const size_t ic = (i < i1) ? i : i0;
        xr[q] = x[(size_t)b * n + ic].x; ur[q] = u[(size_t)b * n + ic].x;
    }
    double lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += NT) {
        lo = fmin(lo, mm[((size_t)b * nblk + i) * 2]);
        hi = fmax(hi, mm[((size_t)b * nblk + i) * 2 + 1]);
    }
    block_minmax(lo, hi, sh);
    const double range = hi - lo;                       // no zero-range guard, as PnP_ADMM.m:174-184
    if (blockIdx.x == 0 && threadIdx.x == 0) { norm[2 * b] = lo; norm[2 * b + 1] = range; }
    float* dst = in32 + (size_t)b * pbs;                 // padded planes [c][w+1][h+1], zero halo untouched
    auto put = [&](size_t i, double v) __attribute__((always_inline)) {
        const int c = (int)(i / plane), rem = (int)(i - (size_t)c * plane);
        const int w = rem / H, h = rem - w * H;
        dst[(size_t)c * pplane + (size_t)(w + 1) * php + h + 1] = (float)((v - lo) / range);
    };
#pragma unroll
    for (int q = 0; q < NRQ; ++q) {
        const size_t i = i0 + threadIdx.x + (size_t)NT * q;
        if (i < i1) put(i, xr[q] + ur[q]);
    }
    for (size_t i = i0 + threadIdx.x + (size_t)NT * NRQ; i < i1; i += NT) put(i, x[(size_t)b * n + i].x + u[(size_t)b * n + i].x);
    if (multi_level) {
        const int per = (plane + gridDim.x - 1) / gridDim.x;
        const int c0 = blockIdx.x * per, c1 = (c0 + per < plane) ? c0 + per : plane;
        for (int i = c0 + threadIdx.x; i < c1; i += NT) {
            const int w = i / H, h = i - w * H;
            dst[(size_t)s * pplane + (size_t)(w + 1) * php + h + 1] = (float)noise_std;
        }
    }
}

// v = double(I)*range + min ;  uold = uold + x - v     (I = CNN output, or input - CNN output)      PnP_ADMM.m:138,144
// and, for the next iteration's x-update, z = v - uold with the partial sums of ||z||^2 (PnP_ADMM.m:102) -- the same
// partition and summation order as k_prepare_z, which only the first iteration still needs.
__global__ __launch_bounds__(NT) void k_unnormalise_dual(size_t n, int plane, int H, int php, int pplane, size_t out_bs,
                                                          size_t in_bs, const float* __restrict__ out32,
                                                          const float* __restrict__ in32, int residual_noise,
                                                          const double* __restrict__ norm, const double2* __restrict__ x,
                                                          double2* __restrict__ u, double2* __restrict__ v,
                                                          double2* __restrict__ z, double* __restrict__ pz) {
    __shared__ double red[NT / 64];
    const int b = blockIdx.y;
    const double lo = norm[2 * b], range = norm[2 * b + 1];
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    double acc = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const int c = (int)(i / plane), rem = (int)(i - (size_t)c * plane);
        const int w = rem / H, h = rem - w * H;
        const size_t pi = (size_t)c * pplane + (size_t)(w + 1) * php + h + 1;
        float I = out32[(size_t)b * out_bs + pi];
        if (residual_noise) I = in32[(size_t)b * in_bs + pi] - I;
        const double vv = (double)I * range + lo;
        const double2 xv = x[(size_t)b * n + i];
        double2 uv = u[(size_t)b * n + i];
        uv.x = uv.x + xv.x - vv;
        uv.y = uv.y + xv.y - 0.0;
        st_wt(u + (size_t)b * n + i, uv);
        if (v) v[(size_t)b * n + i] = make_double2(vv, 0.0);      // (the ADMM loop passes no v: nothing reads it after the first iteration's z)
        const double2 zz = make_double2(vv - uv.x, 0.0 - uv.y);
        st_wt(z + (size_t)b * n + i, zz);
        acc += zz.x * zz.x + zz.y * zz.y;
    }
    // block total: shuffle tree, then the waves in order (k_prepare_z, which only the first iteration uses, has dc_device.h's DPP tree:
    // either is one fixed order, the partials only ever meet inside one solve's ||b||)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) r += red[i];
        pz[(size_t)b * gridDim.x + blockIdx.x] = r;
    }
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

// compact [B][C][W][H] (double or float) -> padded planes (im2single of a double is a cast, denoiseImage_PnP_ADMM.m:72-77)
template <typename T>
__global__ __launch_bounds__(NT) void k_pack(size_t count, int C, int plane, int H, int php, int pplane, size_t pbs,
                                              const T* __restrict__ in, float* __restrict__ out, float scale) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= count) return;
    const int c = (int)(i / plane), rem = (int)(i - (size_t)c * plane);
    const int w = rem / H, h = rem - w * H;
    out[(size_t)b * pbs + (size_t)c * pplane + (size_t)(w + 1) * php + h + 1] = (float)in[(size_t)b * count + i] * scale;   // (scale: a power of two)
}

// padded network output -> compact; optional residual I = input - CNN(input) (denoiseImage_PnP_ADMM.m:99-104,111-115)
template <typename T>
__global__ __launch_bounds__(NT) void k_unpack(size_t count, int plane, int H, int php, int pplane, size_t out_bs, size_t in_bs,
                                                const float* __restrict__ out32, const float* __restrict__ in32,
                                                int residual_noise, T* __restrict__ out, float scale) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= count) return;
    const int c = (int)(i / plane), rem = (int)(i - (size_t)c * plane);
    const int w = rem / H, h = rem - w * H;
    const size_t pi = (size_t)c * pplane + (size_t)(w + 1) * php + h + 1;
    float I = out32[(size_t)b * out_bs + pi];
    if (residual_noise) I = in32[(size_t)b * in_bs + pi] - I;
    out[(size_t)b * count + i] = (T)(I * scale);
}

__global__ __launch_bounds__(NT) void k_real_to_complex(size_t count, const double* __restrict__ in, double2* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < count) out[i] = make_double2(in[i], 0.0);
}

}  // namespace

int ew_launch_minmax_normalise(qmri_ctx* ctx, int B, size_t n, int plane, int H, int s, int multi_level, double noise_std,
                               const double2* x, const double2* u, double* mm, double* norm, int nblk, const PTensor& in32, bool mm_ready) {
    if (!mm_ready) k_minmax<<<dim3(nblk, B), dim3(NT), 0, ctx->stream>>>(n, x, u, mm);      // (else: nblk partials per slice are in mm already, k_adj_h)
    // (measured and removed, round 6: 2 x / 4 x as many workgroups with shorter shares -- the grid need not equal the number of min / max partials --
    //  865 ... 873 against 870 ADMM it/s, profiles/r06_c_ab_normalise_grid_multiplier_not_kept.txt)
    k_normalise<<<dim3(nblk, B), dim3(NT), 0, ctx->stream>>>(n, plane, H, s, multi_level, noise_std, x, u, mm, nblk, norm, in32.base1(),
                                                             in32.hp, (int)in32.plane(), in32.batch_stride());
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_unnormalise_dual(qmri_ctx* ctx, int B, size_t n, int plane, int H, const PTensor& out32, const PTensor& in32,
                               int residual_noise, const double* norm, const double2* x, double2* u, double2* v, double2* z,
                               double* pz, int nblk_z) {
    k_unnormalise_dual<<<dim3(nblk_z, B), dim3(NT), 0, ctx->stream>>>(
        n, plane, H, out32.hp, (int)out32.plane(), out32.batch_stride(), in32.batch_stride(), out32.base1(), in32.base1(), residual_noise, norm,
        x, u, v, z, pz);
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

int ew_launch_pack(qmri_ctx* ctx, int B, int C, int H, int W, const void* src, int src_is_double, const PTensor& dst, float scale) {
    const size_t count = (size_t)C * H * W;
    dim3 grid((unsigned)((count + NT - 1) / NT), B), blk(NT);
    if (src_is_double)
        k_pack<double><<<grid, blk, 0, ctx->stream>>>(count, C, H * W, H, dst.hp, (int)dst.plane(), dst.batch_stride(), (const double*)src, dst.base1(), scale);
    else
        k_pack<float><<<grid, blk, 0, ctx->stream>>>(count, C, H * W, H, dst.hp, (int)dst.plane(), dst.batch_stride(), (const float*)src, dst.base1(), scale);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_unpack(qmri_ctx* ctx, int B, int C, int H, int W, const PTensor& out32, const PTensor& in32, int residual_noise,
                     void* dst, int dst_is_double, float scale) {
    const size_t count = (size_t)C * H * W;
    dim3 grid((unsigned)((count + NT - 1) / NT), B), blk(NT);
    if (dst_is_double)
        k_unpack<double><<<grid, blk, 0, ctx->stream>>>(count, H * W, H, out32.hp, (int)out32.plane(), out32.batch_stride(),
                                                       in32.batch_stride(), out32.base1(), in32.base1(), residual_noise, (double*)dst, scale);
    else
        k_unpack<float><<<grid, blk, 0, ctx->stream>>>(count, H * W, H, out32.hp, (int)out32.plane(), out32.batch_stride(),
                                                      in32.batch_stride(), out32.base1(), in32.base1(), residual_noise, (float*)dst, scale);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// max |x| over a buffer (bit pattern of a non-negative float orders like an unsigned); NaN / Inf give 0x7f800000 or above.
// Calibration only (qmri_set_denoiser), not on the path.
__global__ __launch_bounds__(NT) void k_absmax(const float* __restrict__ x, const float* __restrict__ y, size_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT)
        m = max(m, __float_as_uint(fabsf(y ? x[i] - y[i] : x[i])));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_down((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
int ew_launch_absmax(qmri_ctx* ctx, const float* x, const float* y, size_t n, unsigned* d_out) {
    k_absmax<<<dim3(256), dim3(NT), 0, ctx->stream>>>(x, y, n, d_out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-coil extension of the forward operator (BASELINE.json configs[4]; the reference is single-coil, README.md:63: no counterpart there).
//   k_coil_mul : out[j][c][px] = maps[j0 + j][px] * x[c][px]              (coil sensitivity times image, before the FFT)
//   k_coil_sum : x[c][px] (+)= sum_j conj(maps[j0 + j][px]) * xj[j][c][px]   (coil combination after the inverse FFT; j ascending: one fixed order)
// Streaming, complex fp64 like the operator itself.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_coil_mul(size_t n, size_t plane, int cnt, const double2* __restrict__ x, const double2* __restrict__ maps, double2* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const double2 v = x[i];
    const size_t px = i % plane;
    for (int j = 0; j < cnt; ++j) {
        const double2 c = maps[(size_t)j * plane + px];
        out[(size_t)j * n + i] = make_double2(c.x * v.x - c.y * v.y, c.x * v.y + c.y * v.x);
    }
}
__global__ __launch_bounds__(NT) void k_coil_sum(size_t n, size_t plane, int cnt, const double2* __restrict__ xj, const double2* __restrict__ maps, double2* __restrict__ x, int accumulate) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const size_t px = i % plane;
    double2 a = accumulate ? x[i] : make_double2(0.0, 0.0);
    for (int j = 0; j < cnt; ++j) {
        const double2 c = maps[(size_t)j * plane + px], v = xj[(size_t)j * n + i];
        a.x += c.x * v.x + c.y * v.y;                              // conj(c) * v
        a.y += c.x * v.y - c.y * v.x;
    }
    x[i] = a;
}
int ew_launch_coil_mul(qmri_ctx* ctx, size_t n, size_t plane, int cnt, const double2* x, const double2* maps, double2* out) {
    k_coil_mul<<<dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, ctx->stream>>>(n, plane, cnt, x, maps, out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}
int ew_launch_coil_sum(qmri_ctx* ctx, size_t n, size_t plane, int cnt, const double2* xj, const double2* maps, double2* x, int accumulate) {
    k_coil_sum<<<dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, ctx->stream>>>(n, plane, cnt, xj, maps, x, accumulate);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int ew_launch_real_to_complex(qmri_ctx* ctx, size_t count, const double* in, double2* out) {
    k_real_to_complex<<<dim3((unsigned)((count + NT - 1) / NT)), dim3(NT), 0, ctx->stream>>>(count, in, out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

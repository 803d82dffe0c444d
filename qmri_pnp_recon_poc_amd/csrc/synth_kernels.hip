// synth_kernels.hip -- TSMI synthesis from quantitative maps, the data-preparation step in front of the path
// (main_synthesize_tsmis.m:54,82-100, mode 'real'):
//   Mdl = KDTreeSearcher(dict.lut);  I = knnsearch(Mdl, qm(:,1:2), 'K', 1)       nearest dictionary entry in (T1, T2)
//   X = real(dict.D(I,1:10)) .* dict.normD(I) .* abs(qm(:,3));  X = X .* sign(X(:,:,1))
// The nearest-neighbour search is exhaustive here (K ~ 1e5 entries x 5e4 pixels = 5e9 distance evaluations of fp64 vector
// work): a thread owns four pixels, the look-up table streams through LDS in tiles and is cut into slices over blockIdx.y so
// that ~1000 workgroups exist; strict '<' keeps the first index among equal distances, within a slice and across slices.
// Distances are evaluated in fp64 (the maps are double, dict.lut is widened).
#include <algorithm>
#include <vector>
#include "qmri_internal.h"
#include "dict_device.h"

#pragma clang fp contract(off)      // d1*d1 + d2*d2 unfused, as in the oracle: equal distances must compare equal (first index wins)

namespace {

constexpr int NNT = 256, NN_TILE = 1024, NN_PPT = 4;          // threads, look-up-table entries per LDS tile, pixels per thread
constexpr int NN_KSPLIT_TARGET = 1024;                        // blocks to aim for: the table is cut into slices over blockIdx.y

// Nearest entry of one K slice for NN_PPT pixels per thread: every table entry read from LDS (a broadcast) serves four
// distance evaluations.  Partial results (distance, index) per pixel and slice; k_nn_combine keeps the first minimum.
__global__ __launch_bounds__(NNT) void k_nn_lut(const double* __restrict__ qmap, int Npix, const float* __restrict__ lut, int K, int kslice,
                                                double* __restrict__ pdist, int32_t* __restrict__ pidx) {
    __shared__ double l1[NN_TILE], l2[NN_TILE];
    const int kbeg = blockIdx.y * kslice, kend = min(K, kbeg + kslice);
    double q1[NN_PPT], q2[NN_PPT], best[NN_PPT];
    int bi[NN_PPT];
#pragma unroll
    for (int u = 0; u < NN_PPT; ++u) {
        const int p = (blockIdx.x * NN_PPT + u) * NNT + threadIdx.x;
        q1[u] = (p < Npix) ? qmap[p] : 0.0; q2[u] = (p < Npix) ? qmap[(size_t)Npix + p] : 0.0;
        best[u] = 1.0 / 0.0; bi[u] = kbeg;
    }
    for (int k0 = kbeg; k0 < kend; k0 += NN_TILE) {
        const int cnt = min(NN_TILE, kend - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += NNT) { l1[e] = (double)lut[k0 + e]; l2[e] = (double)lut[(size_t)K + k0 + e]; }
        __syncthreads();
        for (int e = 0; e < cnt; ++e) {
            const double a = l1[e], b = l2[e];
#pragma unroll
            for (int u = 0; u < NN_PPT; ++u) {
                const double d1 = q1[u] - a, d2 = q2[u] - b;
                const double d = d1 * d1 + d2 * d2;
                if (d < best[u]) { best[u] = d; bi[u] = k0 + e; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < NN_PPT; ++u) {
        const int p = (blockIdx.x * NN_PPT + u) * NNT + threadIdx.x;
        if (p < Npix) { pdist[(size_t)blockIdx.y * Npix + p] = best[u]; pidx[(size_t)blockIdx.y * Npix + p] = bi[u]; }
    }
}
// slices are in index order and '<' is strict: the first index among equal distances wins, as in a single pass
__global__ __launch_bounds__(256) void k_nn_combine(const double* __restrict__ pdist, const int32_t* __restrict__ pidx, int Npix, int nslice, int32_t* __restrict__ idx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Npix) return;
    double best = pdist[p];
    int bi = pidx[p];
    for (int sl = 1; sl < nslice; ++sl) {
        const double d = pdist[(size_t)sl * Npix + p];
        if (d < best) { best = d; bi = pidx[(size_t)sl * Npix + p]; }
    }
    idx[p] = bi + 1;                                                      // 1-based, as knnsearch returns it
}

// X(p, c) = D(I, c) * normD(I) * |PD(p)|, then times sign(X(p, 1))  -- single precision, as the reference's arrays are
__global__ __launch_bounds__(256) void k_synth_tsmi(const double* __restrict__ qmap, int Npix, const int32_t* __restrict__ idx, DictView dv,
                                                     const float* __restrict__ normD, int s, float* __restrict__ X) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Npix) return;
    const int a = idx[p] - 1;
    const float nd = normD[a], pd = (float)fabs(qmap[(size_t)2 * Npix + p]);
    auto atom = [&](int c) { return dict_atom(dv, a, c); };               // (qmri_set_dictionary's fragment order: dict_device.h)
    const float x0 = atom(0) * nd * pd;
    const float sg = (x0 > 0.f) ? 1.f : ((x0 < 0.f) ? -1.f : 0.f);        // MATLAB sign(): 0 at 0
    for (int c = 0; c < s; ++c) X[(size_t)c * Npix + p] = atom(c) * nd * pd * sg;
}

// mode 'complex' (main_synthesize_tsmis.m:100-103): X = (D .* normD) .* PD with PD complex, no abs and no sign alignment;
// the output stacks the real parts of the s channels and then their imaginary parts (cat(3, real(X), imag(X)): 2s channels)
__global__ __launch_bounds__(256) void k_synth_tsmi_complex(const double* __restrict__ qmap, const double* __restrict__ pd_imag, int Npix,
                                                             const int32_t* __restrict__ idx, DictView dv,
                                                             const float* __restrict__ normD, int s, float* __restrict__ X) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Npix) return;
    const int a = idx[p] - 1;
    const float nd = normD[a], pr = (float)qmap[(size_t)2 * Npix + p], pi = pd_imag ? (float)pd_imag[p] : 0.f;
    for (int c = 0; c < s; ++c) {
        const float base = dict_atom(dv, a, c) * nd;
        X[(size_t)c * Npix + p] = base * pr;
        X[(size_t)(s + c) * Npix + p] = base * pi;
    }
}

}  // namespace

static int synthesize_impl(qmri_ctx* ctx, const double* qmap, const double* pd_imag, int complex_mode, int Npix, float* X_out, int32_t* idx_out);

extern "C" int qmri_synthesize_tsmi(qmri_ctx* ctx, const double* qmap, int Npix, float* X_out, int32_t* idx_out) {
    return synthesize_impl(ctx, qmap, nullptr, 0, Npix, X_out, idx_out);
}

extern "C" int qmri_synthesize_tsmi_complex(qmri_ctx* ctx, const double* qmap, const double* pd_imag, int Npix, float* X_out, int32_t* idx_out) {
    return synthesize_impl(ctx, qmap, pd_imag, 1, Npix, X_out, idx_out);
}

static int synthesize_impl(qmri_ctx* ctx, const double* qmap, const double* pd_imag, int complex_mode, int Npix, float* X_out, int32_t* idx_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    DictHost& d = ctx->dict;
    if (!d.ready) { qmri_set_error(ctx, "dictionary not set: call qmri_set_dictionary first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, qmap && X_out && Npix > 0, "qmri_synthesize_tsmi: qmap / X_out NULL or Npix <= 0");
    if (d.Q < 2) { qmri_set_error(ctx, "the look-up table needs T1 and T2 columns (Q >= 2)"); return QMRI_ERR_UNSUPPORTED; }
    double* d_q = nullptr; float* d_X = nullptr; int32_t* d_i = nullptr;
    double* d_pd = nullptr; int32_t* d_pi = nullptr; double* d_im = nullptr;
    const int nch = complex_mode ? 2 * d.s : d.s;                   // output channels
    const int nbx = (Npix + NNT * NN_PPT - 1) / (NNT * NN_PPT);
    int nslice = std::max(1, std::min(NN_KSPLIT_TARGET / std::max(nbx, 1), (d.K + NN_TILE - 1) / NN_TILE));
    const int kslice = (((d.K + nslice - 1) / nslice) + NN_TILE - 1) / NN_TILE * NN_TILE;     // whole tiles per slice
    nslice = (d.K + kslice - 1) / kslice;
    int st = QMRI_OK;
    auto fail = [&](const char* what) { qmri_set_error(ctx, "qmri_synthesize_tsmi: %s", what); st = QMRI_ERR_HIP; };
    do {
        if (hipMalloc((void**)&d_q, (size_t)3 * Npix * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_X, (size_t)nch * Npix * sizeof(float)) != hipSuccess ||
            (pd_imag && hipMalloc((void**)&d_im, (size_t)Npix * sizeof(double)) != hipSuccess) ||
            hipMalloc((void**)&d_i, (size_t)Npix * sizeof(int32_t)) != hipSuccess || hipMalloc((void**)&d_pd, (size_t)nslice * Npix * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&d_pi, (size_t)nslice * Npix * sizeof(int32_t)) != hipSuccess) { fail("hipMalloc"); st = QMRI_ERR_NOMEM; break; }
        if (hipMemcpyAsync(d_q, qmap, (size_t)3 * Npix * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
        if (pd_imag && hipMemcpyAsync(d_im, pd_imag, (size_t)Npix * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
        k_nn_lut<<<dim3(nbx, nslice), dim3(NNT), 0, ctx->stream>>>(d_q, Npix, d.d_lut, d.K, kslice, d_pd, d_pi);
        k_nn_combine<<<dim3((Npix + 255) / 256), dim3(256), 0, ctx->stream>>>(d_pd, d_pi, Npix, nslice, d_i);
        const DictView dv = {d.d_pack, d.wide, ((d.s + 1) / 2 <= 4) ? 4 : 8, d.G8};
        if (complex_mode) k_synth_tsmi_complex<<<dim3((Npix + 255) / 256), dim3(256), 0, ctx->stream>>>(d_q, d_im, Npix, d_i, dv, d.d_normD, d.s, d_X);
        else k_synth_tsmi<<<dim3((Npix + 255) / 256), dim3(256), 0, ctx->stream>>>(d_q, Npix, d_i, dv, d.d_normD, d.s, d_X);
        if (hipGetLastError() != hipSuccess) { fail("kernel launch"); break; }
        if (hipMemcpyAsync(X_out, d_X, (size_t)nch * Npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (idx_out && hipMemcpyAsync(idx_out, d_i, (size_t)Npix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { fail("synchronize"); break; }
    } while (0);
    if (d_q) (void)hipFree(d_q);
    if (d_X) (void)hipFree(d_X);
    if (d_i) (void)hipFree(d_i);
    if (d_pd) (void)hipFree(d_pd);
    if (d_pi) (void)hipFree(d_pi);
    if (d_im) (void)hipFree(d_im);
    return st;
}

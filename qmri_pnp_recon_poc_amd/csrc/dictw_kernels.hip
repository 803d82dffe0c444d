// dictw_kernels.hip -- MRF dictionary template match for WIDE dictionaries, 16 < s <= 1024 channels (uncompressed fingerprints, s = T): the
// (atoms x pixels) product as a channel-blocked single-precision GEMM on the matrix cores (gfx950) with max(abs(ip)) fused behind it.
//
// Reference semantics: main_files/dictionary_matching/mrf_dtm_cpu.m -- T-generic (:41-50: [ix,iy,T] = size(data.X), x = reshape(data.X,[N,T]))
//   :54      x = single(x)
//   :91      ip = dict.D * ctranspose(x(cind,:))      (K x T) * (T x B): ip(j,p) = sum_c D(j,c) conj(x(p,c))
//   :92      [mt,dm] = max(abs(ip),[],1)               first index wins ties
//   :94-96   pd = ip(dm) / normD(dm)
// At T = 1000, K = 98 304 and 224 x 224 pixels that is 19.7 TFLOP per slice -- the one place on the path where the product is a compute-bound
// GEMM (the compressed match, s = 10, keeps a pixel tile's X in registers: dict_kernels.hip).  K x Npix is never materialised (the reference
// bounds it by blocks of 1e9 elements, :74): a workgroup owns 128 pixels and walks 128-atom tiles; per tile the channels stream through LDS in
// stages of 16 and the real and imaginary chains accumulate in v_mfma_f32_32x32x2_f32 in ASCENDING channel order -- an f32 MFMA is a k-ordered
// chain of fmaf, so ip carries the bits of the oracle's sequential fmaf chain over c = 0 .. s-1 (zero padding adds fma(0, 0, acc) = acc) -- and
// the finished 32 x 32 tiles go through the same incumbent rule as the narrow kernels (inc_update, dict_device.h).  Atoms are split into P
// parts over workgroups; k_dict_merge picks per pixel by the same rule (larger magnitude, then lower index).
//
// Layouts (fragment order, so that both the global -> LDS copy and the LDS -> register reads are plain 16-byte-per-lane contiguous moves;
// a wave's ds_read_b128 covers 1 KB in lane order: conflict-free by the lane-group table of MI355X_MICROARCH.md):
//   D:  pack[tile32][G8][lane][4]           lane = (atom & 31) + 32 h, entry i = D(atom, 8 g + 2 i + h)     (A operand: k = 2 q + h)
//   X:  xp[tile32][G8][re | -im][lane][4]   lane = (pixel & 31) + 32 h, entry i = single(x(pixel, 8 g + 2 i + h))   (B operand; -im: conj)
// Bound: f32 MFMA, 157.3 TFLOP/s.  Operand traffic is small against it: a 128 x 128 tile needs 24 KB per 524 288 multiply-adds (13 GB/s per
// workgroup); the workgroup order keeps the 64 workgroups an XCD holds on a 4 x 16 block of (pixel tile, atom part), so its L2 serves both operands.
#include <algorithm>
#include "qmri_internal.h"
#include "dict_device.h"

namespace {

constexpr int WT = 256;          // threads per workgroup: 4 waves = 2 (atom halves) x 2 (pixel halves) of a 128 x 128 tile
constexpr int STAGE_F4 = 1536;   // float4 per stage: A 4 x 2 x 64 + B 4 x 2 x 2 x 64

// D (K x s column-major singles, device) -> A-fragment order
__global__ __launch_bounds__(256) void k_dictw_pack_d(const float* __restrict__ D, int K, int s, int G8, int ntile32, float4* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)ntile32 * G8 * 64) return;
    const int lane = (int)(idx & 63), g = (int)((idx >> 6) % G8), ab = (int)((idx >> 6) / G8);
    const int atom = ab * 32 + (lane & 31), h = lane >> 5;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = g * 8 + 2 * i + h;
        v[i] = (atom < K && c < s) ? D[(size_t)atom + (size_t)K * c] : 0.f;
    }
    out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// X (Npix x s complex doubles column-major) -> single(x) (mrf_dtm_cpu.m:54) in B-fragment order, real parts and NEGATED imaginary parts (conj, :91)
__global__ __launch_bounds__(256) void k_dictw_pack_x(const double2* __restrict__ X, int Npix, int s, int G8, int ntile32, float4* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)ntile32 * G8 * 64) return;
    const int lane = (int)(idx & 63), g = (int)((idx >> 6) % G8), pb = (int)((idx >> 6) / G8);
    const int p = pb * 32 + (lane & 31), h = lane >> 5;
    float re[4], im[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = g * 8 + 2 * i + h;
        double2 v = make_double2(0.0, 0.0);
        if (p < Npix && c < s) v = X[(size_t)p + (size_t)Npix * c];
        re[i] = (float)v.x; im[i] = -(float)v.y;
    }
    const size_t o = (((size_t)pb * G8 + g) * 2) * 64 + lane;
    out[o] = make_float4(re[0], re[1], re[2], re[3]);
    out[o + 64] = make_float4(im[0], im[1], im[2], im[3]);
}

// Workgroup (pixel tile pt, atom part prt): atoms tiles [prt tper, ...) of 128, each over all channel stages.
// One barrier per stage: while a stage is multiplied out of one LDS buffer the next one travels global -> registers -> the other buffer.
__global__ __launch_bounds__(WT, 2) void k_dictw_match(const float4* __restrict__ DP, const float4* __restrict__ XP, int G8, int AT, int PT, int P,
                                                        int tper, int Npix, float4* __restrict__ part, int lsp) {
    __shared__ float4 s_buf[2][STAGE_F4];
    // workgroup order: blockIdx.x % 8 is the XCD under round-robin placement (speed only); an XCD walks super-tiles of 2^lsp pixel tiles x
    // 2^(6 - lsp) atom parts (64 workgroups = what it holds at once)
    const int id = blockIdx.x, xcd = id & 7, kk = id >> 3;
    const int lsa = 6 - lsp;
    const int SPT = (PT + (1 << lsp) - 1) >> lsp, SPP = (P + (1 << lsa) - 1) >> lsa;
    const int sidx = xcd + 8 * (kk >> 6), within = kk & 63;
    if (sidx >= SPT * SPP) return;
    const int pt = ((sidx % SPT) << lsp) + (within & ((1 << lsp) - 1)), prt = ((sidx / SPT) << lsa) + (within >> lsp);
    if (pt >= PT || prt >= P) return;
    const int t0 = prt * tper, nt = min(tper, AT - t0);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wa = wave & 1, wp = wave >> 1, h = lane >> 5, j = lane & 31;
    const int nstage = G8 >> 1;

    // (the next stage's operands travel in six named registers quadruples: with arrays captured by lambdas hipcc kept them in scratch memory)
    float4 ga0, ga1, gb0, gb1, gb2, gb3;
    const float4* dp_base = DP + (size_t)(tid >> 7) * G8 * 64 + (tid & 127);          // + (t 4 G8 + 2 st) 64  [+ 2 G8 64 for the second request]
    const float4* xp_base = XP + (size_t)pt * 4 * G8 * 128 + tid;                      // + (r G8 + 2 st) 128
#define DW_GLOAD(t_, st_)                                                                 \
    do {                                                                                  \
        const float4* dq = dp_base + ((size_t)(t_) * 4 * G8 + 2 * (st_)) * 64;            \
        ga0 = dq[0]; ga1 = dq[(size_t)2 * G8 * 64];                                       \
        const float4* xq = xp_base + (size_t)(2 * (st_)) * 128;                           \
        gb0 = xq[0]; gb1 = xq[(size_t)G8 * 128]; gb2 = xq[(size_t)2 * G8 * 128]; gb3 = xq[(size_t)3 * G8 * 128]; \
    } while (0)
#define DW_SSTORE(b_)                                                                     \
    do {                                                                                  \
        float4* sb = s_buf[b_];                                                           \
        sb[tid] = ga0; sb[tid + 256] = ga1;                                               \
        sb[512 + tid] = gb0; sb[768 + tid] = gb1; sb[1024 + tid] = gb2; sb[1280 + tid] = gb3; \
    } while (0)

    f32x16 cre[2][2], cim[2][2];
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) { cre[ma][nb] = f32x16{0}; cim[ma][nb] = f32x16{0}; }
    Inc I[2] = {{-1.0f, -1.0f, 0.f, 0.f, 0}, {-1.0f, -1.0f, 0.f, 0.f, 0}};

    int t = t0, st = 0, cur = 0;
    DW_GLOAD(t, st);
    DW_SSTORE(0);
    __syncthreads();
    const int nsteps = nt * nstage;
    for (int step = 0; step < nsteps; ++step) {
        int tn = t, stn = st + 1;
        if (stn == nstage) { stn = 0; ++tn; }
        const bool more = step + 1 < nsteps;
        if (more) DW_GLOAD(tn, stn);
        const float4* sA = s_buf[cur];
        const float4* sB = s_buf[cur] + 512;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 a[2], br[2], bi[2];
#pragma unroll
            for (int ma = 0; ma < 2; ++ma) a[ma] = __builtin_bit_cast(f32x4, sA[((2 * wa + ma) * 2 + g) * 64 + lane]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                br[nb] = __builtin_bit_cast(f32x4, sB[(((2 * wp + nb) * 2 + g) * 2 + 0) * 64 + lane]);
                bi[nb] = __builtin_bit_cast(f32x4, sB[(((2 * wp + nb) * 2 + g) * 2 + 1) * 64 + lane]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)                                   // channel pairs in ascending order: the chain of the oracle
#pragma unroll
                for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        cre[ma][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ma][i], br[nb][i], cre[ma][nb], 0, 0, 0);
                        cim[ma][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ma][i], bi[nb][i], cim[ma][nb], 0, 0, 0);
                    }
        }
        if (st == nstage - 1) {                                           // an atom tile is complete: ascending 32-atom tiles (inc_update's note on magnitude 0)
#pragma unroll
            for (int ma = 0; ma < 2; ++ma)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    inc_update(t * 4 + 2 * wa + ma, h, cre[ma][nb], cim[ma][nb], I[nb]);
                    cre[ma][nb] = f32x16{0}; cim[ma][nb] = f32x16{0};
                }
        }
        if (more) DW_SSTORE(cur ^ 1);
        __syncthreads();
        cur ^= 1; t = tn; st = stn;
    }
    // merge the two lane halves (same pixel, interleaved atom rows), then the two atom halves of the tile (waves wa = 0, 1) through LDS
    float4* s_c = s_buf[0];                                               // [wp][nb][j]
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float ob = __shfl(I[nb].best, lane ^ 32, 64), ore = __shfl(I[nb].cre, lane ^ 32, 64), oim = __shfl(I[nb].cim, lane ^ 32, 64);
        const int oi = __shfl(I[nb].bidx, lane ^ 32, 64);
        if (cand_better(ob, oi, I[nb].best, I[nb].bidx)) { I[nb].best = ob; I[nb].bidx = oi; I[nb].cre = ore; I[nb].cim = oim; }
        if (wa == 1 && h == 0) s_c[(wp * 2 + nb) * 32 + j] = make_float4(I[nb].best, __int_as_float(I[nb].bidx), I[nb].cre, I[nb].cim);
    }
    __syncthreads();
    if (wa == 0 && h == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int p = pt * 128 + (2 * wp + nb) * 32 + j;
            const float4 o = s_c[(wp * 2 + nb) * 32 + j];
            float4 b = make_float4(I[nb].best, __int_as_float(I[nb].bidx), I[nb].cre, I[nb].cim);
            if (cand_better(o.x, __float_as_int(o.y), b.x, __float_as_int(b.y))) b = o;
            if (p < Npix) part[(size_t)prt * Npix + p] = b;
        }
    }
}

#undef DW_GLOAD
#undef DW_SSTORE

}  // namespace

// qmri_set_dictionary for s > 16: D (host, K x s column-major) -> ctx->dict.d_pack in A-fragment order, atoms padded to whole 128-atom tiles
int dictw_pack_dictionary(qmri_ctx* ctx, const float* D_host, int K, int s) {
    DictHost& d = ctx->dict;
    const int spad = (s + 15) / 16 * 16;
    d.G8 = spad / 8;
    d.ntiles = (K + 127) / 128 * 4;                                      // 32-atom tiles
    float* raw = nullptr;
    const size_t nraw = (size_t)K * s, npack = (size_t)d.ntiles * d.G8 * 64 * 4;
    QMRI_HIP(ctx, hipMalloc((void**)&raw, nraw * sizeof(float)));
    int st = QMRI_OK;
    do {
        if (hipMalloc((void**)&d.d_pack, npack * sizeof(float)) != hipSuccess) { qmri_set_error(ctx, "hipMalloc of the packed dictionary failed"); st = QMRI_ERR_NOMEM; break; }
        if (hipMemcpyAsync(raw, D_host, nraw * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { qmri_set_error(ctx, "H2D copy of D failed"); st = QMRI_ERR_HIP; break; }
        const size_t n4 = npack / 4;
        k_dictw_pack_d<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, ctx->stream>>>(raw, K, s, d.G8, d.ntiles, (float4*)d.d_pack);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { qmri_set_error(ctx, "packing the dictionary failed"); st = QMRI_ERR_HIP; break; }
    } while (0);
    (void)hipFree(raw);
    return st;
}

int dictw_launch(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win) {
    DictHost& D = ctx->dict;
    if (!D.slots_w) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        QMRI_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        QMRI_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dictw_match, WT, 0));
        D.slots_w = std::max(1, per_cu) * prop.multiProcessorCount;
    }
    const int PT = (Npix + 127) / 128, AT = D.ntiles / 4;
    // atom parts: ~20 rounds of the device's resident workgroups (a ragged last round then costs < 5 %), whole groups of 16 for the XCD super-tiles
    int P = std::max(1, std::min(AT, (20 * D.slots_w + PT - 1) / PT));
    if (P < AT) P = std::min(AT, (P + 15) / 16 * 16);
    const int tper = (AT + P - 1) / P;
    P = (AT + tper - 1) / tper;                                          // no empty part
    const size_t nxp = (size_t)PT * 4 * D.G8 * 2 * 64;                   // float4
    QMRI_TRY(dict_scratch(ctx, (void**)&D.d_xp, &D.xp_cap, nxp * sizeof(float4)));
    QMRI_TRY(dict_scratch(ctx, (void**)&D.d_part, &D.part_cap, (size_t)P * Npix * sizeof(float4)));
    const size_t nthr = (size_t)PT * 4 * D.G8 * 64;
    k_dictw_pack_x<<<dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_X, Npix, D.s, D.G8, PT * 4, (float4*)D.d_xp);
    QMRI_HIP(ctx, hipGetLastError());
    // shape of the XCD super-tile (QMRI_DICTW_LSP for A/Bs): 4 pixel tiles x 16 atom parts.  Measured at s = 1000, K = 98 304 (profiles/r04_i_*): the
    // launch time does not depend on it (144 - 147 ms: the kernel is compute-bound), the bytes leaving the L2s do -- FETCH_SIZE as counted 167 GB (64 x 1),
    // 124 GB (8 x 8), 50 GB (4 x 16), 55 GB (2 x 32): X tiles (8 bytes per pixel and channel) are the larger operand, so more parts per pixel tile pay.
    const int lsp_env = qmri_knob(K_DICTW_LSP);
    const int lsp = std::max(0, std::min(6, lsp_env)), lsa = 6 - lsp;
    const int nsuper = ((PT + (1 << lsp) - 1) >> lsp) * ((P + (1 << lsa) - 1) >> lsa);
    const unsigned grid = 8u * 64u * (unsigned)((nsuper + 7) / 8);
    k_dictw_match<<<dim3(grid), dim3(WT), 0, ctx->stream>>>((const float4*)D.d_pack, (const float4*)D.d_xp, D.G8, AT, PT, P, tper, Npix, D.d_part, lsp);
    QMRI_HIP(ctx, hipGetLastError());
    return dict_launch_merge(ctx, D.d_part, P, Npix, d_qmap, d_pd, d_mt, d_dm, win);
}

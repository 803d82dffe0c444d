// conv_kernels.hip -- the convolution engine of the denoiser proximal (gfx950, f32 MFMA).
//
// Reference semantics: every layer of UNetRes (PyTorch_Denoiser/zhang_dpir_testing_code/network_unet.py:68-117):
//   conv 3x3, stride 1, zero pad 1, bias-free          basicblock.py:61-98        (head, tail, 56 ResBlock layers)
//   conv 2x2, stride 2 ("strideconv")                   basicblock.py:437-443      (3 down-samplers)
//   transposed conv 2x2, stride 2 ("convtranspose")     basicblock.py:413-419      (3 up-samplers)
//   ResBlock  x + conv(relu(conv(x)))                   basicblock.py:211-223
// run in fp32 as denoiseImage_PnP_ADMM.m:72-77,88 does.  All three are one implicit GEMM
//   D[cout][pixel] = sum_{ci,tap} Wt[cout][ci][tap] * in[ci][pixel*S + tap - pad]
// on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
//
// Workgroup = 4 waves = one 32*MT (cout) x 32 (pixels: 4 along w x 8 along h) output tile.  The four waves
// split K (input channels) of every CC-channel chunk four ways and are summed in a fixed order through LDS
// at the end (deterministic).  Per chunk: the zero-padded input tile is staged global -> registers -> LDS
// (the next chunk's loads are issued before the current chunk's MFMAs), B operands are ds_read_b32 from that
// tile (lanes 0-31 read channel 2p, lanes 32-63 channel 2p+1 of pair p -- the two k of a 32x32x2 step),
// A operands (weights) come straight from global memory, pre-packed at load time so that one 16-byte load per
// lane feeds four MFMA steps.  Epilogue fusions: + residual, + skip tensor, ReLU.
//
// Tensors are [B][C][W][H] fp32 with h fastest (the memory order of a MATLAB H x W x C array), so a MATLAB
// buffer is consumed without a transpose; kh pairs with h and kw with w.
#include "qmri_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;
constexpr int JW = 4, JH = 8;          // output pixel tile: 4 (w) x 8 (h) = the 32 columns of the MFMA tile

template <int KIND> struct Geo;
template <> struct Geo<CONV_3X3>  { static constexpr int TH = 3, TW = 3, S = 1, PAD = 1, CC = 64; };
template <> struct Geo<CONV_DOWN> { static constexpr int TH = 2, TW = 2, S = 2, PAD = 0, CC = 64; };
template <> struct Geo<CONV_UP>   { static constexpr int TH = 1, TW = 1, S = 1, PAD = 0, CC = 256; };

template <int KIND, int MT>
__global__ __launch_bounds__(NT) void k_conv(const float* __restrict__ in, const float4* __restrict__ wp, float* out,
                                              const float* add1, const float* add2, int Cin, int Cout, int H, int W,
                                              int nch, int tiles_h, int relu_out) {
    typedef Geo<KIND> G_;
    constexpr int TH = G_::TH, TW = G_::TW, S = G_::S, PAD = G_::PAD, CC = G_::CC;
    constexpr int IW = (JW - 1) * S + TW, IH = (JH - 1) * S + TH, PL = IW * IH;
    constexpr int NTAP = TH * TW, G = CC / 32, NSTG = CC * PL / NT;
    constexpr int LDSF = (CC * PL > 4 * MT * 1024) ? CC * PL : 4 * MT * 1024;
    static_assert((CC * PL) % NT == 0, "staging loop must divide evenly");
    __shared__ float lds[LDSF];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int wx = j >> 3, hy = j & 7;
    const int ct = blockIdx.x, pt = blockIdx.y, b = blockIdx.z;
    const int ow0 = (pt / tiles_h) * JW, oh0 = (pt % tiles_h) * JH;
    const int gw0 = ow0 * S - PAD, gh0 = oh0 * S - PAD;
    const float* inb = in + (size_t)b * Cin * W * H;

    // per-thread staging coordinates (fixed across chunks): element f = tid + NT*q of the [CC][IW][IH] tile
    float stg[NSTG];
    auto stage_load = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < NSTG; ++q) {
            const int f = tid + NT * q;
            const int cl = f / PL, rem = f - cl * PL;
            const int ix = rem / IH, iy = rem - ix * IH;
            const int ci = chunk * CC + cl, gw = gw0 + ix, gh = gh0 + iy;
            float v = 0.f;
            if (ci < Cin && gw >= 0 && gw < W && gh >= 0 && gh < H) v = inb[((size_t)ci * W + gw) * H + gh];
            stg[q] = v;
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x16){0};

    const int b_base = (wave * (CC / 4) + h) * PL + (wx * S) * IH + hy * S;
    stage_load(0);
    for (int chunk = 0; chunk < nch; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NSTG; ++q) lds[tid + NT * q] = stg[q];
        __syncthreads();
        if (chunk + 1 < nch) stage_load(chunk + 1);
        const float4* wq[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            wq[mt] = wp + ((((size_t)(ct * MT + mt) * nch + chunk) * 4 + wave) * (NTAP * G)) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int kh = t / TW, kw = t - kh * TW;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = wq[mt][(t * G + g) * 64];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float bv = lds[b_base + 2 * (4 * g + jj) * PL + kw * IH + kh];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = (jj == 0) ? a[mt].x : (jj == 1) ? a[mt].y : (jj == 2) ? a[mt].z : a[mt].w;
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // fixed-order cross-wave reduction through LDS, then the fused epilogue
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[(wave * MT + mt) * 1024 + r * 64 + lane] = acc[mt][r];
    __syncthreads();
    const int OW = (KIND == CONV_DOWN) ? W / 2 : W, OH = (KIND == CONV_DOWN) ? H / 2 : H;
#pragma unroll
    for (int q = 0; q < MT * 4; ++q) {
        const int f = tid + NT * q;
        const int mt = f >> 10, r = (f >> 6) & 15, l = f & 63;
        const float sum = ((lds[(0 * MT + mt) * 1024 + (f & 1023)] + lds[(1 * MT + mt) * 1024 + (f & 1023)]) +
                           lds[(2 * MT + mt) * 1024 + (f & 1023)]) + lds[(3 * MT + mt) * 1024 + (f & 1023)];
        const int v = (ct * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const int ow = ow0 + ((l & 31) >> 3), oh = oh0 + (l & 7);
        if (ow >= OW || oh >= OH) continue;
        if (KIND == CONV_UP) {
            const int kk = v / Cout, o = v - kk * Cout;
            if (kk >= 4) continue;
            const size_t idx = (((size_t)b * Cout + o) * (2 * W) + 2 * ow + (kk & 1)) * (2 * H) + 2 * oh + (kk >> 1);
            out[idx] = sum;
        } else {
            if (v >= Cout) continue;
            const size_t idx = (((size_t)b * Cout + v) * OW + ow) * OH + oh;
            float val = sum;
            if (add1) val = add1[idx] + val;
            if (add2) val = val + add2[idx];
            if (relu_out) val = fmaxf(val, 0.f);
            out[idx] = val;
        }
    }
}

template <int KIND>
int launch_kind(qmri_ctx* ctx, const ConvLayer& L, int B, int H, int W, const float* in, float* out, const float* add1,
                const float* add2, int relu_out) {
    typedef Geo<KIND> G_;
    const int OW = (KIND == CONV_DOWN) ? W / 2 : W, OH = (KIND == CONV_DOWN) ? H / 2 : H;
    const int tiles_w = (OW + JW - 1) / JW, tiles_h = (OH + JH - 1) / JH;
    const int nch = L.cin_pad / G_::CC;
    const int n32 = L.n_ct;                      // number of 32-row cout tiles in the packed weights
    // two MFMA row tiles per wave only when that still leaves enough workgroups to fill 256 CUs
    const long wg2 = (long)(n32 / 2) * tiles_w * tiles_h * B;
    const int MT = (n32 % 2 == 0 && wg2 >= 1024) ? 2 : 1;
    dim3 grid(n32 / MT, tiles_w * tiles_h, B), blk(NT);
    const float4* wp = reinterpret_cast<const float4*>(L.wp);
    if (MT == 2)
        k_conv<KIND, 2><<<grid, blk, 0, ctx->stream>>>(in, wp, out, add1, add2, L.Cin, L.Cout, H, W, nch, tiles_h, relu_out);
    else
        k_conv<KIND, 1><<<grid, blk, 0, ctx->stream>>>(in, wp, out, add1, add2, L.Cin, L.Cout, H, W, nch, tiles_h, relu_out);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

}  // namespace

void conv_plan_layer(ConvLayer& L, ConvKind kind, int Cin, int Cout) {
    L.kind = kind; L.Cin = Cin; L.Cout = Cout;
    const int CC = (kind == CONV_UP) ? Geo<CONV_UP>::CC : 64;
    L.cin_pad = ((Cin + CC - 1) / CC) * CC;
    const int rows = (kind == CONV_UP) ? 4 * Cout : Cout;
    L.n_ct = (rows + 31) / 32;
    L.MT = 1;
    L.wp = nullptr; L.wp_floats = 0;
}

// Pack PyTorch-layout weights (Conv2d OIHW, ConvTranspose2d IOHW) into MFMA A-fragment order:
//   float4 index = ((((ct32*nch + chunk)*4 + wave)*NTAP + t)*G + g)*64 + lane ; component jj
//   row = ct32*32 + (lane&31),  ci = chunk*CC + wave*(CC/4) + 2*(4g+jj) + (lane>>5),  tap t = kh*TW + kw
size_t conv_pack_weights(const ConvLayer& L, const float* w, std::vector<float>& packed) {
    const int TH = (L.kind == CONV_3X3) ? 3 : (L.kind == CONV_DOWN) ? 2 : 1, TW = TH;
    const int CC = (L.kind == CONV_UP) ? Geo<CONV_UP>::CC : 64;
    const int NTAP = TH * TW, G = CC / 32, nch = L.cin_pad / CC;
    const size_t total = (size_t)L.n_ct * nch * 4 * NTAP * G * 64 * 4;
    packed.assign(total, 0.f);
    for (int ct = 0; ct < L.n_ct; ++ct)
        for (int chunk = 0; chunk < nch; ++chunk)
            for (int wave = 0; wave < 4; ++wave)
                for (int t = 0; t < NTAP; ++t)
                    for (int g = 0; g < G; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int jj = 0; jj < 4; ++jj) {
                                const int row = ct * 32 + (lane & 31);
                                const int ci = chunk * CC + wave * (CC / 4) + 2 * (4 * g + jj) + (lane >> 5);
                                if (ci >= L.Cin) continue;
                                float v;
                                if (L.kind == CONV_UP) {
                                    const int kk = row / L.Cout, o = row - kk * L.Cout;     // kk = kh*2 + kw
                                    if (kk >= 4) continue;
                                    v = w[((size_t)ci * L.Cout + o) * 4 + kk];              // IOHW
                                } else {
                                    if (row >= L.Cout) continue;
                                    v = w[((size_t)row * L.Cin + ci) * NTAP + t];           // OIHW, t = kh*TW + kw
                                }
                                packed[((((((size_t)ct * nch + chunk) * 4 + wave) * NTAP + t) * G + g) * 64 + lane) * 4 + jj] = v;
                            }
    return total;
}

int conv_launch(qmri_ctx* ctx, const ConvLayer& L, int B, int H, int W, const float* in, float* out, const float* add1,
                const float* add2, int relu_out) {
    switch (L.kind) {
        case CONV_3X3: return launch_kind<CONV_3X3>(ctx, L, B, H, W, in, out, add1, add2, relu_out);
        case CONV_DOWN: return launch_kind<CONV_DOWN>(ctx, L, B, H, W, in, out, add1, add2, relu_out);
        default: return launch_kind<CONV_UP>(ctx, L, B, H, W, in, out, add1, add2, relu_out);
    }
}

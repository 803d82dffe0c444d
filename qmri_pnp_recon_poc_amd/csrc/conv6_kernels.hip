// conv6_kernels.hip -- 3x3 convolutions of the UNetRes / DRUNet denoiser on the bf16 matrix cores with fp32-level accuracy.
//
// Reference semantics: denoiseImage_PnP_ADMM.m:1-117 runs the network in single precision; layers as in
// oracle/orc_net.c (Conv2d 3x3, stride 1, pad 1, no bias; optional ReLU; residual adds).
//
// Method ("bf16 x 6"): every fp32 operand is split exactly into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 mantissa
// bits; the residuals x - x0 and x - x0 - x1 are exact in fp32).  A product w*a then expands into nine bf16 x bf16
// products, each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the six of order >= 2^-16
//     w0 a0 + (w0 a1 + w1 a0) + (w1 a1 + w0 a2 + w2 a0)
// are accumulated, the three dropped ones are of order 2^-24 and below, i.e. at the rounding level of an fp32 multiply.
// The result differs from an fp32 FMA chain by accumulation rounding only (tools/bf16x6_check.py: 1e-7 relative, the
// same as between two fp32 summation orders).  The bf16 MFMA is 16x the rate of v_mfma_f32_32x32x2_f32 (32 cycles for
// 32x32x16 vs 64 for 32x32x2), so six of them per fp32-equivalent step are still 2.7x faster than the fp32 matrix path.
//
// Implicit GEMM per workgroup: 64 output channels x (TH x TW) pixels, K = Cin*9 walked in chunks of 16 channels x 3 taps.
//   waves 0-3  MFMA: per tap and cout tile 3 A fragments (weights, pre-split and pre-ordered on the host) and 3 B
//              fragments (activations) from LDS feed 6 MFMAs; one accumulation chain per 32x32 tile
//   waves 4-7  loaders: weights global -> LDS (plain copy), activations global fp32 planes -> split -> LDS [pixel][8 ch],
//              requested two steps ahead and kept in registers for one
// Tensors stay fp32 padded planes in HBM (qmri_internal.h PTensor), so this kernel is interchangeable with k_conv.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "qmri_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (register arrays of HIP's uint4 struct are not promoted out of scratch)

constexpr int NT6 = 512;         // threads per workgroup: 4 MFMA waves + 4 loader waves
constexpr int NLD6 = 256;        // loader threads
constexpr int CK = 16;           // input channels per chunk = K of one MFMA
constexpr int AST = 3 * 2 * 3 * 64;   // uint4 per step of A: 3 taps x 2 cout tiles x 3 splits x 64 lanes

struct Conv6Args {
    const float* in; const uint4* wp; float* out; const float* add1; const float* add2;
    int Cout, W, H;
    int in_hp, in_plane; long in_bs;          // padded row pitch, plane size, batch stride (elements)
    int out_hp, out_plane; long out_bs, add1_bs, add2_bs;
    int nchunk, n_ct, tiles_h, tiles_w, relu_out;
};

template <int CFG> struct Cfg6;
template <> struct Cfg6<0> { static constexpr int TH = 16, TW = 8, MW = 2; };   // 64 cout x 128 px: waves = 2 x 2 pixel blocks, 64 cout each
template <> struct Cfg6<1> { static constexpr int TH = 8, TW = 8, MW = 1; };    // 64 cout x 64 px: waves = 2 cout halves x 2 pixel blocks

__device__ __forceinline__ void lds_barrier6() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// x = x0 + x1 + x2 exactly (bf16 pieces); two values packed per dword, low half = first
__device__ __forceinline__ void split_pair(float xa, float xb, unsigned& p0, unsigned& p1, unsigned& p2) {
    const __bf16 a0 = (__bf16)xa, b0 = (__bf16)xb;
    const float ra = xa - (float)a0, rb = xb - (float)b0;
    const __bf16 a1 = (__bf16)ra, b1 = (__bf16)rb;
    const float sa = ra - (float)a1, sb = rb - (float)b1;
    const __bf16 a2 = (__bf16)sa, b2 = (__bf16)sb;
    p0 = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, b0) << 16);
    p1 = (unsigned)__builtin_bit_cast(unsigned short, a1) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
    p2 = (unsigned)__builtin_bit_cast(unsigned short, a2) | ((unsigned)__builtin_bit_cast(unsigned short, b2) << 16);
}

template <int CFG>
__global__ __launch_bounds__(NT6) void k_conv6(const Conv6Args A) {
    typedef Cfg6<CFG> C;
    constexpr int TH = C::TH, TW = C::TW, MW = C::MW;
    constexpr int IH = TH + 2, IW = TW + 2, NPX = IH * IW;         // input tile with halo
    constexpr int NBI = 2 * NPX;                                    // loader items of one chunk of B: (k-half, pixel)
    constexpr int NBQ = (NBI + 3 * NLD6 - 1) / (3 * NLD6);          // ... per loader thread and step (a chunk is spread over its 3 steps)
    constexpr int NAQ = (AST + NLD6 - 1) / NLD6;                    // uint4 of A per loader thread and step
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;                                     // [2][AST]
    uint4* Bbuf = Abuf + 2 * AST;                                   // [2][3 splits][2 k-halves][NPX]  (8 channels = 16 B per entry)

    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ct = bid % A.n_ct; bid /= A.n_ct;
    const int th = bid % A.tiles_h; bid /= A.tiles_h;
    const int tw = bid % A.tiles_w;
    const int b = bid / A.tiles_w;
    const int oh0 = th * TH, ow0 = tw * TW;
    const int nsteps = 3 * A.nchunk;

    if (tid >= NT6 - NLD6) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - (NT6 - NLD6);
        const uint4* wsrc = A.wp + (size_t)ct * A.nchunk * 3 * AST;               // steps of this cout tile are contiguous
        const float* isrc = A.in + (size_t)b * A.in_bs + (size_t)ow0 * A.in_hp + oh0;   // halo origin = padded (oh0, ow0)
        u32x4 ra0[NAQ], ra1[NAQ];
        float rb0[NBQ][8], rb1[NBQ][8];
        // Schedule.  Barrier g precedes compute step g.  Abuf[(g+1)&1] is free once step g-1 is over, i.e. after barrier g:
        // iteration g (between barriers g and g+1) stores A(g+1).  Bbuf[(c+1)&1] is free once chunk c-1 is over, i.e. after
        // barrier 3c: iterations 3c, 3c+1, 3c+2 store the three parts of B(c+1).  What an iteration stores was requested
        // one iteration earlier into the other register set, so a request has a whole step to arrive.
        // (Requests past the end are clamped, not skipped: branch-free code lets the compiler count vmcnt exactly.)
#define LOAD_A(g_, ra_)                                                                                          \
        {                                                                                                        \
            const uint4* ws = wsrc + (size_t)(((g_) < nsteps) ? (g_) : nsteps - 1) * AST;                        \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) { const int i = lt + NLD6 * q; ra_[q] = __builtin_bit_cast(u32x4, ws[(i < AST) ? i : 0]); } \
        }
#define STORE_A(g_, ra_)                                                                                         \
        if ((g_) < nsteps) {                                                                                     \
            uint4* ad = Abuf + ((g_) & 1) * AST;                                                                 \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) { const int i = lt + NLD6 * q; if (i < AST) ad[i] = __builtin_bit_cast(uint4, ra_[q]); } \
        }
#define LOAD_B(c_, part_, rb_)                                                                                   \
        {                                                                                                        \
            const int cc = ((c_) < A.nchunk) ? (c_) : A.nchunk - 1;                                              \
            _Pragma("unroll") for (int q = 0; q < NBQ; ++q) {                                                    \
                int item = (part_) * (NBQ * NLD6) + lt + NLD6 * q;                                               \
                if (item >= NBI) item = 0;                                                                       \
                const int h2 = item / NPX, px = item - h2 * NPX;                                                 \
                const int dw = px / IH, dh = px - dw * IH;                                                       \
                const float* p = isrc + (size_t)(cc * CK + h2 * 8) * A.in_plane + dw * A.in_hp + dh;             \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) rb_[q][j] = p[(size_t)j * A.in_plane];             \
            }                                                                                                    \
        }
#define STORE_B(c_, part_, rb_)                                                                                  \
        if ((c_) < A.nchunk) {                                                                                   \
            uint4* bd = Bbuf + ((c_) & 1) * (3 * 2 * NPX);                                                       \
            _Pragma("unroll") for (int q = 0; q < NBQ; ++q) {                                                    \
                const int item = (part_) * (NBQ * NLD6) + lt + NLD6 * q;                                         \
                if (item < NBI) {                                                                                \
                    uint4 s0, s1, s2;                                                                            \
                    split_pair(rb_[q][0], rb_[q][1], s0.x, s1.x, s2.x);                                          \
                    split_pair(rb_[q][2], rb_[q][3], s0.y, s1.y, s2.y);                                          \
                    split_pair(rb_[q][4], rb_[q][5], s0.z, s1.z, s2.z);                                          \
                    split_pair(rb_[q][6], rb_[q][7], s0.w, s1.w, s2.w);                                          \
                    bd[item] = s0;               /* item = h2*NPX + px ; split planes are 2*NPX apart */           \
                    bd[2 * NPX + item] = s1;                                                                     \
                    bd[4 * NPX + item] = s2;                                                                     \
                }                                                                                                \
            }                                                                                                    \
        }
        // prologue: B(chunk 0) in three parts, A(0); then the requests for iteration 0's stores
        for (int part = 0; part < 3; ++part) { LOAD_B(0, part, rb0) STORE_B(0, part, rb0) }
        LOAD_A(0, ra0) STORE_A(0, ra0)
        LOAD_A(1, ra1) LOAD_B(1, 0, rb1)
        lds_barrier6();                                             // barrier 0: step 0 may start
        for (int g = 0; g < nsteps; g += 2) {
            {   // iteration g: set 1 holds A(g+1), B(chunk(g)+1, part g%3); request iteration g+1's stores into set 0
                const int c = g / 3, part = g - 3 * c;
                const int c1 = (g + 1) / 3, part1 = (g + 1) - 3 * c1;
                LOAD_A(g + 2, ra0) LOAD_B(c1 + 1, part1, rb0)
                STORE_A(g + 1, ra1) STORE_B(c + 1, part, rb1)
                lds_barrier6();
            }
            if (g + 1 < nsteps) {
                const int c = (g + 1) / 3, part = (g + 1) - 3 * c;
                const int c1 = (g + 2) / 3, part1 = (g + 2) - 3 * c1;
                LOAD_A(g + 3, ra1) LOAD_B(c1 + 1, part1, rb1)
                STORE_A(g + 2, ra0) STORE_B(c + 1, part, rb0)
                lds_barrier6();
            }
        }
#undef LOAD_A
#undef STORE_A
#undef LOAD_B
#undef STORE_B
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h2 = lane >> 5;
    // pixel block (8h x 4w) and cout half of this wave
    const int pb = (CFG == 0) ? wave : (wave >> 1);
    const int pbh = (CFG == 0) ? (pb & 1) * 8 : 0, pbw = (CFG == 0) ? (pb >> 1) * 4 : pb * 4;
    const int m0 = (CFG == 0) ? 0 : (wave & 1);                     // first cout tile (of the workgroup's two) of this wave
    const int pxl = (pbw + (li >> 3)) * IH + pbh + (li & 7);        // halo-tile pixel of this lane at tap (0,0)
    f32x16 acc[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    lds_barrier6();                                                 // barrier 0
    for (int g = 0; g < nsteps; ++g) {
        const int c = g / 3, part = g - 3 * c;
        const uint4* ab = Abuf + (g & 1) * AST;
        const uint4* bb = Bbuf + (c & 1) * (3 * 2 * NPX) + h2 * NPX + pxl;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tap = part * 3 + t;                           // tap = kh*3 + kw ; part = kh
            const int toff = t * IH + part;                         // kw = t, kh = part
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bb[toff]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, bb[2 * NPX + toff]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, bb[4 * NPX + toff]);
            (void)tap;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const uint4* af = ab + ((t * 2 + (m0 + m)) * 3) * 64 + lane;
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, af[0]);
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[64]);
                const bf16x8 a2 = __builtin_bit_cast(bf16x8, af[128]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m], 0, 0, 0);
            }
        }
        lds_barrier6();                                             // barrier g+1
    }

    // ---- epilogue: C/D layout col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int oh = oh0 + pbh + (li & 7), ow = ow0 + pbw + (li >> 3);
    if (oh < A.H && ow < A.W) {
        const size_t po = (size_t)(ow + 1) * A.out_hp + (oh + 1);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 64 + (m0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                if (co < A.Cout) {
                    const size_t o = (size_t)co * A.out_plane + po;
                    float v = acc[m][r];
                    if (A.add1) v += A.add1[(size_t)b * A.add1_bs + o];
                    if (A.add2) v += A.add2[(size_t)b * A.add2_bs + o];
                    if (A.relu_out) v = fmaxf(v, 0.f);
                    A.out[(size_t)b * A.out_bs + o] = v;
                }
            }
        }
    }
}

template <int CFG> constexpr size_t conv6_lds() {
    return (size_t)(2 * AST + 2 * 3 * 2 * (Cfg6<CFG>::TH + 2) * (Cfg6<CFG>::TW + 2)) * 16;
}

template <int CFG>
int launch6(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
            const PTensor* add2, int relu_out) {
    typedef Cfg6<CFG> C;
    Conv6Args A;
    A.in = in.p; A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.p;
    A.add1 = add1 ? add1->p : nullptr; A.add2 = add2 ? add2->p : nullptr;
    A.Cout = L.Cout; A.W = in.W; A.H = in.H;
    A.in_hp = in.H + 2; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.H + 2; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.add1_bs = add1 ? (long)add1->Cal * add1->plane() : 0;
    A.add2_bs = add2 ? (long)add2->Cal * add2->plane() : 0;
    A.nchunk = L.nchunk6; A.n_ct = L.n_ct6;
    A.tiles_h = (in.H + C::TH - 1) / C::TH; A.tiles_w = (in.W + C::TW - 1) / C::TW;
    A.relu_out = relu_out;
    if (!ctx->conv6_attr[CFG]) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6_lds<CFG>()));
        ctx->conv6_attr[CFG] = true;
    }
    const int grid = A.n_ct * A.tiles_h * A.tiles_w * B;
    k_conv6<CFG><<<dim3(grid), dim3(NT6), conv6_lds<CFG>(), ctx->stream>>>(A);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

inline uint16_t host_bf16(float x) {                               // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u; std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float host_bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

}  // namespace

bool conv6_enabled() {
    static const bool on = !(getenv("QMRI_CONV_F32") && atoi(getenv("QMRI_CONV_F32")) > 0);
    return on;
}

// Weights (Conv2d OIHW) -> pre-split A fragments:
//   uint4 index = ((((ct64*nchunk + chunk)*9 + tap)*2 + m)*3 + split)*64 + lane ; the uint4 holds 8 bf16, element j:
//   row = ct64*64 + m*32 + (lane&31),  ci = chunk*16 + 8*(lane>>5) + j,  tap = kh*3 + kw
void conv6_plan_pack(ConvLayer& L, const float* w, std::vector<uint16_t>& packed) {
    L.nchunk6 = (L.Cin + CK - 1) / CK;
    L.n_ct6 = (L.Cout + 63) / 64;
    packed.assign((size_t)L.n_ct6 * L.nchunk6 * 9 * 2 * 3 * 64 * 8, 0);
    for (int ct = 0; ct < L.n_ct6; ++ct)
        for (int chunk = 0; chunk < L.nchunk6; ++chunk)
            for (int tap = 0; tap < 9; ++tap)
                for (int m = 0; m < 2; ++m)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int row = ct * 64 + m * 32 + (lane & 31);
                            const int ci = chunk * CK + 8 * (lane >> 5) + j;
                            if (row >= L.Cout || ci >= L.Cin) continue;
                            const float v = w[((size_t)row * L.Cin + ci) * 9 + tap];
                            const uint16_t h0 = host_bf16(v);
                            const float r1 = v - host_bf16_to_f(h0);
                            const uint16_t h1 = host_bf16(r1);
                            const float r2 = r1 - host_bf16_to_f(h1);
                            const uint16_t h2 = host_bf16(r2);
                            const size_t base = ((((size_t)ct * L.nchunk6 + chunk) * 9 + tap) * 2 + m) * 3;
                            packed[((base + 0) * 64 + lane) * 8 + j] = h0;
                            packed[((base + 1) * 64 + lane) * 8 + j] = h1;
                            packed[((base + 2) * 64 + lane) * 8 + j] = h2;
                        }
}

int conv6_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                 const PTensor* add2, int relu_out) {
    // wide pixel tiles while they still give every CU a workgroup; the small feature maps take the 64-pixel tile
    const long wide = (long)L.n_ct6 * ((in.H + 15) / 16) * ((in.W + 7) / 8) * B;
    if (wide >= 192 && in.H % 16 == 0) return launch6<0>(ctx, L, B, in, out, add1, add2, relu_out);
    return launch6<1>(ctx, L, B, in, out, add1, add2, relu_out);
}

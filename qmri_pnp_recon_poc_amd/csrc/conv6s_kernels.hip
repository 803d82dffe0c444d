// conv6s_kernels.hip -- k_conv6s: the 2x2 / stride-2 convolutions and transposed convolutions of UNetRes (network_unet.py:84-86, 100-102;
// basicblock.py:413-419, 437-443) on the operand-splitting schemes of conv6_kernels.hip (shared device code: conv6_device.h).
#include "conv6_device.h"

namespace {

// =====================================================================================================================
// k_conv6s : the 2x2 / stride-2 layers on the same operand-splitting schemes.
//   DOWN  Conv2d(k=2, s=2)           out[co][oh][ow]       = sum_ci,kh,kw w[co][ci][kh][kw] in[ci][2oh+kh][2ow+kw]
//   UP    ConvTranspose2d(k=2, s=2)  out[co][2ih+kh][2iw+kw] = sum_ci     w[ci][co][kh][kw] in[ci][ih][iw]
// Both are GEMMs over an 8h x 8w pixel tile (output pixels for DOWN, input pixels for UP) whose K steps hold two "planes":
//   DOWN  step g = (16-channel chunk c, kw): plane = kh (the tile's input pixels of row parity kh, column parity kw)
//   UP    step g = 32 channels: plane = 16-channel slice of the same pixels; the workgroup's 64 rows are kh = 0 / 1 x 32
//         output channels for one kw, so the LDS output tile interleaves the two kh rows and stores contiguous h.
// Waves 0-3: 2 row tiles x 2 pixel blocks (8h x 4w), 12 MFMAs per step; waves 4-7: loaders as in k_conv6 (asm requests two
// steps ahead, counted waits), each thread carries 2 channels x 4 consecutive h (one aligned float4 per channel).
// 32 KB (SP = 2) / 49 KB (SP = 3) of LDS: several workgroups share a CU and hide each other's barriers.
// =====================================================================================================================
constexpr int asts6(int SP) { return 2 * 2 * SP * 64; }   // uint4 per step of A: 2 planes x 2 row tiles x SP splits x 64 lanes
constexpr int STH = 8, STW = 8;           // pixel tile
constexpr int SNPX = STH * STW;           // LDS entries per (split, k-half, plane): pitch 8 = 8 mod 16, conflict-free

struct Conv6sArgs {
    const float* in; const uint4* wp; float* out;
    int Cout;                     // real output channels
    int GH, GW;                   // extent of the GEMM pixel grid (DOWN: output image, UP: input image)
    int in_hp, in_plane; long in_bs;
    int out_hp, out_plane; long out_bs;
    int nsteps, n_ct, tiles_h, tiles_w;   // nsteps is a multiple of 3 (the register rotation of the loaders); steps >= nsteps_real
    int nsteps_real;                      // carry zero weights and repeat the last step's activations
    unsigned* range_flag;                 // as in Conv6Args
    float descale_hi, descale_lo;
    int wt, xcd;                          // as in Conv6Args
    ActMax am;                            // as in Conv6Args
};

template <int N> __device__ __forceinline__ void gwait_s(u32x4 (&a)[3], f32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gwait_s(u32x4 (&a)[2], f32x4 (&b)[2]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}
__device__ __forceinline__ void gload4f(f32x4& dst, unsigned off, const void* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory"); }

template <int KIND, int SP, bool BLK>   // 0 = DOWN, 1 = UP; SP as in k_conv6; BLK: input and output are BLOCKED tensors (BRegs)
__global__ __launch_bounds__(NT6) void k_conv6s(const Conv6sArgs A) {
    constexpr int ASTS = asts6(SP);
    constexpr int NAS = ASTS / NLD6;                                // uint4 of A per loader thread and step
    static_assert(NAS * NLD6 == ASTS && NAS == SP, "loader split of A");
    constexpr int NLS = NAS + 2;                                    // vector-memory loads a loader thread issues per step
    constexpr int OPX = (KIND == 0) ? SNPX : 2 * SNPX;              // output pixels per row of the LDS output tile
    constexpr int OROWS = (KIND == 0) ? 64 : 32;                    // output channels of the workgroup
    constexpr int PPs = OPX + 4;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;                                     // [2][ASTS]
    unsigned* Bbuf = (unsigned*)(Abuf + 2 * ASTS);                  // [2][SP splits][2 k-halves][2 planes][SNPX] x 4 dwords
    constexpr int BSTEP = SP * 2 * 2 * SNPX * 4;                    // dwords of B per step; split planes are 2*2*SNPX*4 dwords apart
    float* ot = (SP == 3) ? (float*)Bbuf : (float*)smem;            // (SP == 2: aliases A too; the last stores into A precede the loop's last barrier)
    static_assert(SP == 3 ? (OROWS * PPs * 4 <= 2 * BSTEP * 4) : (OROWS * PPs * 4 <= 2 * ASTS * 16 + 2 * BSTEP * 4), "output tile must fit the operand buffers");
    const int tid = threadIdx.x;
    int bid = A.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;   // (the cout tiles of a pixel tile read the same activations: one L2)
    const int ct = bid % A.n_ct; bid /= A.n_ct;
    const int th = bid % A.tiles_h; bid /= A.tiles_h;
    const int tw = bid % A.tiles_w;
    const int b = bid / A.tiles_w;
    const int gh0 = th * STH, gw0 = tw * STW;                       // tile origin in the GEMM pixel grid
    const int nsteps = A.nsteps;

    if (tid >= NT6 - NLD6) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - (NT6 - NLD6);
        const uint4* wsrc = A.wp + (size_t)ct * nsteps * ASTS;
        unsigned aoff[NAS];
#pragma unroll
        for (int q = 0; q < NAS; ++q) aoff[q] = (unsigned)((lt + NLD6 * q) * 16);
        // PLANAR: this thread's activations are 2 channels (pair cp of an 8-channel half) x 4 consecutive h.
        // BLOCKED: one item = the 8 channels of (k-half h2, plane pl) at one pixel of the tile, 32 contiguous bytes; consecutive
        //          lanes take consecutive h of the input (DOWN: 16 = 8 output rows x kh; UP: 8), i.e. contiguous runs of 512 / 256 bytes
        const int cp = lt & 3, rest = lt >> 2;
        int h2, pl, hg, wq;                                         // k-half, plane (UP: channel slice), h group, column
        if (KIND == 0) { h2 = rest & 1; hg = (rest >> 1) & 3; wq = rest >> 3; pl = 0; }
        else { pl = rest & 1; h2 = (rest >> 1) & 1; hg = (rest >> 2) & 1; wq = rest >> 3; }
        // byte offsets of this thread's two requests relative to the step's base pointer.  PLANAR: channel 0 of the pair, first h; the
        // second channel = + plane.  BLOCKED: half (lt & 1) of items (lt >> 1) and (lt >> 1) + 128 (lane pairs = the halves of a pixel)
        unsigned boff, boff2;
        int bent[2] = {0, 0};                                       // BLOCKED: LDS entry (uint4 index inside one split plane of a step) of each item
        if constexpr (BLK) {
            unsigned bo[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int it = (lt >> 1) + (NLD6 / 2) * q;
                if (KIND == 0) {
                    const int bh = it & 15, ih2 = (it >> 4) & 1, iwq = it >> 5;      // input h inside the tile (= 2 * output row + kh), k-half, column
                    bo[q] = (unsigned)((((size_t)ih2) * A.in_plane + (size_t)(2 * iwq) * A.in_hp + bh) * 32);
                    bent[q] = (ih2 * 2 + (bh & 1)) * SNPX + iwq * STH + (bh >> 1);
                } else {
                    const int bh = it & 7, iwq = (it >> 3) & 7, ih2 = (it >> 6) & 1, ipl = it >> 7;
                    bo[q] = (unsigned)((((size_t)(ipl * 2 + ih2)) * A.in_plane + (size_t)iwq * A.in_hp + bh) * 32);
                    bent[q] = (ih2 * 2 + ipl) * SNPX + iwq * STH + bh;
                }
            }
            boff = bo[0] + 16u * (lt & 1); boff2 = bo[1] + 16u * (lt & 1);
        } else {
            if (KIND == 0) boff = (unsigned)((((size_t)(h2 * 8 + cp * 2)) * A.in_plane + (size_t)(2 * wq) * A.in_hp + 4 * hg) * 4);
            else boff = (unsigned)((((size_t)(pl * 16 + h2 * 8 + cp * 2)) * A.in_plane + (size_t)wq * A.in_hp + 4 * hg) * 4);
            boff2 = boff + (unsigned)A.in_plane * 4u;
        }
        // halo-free tile origin: padded coordinates = logical + 1  (BLOCKED: a pixel is 8 floats)
        constexpr int EPX = BLK ? 8 : 1;
        const float* isrc = A.in + (size_t)b * A.in_bs + ((KIND == 0) ? ((size_t)(2 * gw0 + 1) * A.in_hp + 2 * gh0 + 1)
                                                                      : ((size_t)(gw0 + 1) * A.in_hp + gh0 + 1)) * EPX;
        __builtin_amdgcn_s_setprio(2);
        u32x4 ra0[NAS], ra1[NAS], ra2[NAS];
        f32x4 rb0[2], rb1[2], rb2[2];
#define SLOAD(g_, ra_, rb_)                                                                                      \
        {                                                                                                        \
            const int ga = ((g_) < nsteps) ? (g_) : nsteps - 1, gg = (ga < A.nsteps_real) ? ga : A.nsteps_real - 1;   \
            const uint4* ws = uniform_ptr(wsrc + (size_t)ga * ASTS);                                             \
            _Pragma("unroll") for (int q = 0; q < NAS; ++q) gload4(ra_[q], aoff[q], ws);                         \
            const float* bs_ = (KIND == 0) ? uniform_ptr(isrc + (size_t)(gg >> 1) * CK * A.in_plane + (size_t)(gg & 1) * A.in_hp * EPX) \
                                           : uniform_ptr(isrc + (size_t)gg * 32 * A.in_plane);                   \
            gload4f(rb_[0], boff, bs_); gload4f(rb_[1], boff2, bs_);                                             \
        }
#define SSTORE(g_, ra_, rb_)                                                                                     \
        {                                                                                                        \
            uint4* ad = Abuf + ((g_) & 1) * ASTS;                                                                \
            _Pragma("unroll") for (int q = 0; q < NAS; ++q) ad[lt + NLD6 * q] = __builtin_bit_cast(uint4, ra_[q]); \
            unsigned* bd = Bbuf + ((g_) & 1) * BSTEP;                                                            \
            if constexpr (BLK) {                                                                                 \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                  \
                    uint2 s0, s1, s2;                                                                            \
                    if constexpr (SP == 3) { split_pair(rb_[q][0], rb_[q][1], s0.x, s1.x, s2.x); split_pair(rb_[q][2], rb_[q][3], s0.y, s1.y, s2.y); } \
                    else { split_pair_h(rb_[q][0], rb_[q][1], s0.x, s1.x); split_pair_h(rb_[q][2], rb_[q][3], s0.y, s1.y); } \
                    uint2* be = (uint2*)((uint4*)bd + bent[q]) + (lt & 1);                                       \
                    be[0] = s0; be[2 * (2 * 2 * SNPX)] = s1;                                                     \
                    if constexpr (SP == 3) be[2 * (2 * 2 * 2 * SNPX)] = s2;                                      \
                }                                                                                                \
            } else                                                                                               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                      \
                unsigned p0, p1, p2 = 0;                                                                         \
                if constexpr (SP == 3) split_pair(rb_[0][j], rb_[1][j], p0, p1, p2);                             \
                else split_pair_h(rb_[0][j], rb_[1][j], p0, p1);                                                 \
                const int hh = 4 * hg + j;                                                                       \
                const int plane_ = (KIND == 0) ? (hh & 1) : pl;                                                  \
                const int px = (KIND == 0) ? (wq * STH + (hh >> 1)) : (wq * STH + hh);                           \
                const int e = ((h2 * 2 + plane_) * SNPX + px) * 4 + cp;                                          \
                bd[e] = p0; bd[2 * 2 * SNPX * 4 + e] = p1;                                                       \
                if constexpr (SP == 3) bd[2 * 2 * 2 * SNPX * 4 + e] = p2;                                        \
            }                                                                                                    \
        }
        SLOAD(0, ra0, rb0) SLOAD(1, ra1, rb1) SLOAD(2, ra2, rb2)
        gwait_s<2 * NLS>(ra0, rb0);
        SSTORE(0, ra0, rb0)
        lds_barrier6();                                             // barrier 0
        // iteration g stores step g+1 (requested two iterations ago) and requests step g+3
#define SITER(k_, rs_a, rs_b, rq_a, rq_b)                                                                        \
        {                                                                                                        \
            __builtin_amdgcn_s_setprio(2);                                                                       \
            SLOAD(g + (k_) + 3, rq_a, rq_b)                                                                      \
            __builtin_amdgcn_s_setprio(0);                                                                       \
            gwait_s<2 * NLS>(rs_a, rs_b);                                                                        \
            SSTORE(g + (k_) + 1, rs_a, rs_b)                                                                     \
            lds_barrier6();                                                                                      \
        }
        for (int g = 0; g < nsteps; g += 3) {                      // (nsteps % 3 == 0: straight-line rotation, no copies of in-flight registers)
            SITER(0, ra1, rb1, ra0, rb0)
            SITER(1, ra2, rb2, ra1, rb1)
            SITER(2, ra0, rb0, ra2, rb2)
        }
        // drain; naming every register set here keeps the compiler from reusing the destinations of requests whose data is
        // never consumed (the clamped ones past the end) while they are still in flight
        gwait_s<0>(ra0, rb0); gwait_s<0>(ra1, rb1); gwait_s<0>(ra2, rb2);
#undef SITER
#undef SLOAD
#undef SSTORE
    } else {
        // ------------------------------------------------------------------ MFMA waves: row tile m0, pixel block (8h x 4w)
        const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h2 = lane >> 5;
        const int m0 = wave & 1, pbw = 4 * (wave >> 1);
        const int pxl = (pbw + (li >> 3)) * STH + (li & 7);
        f32x16 acc, accl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }
        lds_barrier6();                                             // barrier 0
        for (int g = 0; g < nsteps; ++g) {
            const uint4* ab = Abuf + (g & 1) * ASTS + lane;
            const uint4* bb = (const uint4*)(Bbuf + (g & 1) * BSTEP) + (h2 * 2) * SNPX + pxl;
            u32x4 bf[2][SP], af[2][SP];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) {
                    bf[t][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * 2 * SNPX + t * SNPX]);
                    af[t][sp] = __builtin_bit_cast(u32x4, ab[((t * 2 + m0) * SP + sp) * 64]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (SP == 3) {
                    acc = mfma_b(af[t][2], bf[t][0], acc);
                    acc = mfma_b(af[t][0], bf[t][2], acc);
                    acc = mfma_b(af[t][1], bf[t][1], acc);
                    acc = mfma_b(af[t][1], bf[t][0], acc);
                    acc = mfma_b(af[t][0], bf[t][1], acc);
                    acc = mfma_b(af[t][0], bf[t][0], acc);
                } else {
                    accl = mfma_h(af[t][1], bf[t][0], accl);
                    accl = mfma_h(af[t][0], bf[t][1], accl);
                    acc = mfma_h(af[t][0], bf[t][0], acc);
                }
            }
            lds_barrier6();                                         // barrier g+1
        }
        if constexpr (SP == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] * A.descale_hi + accl[r] * A.descale_lo;
        }
        // accumulators -> LDS output tile.  C/D layout: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h2;
            if (KIND == 0) ot[(m0 * 32 + row) * PPs + (pbw + (li >> 3)) * STH + (li & 7)] = acc[r];
            else ot[row * PPs + (pbw + (li >> 3)) * (2 * STH) + 2 * (li & 7) + m0] = acc[r];      // m0 = kh: rows interleave in h
        }
    }
    lds_barrier6();
    // ---- all eight waves.  BLOCKED: two half-items (4 channels of a block at one output pixel, 16 bytes) per thread; lane pairs take
    // the two halves of one pixel, so a wave stores contiguous runs (see k_conv6)
    if constexpr (BLK) {
        static_assert((OROWS / 8) * OPX == NT6 && PPs % 32 == 4, "epilogue");
        bool bad = false;
        float tmax = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e2 = k * NT6 + tid, half = e2 & 1, e = e2 >> 1;
            const int g = e / OPX, px = e - g * OPX;
            const float* op = ot + (g * 8 + 4 * half) * PPs + px;
            f32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = op[j * PPs];
            const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
            int cb, oh, ow;                                         // output channel block; output coordinates
            bool ok;
            if (KIND == 0) {
                const int w = px / STH, h = px - w * STH;
                cb = ct * 8 + g; oh = gh0 + h; ow = gw0 + w;
                ok = cb * 8 < A.Cout && oh < A.GH && ow < A.GW;
            } else {
                const int iw = px / (2 * STH), hh = px - iw * (2 * STH);   // hh = 2*ih + kh
                const int kw = ct & 1, ih = gh0 + (hh >> 1), iwg = gw0 + iw;
                cb = (ct >> 1) * 4 + g; oh = 2 * gh0 + hh; ow = 2 * iwg + kw;
                ok = cb * 8 < A.Cout && ih < A.GH && iwg < A.GW;
            }
            if (ok) {
                if constexpr (SP == 2) bad |= !(gm <= F16_RANGE);   // (stored values only)
                tmax = fmaxf(tmax, gm);
                store4(A.out + (size_t)b * A.out_bs + ((size_t)cb * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1)) * 8 + 4 * half, x, A.wt);
            }
        }
        if constexpr (SP == 2) {
            if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
            act_report(A.am, tmax, NT6 / 64);
        }
    } else {
        // PLANAR: aligned float4 rows of the output tile
        constexpr int NG = OROWS * OPX / 4, GQ = NG / NT6;
        static_assert(NG % NT6 == 0, "epilogue");
        bool bad = false;
        float tmax = 0.f;
#pragma unroll
        for (int k = 0; k < GQ; ++k) {
            const int e = k * NT6 + tid;
            const int co = e / (OPX / 4), rem = e - co * (OPX / 4);
            const f32x4 x = *(const f32x4*)(ot + co * PPs + 4 * rem);
            const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
            if (KIND == 0) {
                const int w = rem / (STH / 4), h = 4 * (rem - w * (STH / 4));
                const int cog = ct * 64 + co, oh = gh0 + h, ow = gw0 + w;
                if (cog < A.Cout && oh < A.GH && ow < A.GW) { tmax = fmaxf(tmax, gm); if constexpr (SP == 2) bad |= !(gm <= F16_RANGE); }
                if (cog < A.Cout && oh < A.GH && ow < A.GW)
                    store4(A.out + (size_t)b * A.out_bs + (size_t)cog * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1), x, A.wt);
            } else {
                const int iw = rem / (2 * STH / 4), hh = 4 * (rem - iw * (2 * STH / 4));   // hh = 2*ih + kh
                const int kw = ct & 1, cog = (ct >> 1) * 32 + co, ih = gh0 + (hh >> 1), iwg = gw0 + iw;
                if (cog < A.Cout && ih < A.GH && iwg < A.GW) { tmax = fmaxf(tmax, gm); if constexpr (SP == 2) bad |= !(gm <= F16_RANGE); }
                if (cog < A.Cout && ih < A.GH && iwg < A.GW)
                    store4(A.out + (size_t)b * A.out_bs + (size_t)cog * A.out_plane + (size_t)(2 * iwg + kw + 1) * A.out_hp + (2 * gh0 + hh + 1), x, A.wt);
            }
        }
        if constexpr (SP == 2) {
            if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
            act_report(A.am, tmax, NT6 / 64);
        }
    }
}

constexpr size_t conv6s_lds(int SP) { return (size_t)(2 * asts6(SP)) * 16 + (size_t)2 * SP * 2 * 2 * SNPX * 16; }

}  // namespace

// 2x2 / stride-2 layers: pre-split A fragments for k_conv6s
//   uint4 index = ((((ct*nsteps + g)*2 + plane)*2 + m)*SP + split)*64 + lane, element j, k = 8*(lane>>5) + j
//   DOWN (Conv2d OIHW):          row = ct*64 + m*32 + (lane&31) ; g = chunk*2 + kw ; plane = kh ; ci = chunk*16 + k
//   UP   (ConvTranspose2d IOHW): ct = cob*2 + kw ; m = kh ; co = cob*32 + (lane&31) ; plane = slice ; ci = g*32 + slice*16 + k
void conv6s_plan_pack(ConvLayer& L, const float* w, std::vector<uint16_t>& packed) {
    const bool up = (L.kind == CONV_UP);
    L.nsteps6s = up ? (L.Cin + 31) / 32 : 2 * ((L.Cin + CK - 1) / CK);        // real steps
    L.nchunk6 = ((L.nsteps6s + 2) / 3) * 3;                                   // padded with zero-weight steps to a multiple of 3
    L.n_ct6 = up ? 2 * ((L.Cout + 31) / 32) : (L.Cout + 63) / 64;
    const int SP = L.sp6;
    const float wscale = conv6_weight_scale(L, w, (size_t)L.Cout * L.Cin * 4);
    packed.assign((size_t)L.n_ct6 * L.nchunk6 * asts6(SP) * 8, 0);
    for (int ct = 0; ct < L.n_ct6; ++ct)
        for (int g = 0; g < L.nsteps6s; ++g)
            for (int plane = 0; plane < 2; ++plane)
                for (int m = 0; m < 2; ++m)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 8 * (lane >> 5) + j;
                            float v;
                            if (up) {
                                const int kw = ct & 1, kh = m, co = (ct >> 1) * 32 + (lane & 31), ci = g * 32 + plane * 16 + k;
                                if (co >= L.Cout || ci >= L.Cin) continue;
                                v = w[(((size_t)ci * L.Cout + co) * 2 + kh) * 2 + kw];
                            } else {
                                const int kw = g & 1, kh = plane, row = ct * 64 + m * 32 + (lane & 31), ci = (g >> 1) * CK + k;
                                if (row >= L.Cout || ci >= L.Cin) continue;
                                v = w[(((size_t)row * L.Cin + ci) * 2 + kh) * 2 + kw];
                            }
                            uint16_t h[3];
                            host_split(SP, v, h, wscale);
                            const size_t base = ((((size_t)ct * L.nchunk6 + g) * 2 + plane) * 2 + m) * SP;
                            for (int sp = 0; sp < SP; ++sp) packed[((base + sp) * 64 + lane) * 8 + j] = h[sp];
                        }
}

// the same packing on the device (see conv6_pack_dev): one thread per (ct, step, plane, m, lane) entry
namespace {
__global__ __launch_bounds__(256) void k_pack6s_w(const float* __restrict__ w, uint4* __restrict__ out, int up, int Cin, int Cout, int nsteps, int nsteps_real,
                                                   long nent, int SP, float scale) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nent) return;
    const int lane = (int)(e & 63), m = (int)((e >> 6) & 1), plane = (int)((e >> 7) & 1);
    long r = e >> 8;
    const int g = (int)(r % nsteps);
    const int ct = (int)(r / nsteps);
    unsigned short h[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 5) + j;
        float v = 0.f;
        if (g < nsteps_real) {
            if (up) {
                const int kw = ct & 1, kh = m, co = (ct >> 1) * 32 + (lane & 31), ci = g * 32 + plane * 16 + k;
                if (co < Cout && ci < Cin) v = w[(((size_t)ci * Cout + co) * 2 + kh) * 2 + kw];
            } else {
                const int kw = g & 1, kh = plane, row = ct * 64 + m * 32 + (lane & 31), ci = (g >> 1) * CK + k;
                if (row < Cout && ci < Cin) v = w[(((size_t)row * Cin + ci) * 2 + kh) * 2 + kw];
            }
        }
        dev_split(SP, v, h[j], scale);
    }
    const long base = ((((long)ct * nsteps + g) * 2 + plane) * 2 + m) * SP;
    for (int sp = 0; sp < SP; ++sp) {
        uint4 o;
        o.x = h[0][sp] | ((unsigned)h[1][sp] << 16); o.y = h[2][sp] | ((unsigned)h[3][sp] << 16);
        o.z = h[4][sp] | ((unsigned)h[5][sp] << 16); o.w = h[6][sp] | ((unsigned)h[7][sp] << 16);
        out[(base + sp) * 64 + lane] = o;
    }
}
}  // namespace

int conv6s_pack_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w, float wmax) {
    const bool up = (L.kind == CONV_UP);
    L.nsteps6s = up ? (L.Cin + 31) / 32 : 2 * ((L.Cin + CK - 1) / CK);
    L.nchunk6 = ((L.nsteps6s + 2) / 3) * 3;
    L.n_ct6 = up ? 2 * ((L.Cout + 31) / 32) : (L.Cout + 63) / 64;
    const float wscale = conv6_scale_from_max(L, wmax);
    const long nent = (long)L.n_ct6 * L.nchunk6 * 2 * 2 * 64;
    QMRI_HIP(ctx, hipMalloc(&L.wp6, (size_t)nent * L.sp6 * sizeof(uint4)));
    k_pack6s_w<<<dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_w, (uint4*)L.wp6, up ? 1 : 0, L.Cin, L.Cout, L.nchunk6, L.nsteps6s, nent, L.sp6, wscale);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// returns false if the layer/tensors do not meet the kernel's alignment assumptions (the f32 kernel then runs)
bool conv6s_usable(const ConvLayer& L, const PTensor& in, const PTensor& out) {
    if (!L.wp6 || (L.kind != CONV_DOWN && L.kind != CONV_UP)) return false;
    if (in.blk != out.blk) return false;
    if (in.blk) {                                                   // BLOCKED: 32-byte items, no alignment along h
        if (L.Cout % 8) return false;
        if (L.kind == CONV_DOWN) return in.H % 2 == 0 && in.W % 2 == 0 && in.Cal >= (L.nsteps6s / 2) * CK;
        return in.Cal >= L.nsteps6s * 32;
    }
    if (in.h0 % 4 || in.hp % 4 || out.h0 % 4 || out.hp % 4) return false;
    if (L.kind == CONV_DOWN) return in.H % 2 == 0 && in.W % 2 == 0 && (in.H / 2) % 4 == 0 && in.Cal >= (L.nsteps6s / 2) * CK;
    return in.H % 2 == 0 && in.Cal >= L.nsteps6s * 32;
}

int conv6s_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out) {
    const bool up = (L.kind == CONV_UP);
    Conv6sArgs A;
    if (in.blk != out.blk || (out.blk && L.Cout % 8 != 0)) {
        qmri_set_error(ctx, "conv layer %d: input and output of a 2x2 layer must share one tensor format", L.index);
        return QMRI_ERR_STATE;
    }
    A.in = in.fbase(); A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.fbase();
    A.Cout = L.Cout;
    A.GH = up ? in.H : in.H / 2; A.GW = up ? in.W : in.W / 2;
    A.in_hp = in.hp; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.hp; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.nsteps = L.nchunk6; A.nsteps_real = L.nsteps6s; A.n_ct = L.n_ct6;
    A.range_flag = ctx->net.d_range_flag;
    A.am = conv6_act_slot(ctx, L.sp6 == 2, L);
    A.wt = qmri_knob(K_CONV_WT);
    A.xcd = qmri_knob(K_CONV_XCD);
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    A.tiles_h = (A.GH + STH - 1) / STH; A.tiles_w = (A.GW + STW - 1) / STW;
    const int grid = A.n_ct * A.tiles_h * A.tiles_w * B;
    hipEvent_t e0 = nullptr, e1 = nullptr;                          // (profile level 2 only)
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_CONV2, conv_layer_flop(L, B, in.H, in.W)));
#define LAUNCH6S_(KIND_, SP_, BLK_)                                                                              \
    {                                                                                                            \
        if (e0) hipExtLaunchKernelGGL((k_conv6s<KIND_, SP_, BLK_>), dim3(grid), dim3(NT6), (std::uint32_t)conv6s_lds(SP_), ctx->stream, e0, e1, 0, A); \
        else k_conv6s<KIND_, SP_, BLK_><<<dim3(grid), dim3(NT6), conv6s_lds(SP_), ctx->stream>>>(A);             \
    }
#define LAUNCH6S(KIND_, SP_) { if (in.blk) LAUNCH6S_(KIND_, SP_, true) else LAUNCH6S_(KIND_, SP_, false) }
    if (L.sp6 == 2) { if (up) LAUNCH6S(1, 2) else LAUNCH6S(0, 2) }
    else { if (up) LAUNCH6S(1, 3) else LAUNCH6S(0, 3) }
#undef LAUNCH6S
#undef LAUNCH6S_
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

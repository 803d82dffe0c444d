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
// Data layout: activations live in HBM as [B][C][W+2][H+2] fp32 with a permanent ZERO HALO around every plane
// (h fastest: the memory order of a MATLAB H x W x C array, so kh pairs with h and kw with w).  Kernels write plane
// interiors only, so "zero padding 1" costs no bounds checks, no masks and no branches when an input tile is staged.
//
// Structure (each step below was asked for by a measurement; see DESIGN.md section 5 and profiles/):
//   * PERSISTENT workgroups (2 per CU) pull 32*MT (cout) x 32 (pixels: 4 along w x 8 along h) output tiles from a
//     device-side queue (one atomic per tile); tile coordinates come from a host-built table (no integer division).
//   * Waves 0-3 are MFMA waves: they split K (the input channels of every CC-channel chunk) four ways, read B
//     operands with ds_read_b32 from the staged input tile (lanes 0-31 channel 2p, lanes 32-63 channel 2p+1 of pair
//     p: the two k of a 32x32x2 step), stream A operands (weights) straight from L2 through a ring of R hand-issued
//     16-byte loads per lane with counted vmcnt (packed at load time so one load feeds four MFMA steps), and at a
//     tile's end only dump their accumulators to LDS and start the next tile.
//   * Waves 4-7 are loader waves (raised priority): they stage the input tile of the NEXT step global -> registers ->
//     LDS (double-buffered, one barrier per chunk) from offsets precomputed once per workgroup, fetch the next tile
//     id, and perform the PREVIOUS tile's epilogue (fixed-order sum of the four K-slices, + residual, + skip tensor,
//     ReLU, store) while the MFMA waves are already computing.
// Results are deterministic: every sum has a fixed order; the queue only decides which workgroup does a tile.
#include "qmri_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512;                // 4 MFMA waves + 4 loader waves
constexpr int NLD = 256;               // loader threads
constexpr int JW = 4, JH = 8;          // output pixel tile: 4 (w) x 8 (h) = the 32 columns of the MFMA tile

// geometry per layer kind; CC = input channels per chunk, R1/R2 = weight prefetch ring for MT = 1/2 (units of 4 MFMA
// steps: the lookahead LA = R-1 units is 4*MT MFMAs = 256*MT cycles each), VW = floats per staging load
template <int KIND> struct Geo;
template <> struct Geo<CONV_3X3>  { static constexpr int TH = 3, TW = 3, S = 1, PAD = 1, CC = 64, R1 = 6, R2 = 3, VW = 2; };
template <> struct Geo<CONV_3X3N> { static constexpr int TH = 3, TW = 3, S = 1, PAD = 1, CC = 32, R1 = 3, R2 = 3, VW = 2; };
template <> struct Geo<CONV_DOWN> { static constexpr int TH = 2, TW = 2, S = 2, PAD = 0, CC = 32, R1 = 4, R2 = 2, VW = 2; };
template <> struct Geo<CONV_UP>   { static constexpr int TH = 1, TW = 1, S = 1, PAD = 0, CC = 128, R1 = 4, R2 = 2, VW = 2; };

// The weight stream is issued by hand: hipcc (ROCm 7.2) sinks ordinary prefetch loads next to their first use and
// waits vmcnt(0) there, which exposes the L2 latency on every unit.  An asm load is invisible to the compiler's
// waitcnt bookkeeping (cdna_hip_programming.md section 5.7), so the ring is counted by hand: the MFMA waves issue
// no other vector-memory instruction, loads complete in issue order, and every wait names the registers it
// releases ("+v") so no consumer can be scheduled above it.
__device__ __forceinline__ void wload(f32x4& dst, unsigned byte_off, const float4* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void wwait(f32x4& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wwait(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

struct ConvArgs {
    const float* in; const float4* wp; float* out; const float* add1; const float* add2;
    const int4* tab;                  // per tile: {cout tile, ow0, oh0, batch index}
    int Cout;                         // logical output channels (UP: of the real output tensor)
    int W, H;                         // logical input extent; OW, OH logical output extent
    int OW, OH;
    int in_hp, in_plane; long in_bs;          // padded row pitch, plane size, batch stride (elements)
    int out_hp, out_plane; long out_bs, add1_bs, add2_bs;
    int nch, ntiles, relu_out;
    unsigned* counter;                // monotonic tile-queue counter shared by all launches of a context
    unsigned base;                    // its value when this launch starts (launches are stream ordered)
    unsigned long long* stamps;       // diagnostic build only (knob conv_stamps): per-workgroup timing stamps
};

template <int KIND, int MT>
__global__ __launch_bounds__(NT, 4) void k_conv(const ConvArgs A) {
    typedef Geo<KIND> G_;
    constexpr int TH = G_::TH, TW = G_::TW, S = G_::S, PAD = G_::PAD, CC = G_::CC, VW = G_::VW;
    constexpr int R = (MT == 2) ? G_::R2 : G_::R1, LA = R - 1;
    constexpr int IW = (JW - 1) * S + TW, IH = (JH - 1) * S + TH, PL = IW * IH, IHV = IH / VW;
    constexpr int NTAP = TH * TW, G = CC / 32, U = NTAP * G;
    constexpr int NITEM = CC * IW * IHV, NSTG = (NITEM + NLD - 1) / NLD, NEP = MT * 1024 / NLD;
    constexpr bool UPK = (KIND == CONV_UP);
    static_assert(IH % VW == 0, "staging vector width must divide the tile height");
    static_assert(U % R == 0, "prefetch ring must divide the units of a chunk");
    __shared__ __attribute__((aligned(16))) float inbuf[2 * CC * PL];
    __shared__ float red[4 * MT * 1024];
    __shared__ int4 s_tab[2];                          // table entry of this workgroup's next tile (by tile parity)
    __shared__ int s_id[2];                            // its id, -1 = queue exhausted

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nch = A.nch;
    int tile = blockIdx.x, tcount = 0;
    unsigned sp = 0;                                   // step parity: which half of inbuf the current step reads

    if (wave >= 4) {
        // =============================== loader waves ===============================
        __builtin_amdgcn_s_setprio(2);                 // their few VALU/VMEM instructions are the critical chain
        const int ltid = tid - 256;
        // ---- per-workgroup constants: relative offsets of this lane's staging items and output elements ----
        int rel[NSTG];
#pragma unroll
        for (int q = 0; q < NSTG; ++q) {
            int e = ltid + NLD * q;
            if (e >= NITEM) e = 0;                     // ragged tail: a harmless duplicate load, never written to LDS
            const int cl = e / (IW * IHV), rem = e - cl * (IW * IHV);
            const int ix = rem / IHV, iyv = rem - ix * IHV;
            rel[q] = cl * A.in_plane + ix * A.in_hp + iyv * VW;
        }
        int orel[NEP], opk[NEP];
#pragma unroll
        for (int q = 0; q < NEP; ++q) {
            const int f = ltid + NLD * q;
            const int mt = f >> 10, r = (f >> 6) & 15, l = f & 63;
            const int vrow = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            const int wx = (l & 31) >> 3, hy = l & 7;
            orel[q] = UPK ? (vrow * A.out_plane + 2 * wx * A.out_hp + 2 * hy) : (vrow * A.out_plane + wx * A.out_hp + hy);
            opk[q] = vrow | (wx << 8) | (hy << 16);
        }
        const bool up_fast = UPK && (A.Cout % (32 * MT) == 0);
        float v[NSTG * VW];
        float r1[NEP], r2[NEP];
        int4 cur = A.tab[tile];                        // {ct, ow0, oh0, b}
        int4 prv = cur;
        bool have_prev = false;

        auto issue_loads = [&](const int4& t, int c) {
            const float* src = A.in + (size_t)t.w * A.in_bs + (size_t)c * CC * A.in_plane +
                               (size_t)(t.y * S - PAD + 1) * A.in_hp + (t.z * S - PAD + 1);
#pragma unroll
            for (int q = 0; q < NSTG; ++q) {
                if (VW == 2) {
                    const float2 x = *reinterpret_cast<const float2*>(src + rel[q]);
                    v[2 * q] = x.x; v[2 * q + 1] = x.y;
                } else {
                    v[q] = src[rel[q]];
                }
            }
        };
        auto write_lds = [&](unsigned parity) {
            float* dst = inbuf + (parity & 1) * (CC * PL);
#pragma unroll
            for (int q = 0; q < NSTG; ++q) {
                const int e = ltid + NLD * q;
                if (e < NITEM) {
                    if (VW == 2) *reinterpret_cast<float2*>(dst + 2 * e) = make_float2(v[2 * q], v[2 * q + 1]);
                    else dst[e] = v[q];
                }
            }
        };
        // flat offset of this lane's output element q of tile t relative to a tensor with batch stride bs; <0: outside
        auto out_off = [&](const int4& t, int q, long bs, bool full) -> long {
            if (!UPK) {
                const long base = (long)t.w * bs + (long)(t.x * MT * 32) * A.out_plane + (long)(t.y + 1) * A.out_hp + t.z + 1;
                if (full) return base + orel[q];
                const int vrow = opk[q] & 255, wx = (opk[q] >> 8) & 255, hy = opk[q] >> 16;
                if (t.x * MT * 32 + vrow >= A.Cout || t.y + wx >= A.OW || t.z + hy >= A.OH) return -1;
                return base + orel[q];
            }
            const int vrow = opk[q] & 255, wx = (opk[q] >> 8) & 255, hy = opk[q] >> 16;
            if (up_fast) {
                const int vv0 = t.x * MT * 32, kk = vv0 / A.Cout, o0 = vv0 - kk * A.Cout;      // uniform per tile
                const long base = (long)t.w * bs + (long)o0 * A.out_plane + (long)(2 * t.y + (kk & 1) + 1) * A.out_hp + 2 * t.z + (kk >> 1) + 1;
                if (full) return base + orel[q];
                if (t.y + wx >= A.W || t.z + hy >= A.H) return -1;
                return base + orel[q];
            }
            const int vv = t.x * MT * 32 + vrow, kk = vv / A.Cout, o = vv - kk * A.Cout;
            if (kk >= 4 || t.y + wx >= A.W || t.z + hy >= A.H) return -1;
            return (long)t.w * bs + (long)o * A.out_plane + (long)(2 * (t.y + wx) + (kk & 1) + 1) * A.out_hp + 2 * (t.z + hy) + (kk >> 1) + 1;
        };
        auto tile_full = [&](const int4& t) -> bool {
            if (UPK) return up_fast && t.y + JW <= A.W && t.z + JH <= A.H;
            return (t.x + 1) * MT * 32 <= A.Cout && t.y + JW <= A.OW && t.z + JH <= A.OH;
        };
        auto prefetch_residual = [&](const int4& t) {
            if (UPK || (!A.add1 && !A.add2)) return;
            const bool full = tile_full(t);
#pragma unroll
            for (int q = 0; q < NEP; ++q) {
                if (A.add1) { const long o1 = out_off(t, q, A.add1_bs, full); r1[q] = (o1 >= 0) ? A.add1[o1] : 0.f; }
                if (A.add2) { const long o2 = out_off(t, q, A.add2_bs, full); r2[q] = (o2 >= 0) ? A.add2[o2] : 0.f; }
            }
        };
        auto epilogue = [&](const int4& t) {
            const bool full = tile_full(t);
#pragma unroll
            for (int q = 0; q < NEP; ++q) {
                const int f = ltid + NLD * q;
                const int mt = f >> 10, e = f & 1023;
                const float sum = ((red[(0 * MT + mt) * 1024 + e] + red[(1 * MT + mt) * 1024 + e]) +
                                   red[(2 * MT + mt) * 1024 + e]) + red[(3 * MT + mt) * 1024 + e];
                const long o = out_off(t, q, A.out_bs, full);
                if (o < 0) continue;
                float val = sum;
                if (!UPK) {
                    if (A.add1) val = r1[q] + val;
                    if (A.add2) val = val + r2[q];
                    if (A.relu_out) val = fmaxf(val, 0.f);
                }
                A.out[o] = val;
            }
        };

        // Tile-fetch pipeline (lane 0 of the first loader wave): the id and table entry of tile i+1 are obtained while
        // tile i-1 runs, so no atomic or table-load latency is ever exposed.  Every workgroup issues (tiles + 2) fetches.
        int f1 = -1;
        int4 t1 = cur;
        unsigned pend = 0;
        if (ltid == 0) {
            const unsigned r0 = gridDim.x + (atomicAdd(A.counter, 1u) - A.base);
            f1 = (r0 < (unsigned)A.ntiles) ? (int)r0 : -1;
            if (f1 >= 0) t1 = A.tab[f1];
            pend = atomicAdd(A.counter, 1u);
        }
        issue_loads(cur, 0);
        write_lds(0);
        for (;;) {
            int next_tile = -1;
            int4 nxt = cur;
            for (int c = 0; c < nch; ++c) {
                __syncthreads();                                         // step (tile, c) is published in inbuf[sp&1]
                if (c == 0 && ltid == 0) {
                    s_id[tcount & 1] = f1;                               // publish tile i+1 ...
                    s_tab[tcount & 1] = t1;
                    const unsigned r2 = gridDim.x + (pend - A.base);     // ... and move on to tile i+2
                    f1 = (r2 < (unsigned)A.ntiles) ? (int)r2 : -1;
                    if (f1 >= 0) t1 = A.tab[f1];
                    pend = atomicAdd(A.counter, 1u);
                }
                bool loaded = false;
                if (c + 1 < nch) { issue_loads(cur, c + 1); loaded = true; }
                if (c == 0 && have_prev) epilogue(prv);                  // previous tile: red -> out
                if (c == nch - 1) {
                    prefetch_residual(cur);                              // consumed by the epilogue one step later
                    next_tile = s_id[tcount & 1];                        // written at step 0, >= one barrier ago (nch >= 2)
                    if (next_tile >= 0) { nxt = s_tab[tcount & 1]; issue_loads(nxt, 0); loaded = true; }
                }
                if (loaded) write_lds(sp + 1);
                ++sp;
            }
            prv = cur; cur = nxt;
            have_prev = true;
            ++tcount;
            if (next_tile < 0) break;
        }
        __syncthreads();                                                 // last tile's accumulators are in red
        epilogue(prv);
    } else {
        // =============================== MFMA waves ===============================
        unsigned long long t_start = 0;
        if (A.stamps) t_start = __builtin_amdgcn_s_memrealtime();
        const int j = lane & 31, h = lane >> 5;
        const int wx = j >> 3, hy = j & 7;
        const int b_base = (wave * (CC / 4) + h) * PL + (wx * S) * IH + hy * S;
        constexpr unsigned CHSTRIDE = 4u * U * 64 * 16;                  // bytes between consecutive chunks of a cout tile
        auto tile_weights = [&](int ct, int mt) -> unsigned {            // byte offset of (cout tile, chunk 0, this wave, unit 0, lane)
            return (unsigned)(((((size_t)(ct * MT + mt) * nch) * 4 + wave) * (U * 64) + lane) * 16);
        };
        unsigned wcur[MT];                                               // weights of the current step (chunk of a tile)
        {
            const int ct0 = A.tab[tile].x;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) wcur[mt] = tile_weights(ct0, mt);
        }
        f32x4 a[R][MT];
#pragma unroll
        for (int u = 0; u < LA; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) wload(a[u][mt], wcur[mt] + u * 1024, A.wp);
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x16){0};
        for (;;) {
            int next_tile = -1;
            for (int c = 0; c < nch; ++c) {
                __syncthreads();
                const float* bt = inbuf + (sp & 1) * (CC * PL) + b_base;
                float bnxt[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bnxt[jj] = bt[2 * jj * PL];      // unit 0: tap (0,0), pairs 0..3
                unsigned wnxt[MT];                                       // weights of the step after this one
                if (c + 1 < nch) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) wnxt[mt] = wcur[mt] + CHSTRIDE;
                } else {
                    next_tile = s_id[tcount & 1];
                    const int ctn = s_tab[tcount & 1].x;                  // (a valid cout tile even when the queue is exhausted)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) wnxt[mt] = tile_weights(ctn, mt);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // Issue the weights of unit u+LA (first units of the next step when it crosses the chunk), then wait
                    // for unit u: exactly LA*MT younger loads stay in flight.  A load is issued on EVERY unit -- at the
                    // very end of the queue wnxt re-reads the current chunk -- so the count never changes and the ring
                    // has no branch the compiler could answer with register copies of in-flight destinations.
                    const int un = u + LA;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        wload(a[un % R][mt], (un < U) ? wcur[mt] + un * 1024 : wnxt[mt] + (un - U) * 1024, A.wp);
                    if (MT == 2) wwait<LA * MT>(a[u % R][0], a[u % R][MT - 1]); else wwait<LA * MT>(a[u % R][0]);
                    // B operands of unit u were read from LDS one unit ago (bcur); read unit u+1's now (bnxt) so the
                    // LDS latency hides behind this unit's MFMAs instead of sitting in front of each of them
                    float bcur[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bcur[jj] = bnxt[jj];
                    if (u + 1 < U) {
                        const int t1 = (u + 1) / G, g1 = (u + 1) - t1 * G;
                        const int kh1 = t1 / TW, kw1 = t1 - kh1 * TW;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bnxt[jj] = bt[2 * (4 * g1 + jj) * PL + kw1 * IH + kh1];
                    }
                    __builtin_amdgcn_sched_barrier(0);                   // keep those reads ABOVE this unit's MFMAs
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u % R][mt][jj], bcur[jj], acc[mt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) wcur[mt] = wnxt[mt];
                ++sp;
            }
            // tile finished: hand the four K-slices to the loader waves and go on with the next tile
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * MT + mt) * 1024 + r * 64 + lane] = acc[mt][r];
                acc[mt] = (f32x16){0};
            }
            ++tcount;
            if (next_tile < 0) break;
        }
        __syncthreads();
        if (A.stamps && tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* p = A.stamps + (size_t)blockIdx.x * 5;
            p[0] = t_start; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = hw; p[3] = xcc; p[4] = tcount;
        }
    }
}

int kind_cc(ConvKind k) {
    switch (k) {
        case CONV_3X3: return Geo<CONV_3X3>::CC;
        case CONV_3X3N: return Geo<CONV_3X3N>::CC;
        case CONV_DOWN: return Geo<CONV_DOWN>::CC;
        default: return Geo<CONV_UP>::CC;
    }
}

template <int KIND>
int launch_kind(qmri_ctx* ctx, ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                const PTensor* add2, int relu_out) {
    typedef Geo<KIND> G_;
    constexpr bool DOWNK = (KIND == CONV_DOWN);
    const int W = in.W, H = in.H;
    const int OW = DOWNK ? W / 2 : W, OH = DOWNK ? H / 2 : H;          // UP: tiles run over the INPUT pixels
    const int tiles_w = (OW + JW - 1) / JW, tiles_h = (OH + JH - 1) / JH, n_pt = tiles_w * tiles_h;
    const int n32 = L.n_ct;                                            // 32-row cout tiles in the packed weights
    // two MFMA row tiles per wave when the layer still has several tiles per workgroup slot that way
    const long t2 = (long)(n32 / 2) * n_pt * B;
    const bool is3 = (KIND == CONV_3X3 || KIND == CONV_3X3N);           // the six 2x2 layers always run one row tile per wave
    const long mt2_min = qmri_knob(K_CONV_MT2);
    const int MT = (is3 && n32 % 2 == 0 && t2 >= mt2_min) ? 2 : 1;
    const int n_ctiles = n32 / MT, ntiles = n_ctiles * n_pt * B;
    if (L.tab_B != B || L.tab_MT != MT || !L.d_tab) {                  // tile table: cout tile fastest (shared input tile -> L2 hits)
        std::vector<int4> tab((size_t)ntiles);
        size_t i = 0;
        for (int b = 0; b < B; ++b)
            for (int pt = 0; pt < n_pt; ++pt)
                for (int ct = 0; ct < n_ctiles; ++ct) tab[i++] = make_int4(ct, (pt / tiles_h) * JW, (pt % tiles_h) * JH, b);
        if (L.d_tab) QMRI_HIP(ctx, hipFree(L.d_tab));
        QMRI_HIP(ctx, hipMalloc((void**)&L.d_tab, tab.size() * sizeof(int4)));
        QMRI_HIP(ctx, hipMemcpyAsync(L.d_tab, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        L.tab_B = B; L.tab_MT = MT;
    }
    ConvArgs A;
    A.in = in.base1(); A.wp = reinterpret_cast<const float4*>(L.wp); A.out = out.base1();
    A.add1 = add1 ? add1->base1() : nullptr; A.add2 = add2 ? add2->base1() : nullptr;
    A.tab = reinterpret_cast<const int4*>(L.d_tab);
    A.Cout = L.Cout; A.W = W; A.H = H; A.OW = OW; A.OH = OH;
    A.in_hp = in.hp; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.hp; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.add1_bs = add1 ? (long)add1->Cal * add1->plane() : 0;
    A.add2_bs = add2 ? (long)add2->Cal * add2->plane() : 0;
    A.nch = L.cin_pad / G_::CC; A.ntiles = ntiles; A.relu_out = relu_out;
    A.counter = ctx->net.d_counter;
    const bool stamps6 = qmri_knob(K_CONV_STAMP_LAUNCH) >= 0;   // the buffer then belongs to k_conv6
    A.stamps = stamps6 ? nullptr : (unsigned long long*)ctx->net.d_stamps;
    // persistent grid: exactly as many workgroups as are resident at once (measured occupancy x CU count)
    // (per context, not function-local statics: qmri_recon_batch runs one host thread + context per device)
    int& ncu = ctx->conv_ncu;
    int* occ = ctx->conv_occ[(int)KIND];
    if (!ncu) {
        hipDeviceProp_t prop;
        QMRI_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ncu = prop.multiProcessorCount;
    }
    if (!occ[MT - 1]) {
        int nb = 0;
        if (MT == 2) QMRI_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_conv<(KIND == CONV_3X3N ? CONV_3X3N : CONV_3X3), 2>, NT, 0));
        else QMRI_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_conv<KIND, 1>, NT, 0));
        const int occ_cap = qmri_knob(K_CONV_OCC);
        occ[MT - 1] = std::max(1, std::min(nb, occ_cap));
    }
    const int grid = std::min(ntiles, ncu * occ[MT - 1]);
    // every workgroup issues (tiles it processes + 2) fetches, so a launch advances the counter by ntiles + 2*grid and the
    // host can mirror its value.  Inside a stream capture (a caller capturing its own stream) the arguments are frozen, so the queue is reset by
    // a memset node instead.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cap);
    if (cap == hipStreamCaptureStatusActive || ctx->net.counter_by_memset) {
        ctx->net.counter_by_memset = true;                 // once a graph exists the mirror is no longer valid
        QMRI_HIP(ctx, hipMemsetAsync(ctx->net.d_counter, 0, sizeof(unsigned), ctx->stream));
        A.base = 0;
    } else {
        A.base = ctx->net.counter_base;
        ctx->net.counter_base += (unsigned)(ntiles + 2 * grid);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;                  // profile level 2: the kernel's own dispatch timestamps
    {
        const bool k3 = (KIND == CONV_3X3 || KIND == CONV_3X3N);
        const double hw = k3 ? (double)in.H * in.W : (KIND == CONV_DOWN ? (double)(in.H / 2) * (in.W / 2) : (double)(2 * in.H) * (2 * in.W));
        QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, k3 ? PROF_CONV3 : PROF_CONV2, 2.0 * L.Cout * L.Cin * (k3 ? 9.0 : (KIND == CONV_DOWN ? 4.0 : 1.0)) * hw * B));
    }
    if (e0 && MT == 2) hipExtLaunchKernelGGL((k_conv<(KIND == CONV_3X3N ? CONV_3X3N : CONV_3X3), 2>), dim3(grid), dim3(NT), 0, ctx->stream, e0, e1, 0, A);
    else if (e0) hipExtLaunchKernelGGL((k_conv<KIND, 1>), dim3(grid), dim3(NT), 0, ctx->stream, e0, e1, 0, A);
    else if (MT == 2) k_conv<(KIND == CONV_3X3N ? CONV_3X3N : CONV_3X3), 2><<<dim3(grid), dim3(NT), 0, ctx->stream>>>(A);
    else k_conv<KIND, 1><<<dim3(grid), dim3(NT), 0, ctx->stream>>>(A);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

}  // namespace

int conv_cin_pad(ConvKind kind, int Cin) {
    if (kind == CONV_3X3 && Cin <= 64) kind = CONV_3X3N;
    const int CC = kind_cc(kind);
    return std::max(2 * CC, ((Cin + CC - 1) / CC) * CC);
}

void conv_plan_layer(ConvLayer& L, ConvKind kind, int Cin, int Cout) {
    if (kind == CONV_3X3 && Cin <= 64) kind = CONV_3X3N;       // narrow layers: 32-channel chunks
    L.kind = kind; L.Cin = Cin; L.Cout = Cout;
    const int CC = kind_cc(kind);
    L.cin_pad = std::max(2 * CC, ((Cin + CC - 1) / CC) * CC);   // >= 2 chunks: the tile pipeline needs a barrier between
    const int rows = (kind == CONV_UP) ? 4 * Cout : Cout;       // a tile's first step and its last
    L.n_ct = (rows + 31) / 32;
    L.MT = 1;
    L.wp = nullptr; L.wp_floats = 0;
    L.d_tab = nullptr; L.tab_B = 0; L.tab_MT = 0;
}

// Pack PyTorch-layout weights (Conv2d OIHW, ConvTranspose2d IOHW) into MFMA A-fragment order:
//   float4 index = (((ct32*nch + chunk)*4 + wave)*U + u)*64 + lane,  u = t*G + g ; component jj
//   row = ct32*32 + (lane&31),  ci = chunk*CC + wave*(CC/4) + 2*(4g+jj) + (lane>>5),  tap t = kh*TW + kw
size_t conv_pack_weights(const ConvLayer& L, const float* w, std::vector<float>& packed) {
    const bool is3 = (L.kind == CONV_3X3 || L.kind == CONV_3X3N);
    const int TH = is3 ? 3 : (L.kind == CONV_DOWN) ? 2 : 1, TW = TH;
    const int CC = kind_cc(L.kind);
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

// the same packing on the device (round 6, see conv6_pack_dev): one thread per float4 of the packed layout
namespace {
__global__ __launch_bounds__(256) void k_pack_w32(const float* __restrict__ w, float4* __restrict__ out, int kind, int Cin, int Cout, int CC, int NTAP, int nch, long nent) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nent) return;
    const int G = CC / 32;
    const int lane = (int)(e & 63);
    long r = e >> 6;
    const int g = (int)(r % G); r /= G;
    const int t = (int)(r % NTAP); r /= NTAP;
    const int wave = (int)(r & 3); r >>= 2;
    const int chunk = (int)(r % nch);
    const int ct = (int)(r / nch);
    const int row = ct * 32 + (lane & 31);
    float v[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int ci = chunk * CC + wave * (CC / 4) + 2 * (4 * g + jj) + (lane >> 5);
        v[jj] = 0.f;
        if (ci >= Cin) continue;
        if (kind == CONV_UP) {
            const int kk = row / Cout, o = row - kk * Cout;
            if (kk < 4) v[jj] = w[((size_t)ci * Cout + o) * 4 + kk];
        } else if (row < Cout) v[jj] = w[((size_t)row * Cin + ci) * NTAP + t];
    }
    out[e] = make_float4(v[0], v[1], v[2], v[3]);
}
}  // namespace

int conv_pack_weights_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w) {
    const bool is3 = (L.kind == CONV_3X3 || L.kind == CONV_3X3N);
    const int TH = is3 ? 3 : (L.kind == CONV_DOWN) ? 2 : 1;
    const int CC = kind_cc(L.kind), NTAP = TH * TH, G = CC / 32, nch = L.cin_pad / CC;
    const long nent = (long)L.n_ct * nch * 4 * NTAP * G * 64;
    L.wp_floats = (size_t)nent * 4;
    QMRI_HIP(ctx, hipMalloc((void**)&L.wp, L.wp_floats * sizeof(float)));
    k_pack_w32<<<dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_w, (float4*)L.wp, (int)L.kind, L.Cin, L.Cout, CC, NTAP, nch, nent);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int conv_launch(qmri_ctx* ctx, ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                const PTensor* add2, int relu_out) {
    if (L.wp6 && conv6_enabled() && !ctx->net.force_f32) {
        if (L.kind == CONV_3X3 || L.kind == CONV_3X3N) return conv6_launch(ctx, L, B, in, out, add1, add2, relu_out);
        if (!add1 && !add2 && !relu_out && conv6s_usable(L, in, out)) return conv6s_launch(ctx, L, B, in, out);
    }
    if (in.blk || out.blk || (add1 && add1->blk) || (add2 && add2->blk)) {      // (the f32-MFMA kernels below read planar tensors only)
        qmri_set_error(ctx, "conv layer %d: no matrix-core kernel for a blocked tensor", L.index);
        return QMRI_ERR_STATE;
    }
    switch (L.kind) {
        case CONV_3X3: return launch_kind<CONV_3X3>(ctx, L, B, in, out, add1, add2, relu_out);
        case CONV_3X3N: return launch_kind<CONV_3X3N>(ctx, L, B, in, out, add1, add2, relu_out);
        case CONV_DOWN: return launch_kind<CONV_DOWN>(ctx, L, B, in, out, add1, add2, relu_out);
        default: return launch_kind<CONV_UP>(ctx, L, B, in, out, add1, add2, relu_out);
    }
}

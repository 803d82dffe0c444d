// conv6p_kernels.hip -- k_conv6p, the persistent, software-pipelined form of the 3x3 convolution for slice batches (f16 x 3 scheme; the scheme
// and the one-launch-per-layer kernel it derives from: conv6_kernels.hip; shared device code: conv6_device.h).
// Reference semantics: Conv2d 3x3, stride 1, pad 1, no bias, optional ReLU / residual adds (basicblock.py:61-98, 211-223), single precision
// (denoiseImage_PnP_ADMM.m:72-77).
#include "conv6_device.h"

namespace {

// =====================================================================================================================
// k_conv6p : persistent, software-pipelined form of k_conv6 (f16 x 3 scheme) for launches with several tiles per CU -- slice
// batches (qmri_pnp_admm_dev with nslices > 1, qmri_recon_batch, bench.py --workload slices).
//
// In k_conv6 a workgroup is a serial prologue (first operands: 3.3 us) -> loop (11 us) -> epilogue (2.4-4.5 us), one workgroup per
// CU by LDS size, so with 11.5 tiles per CU (15 slices) the matrix cores idle for a third of the time.  Here one workgroup per CU
// walks tiles t = blockIdx.x, blockIdx.x + gridDim.x, ...:
//   * the loader waves treat the (tile, step) sequence as ONE stream: the requests that k_conv6 clamps "past the end" are the
//     next tile's first operands, so every tile after the first starts with its operands already in LDS;
//   * the MFMA waves, after a tile's last step, put the accumulators into an LDS tile `ot` of its own (158 KB of LDS in all) and
//     start the next tile at once;
//   * the loader waves run the finished tile's epilogue -- LDS tile + residual operands, ReLU, range guard, write-through stores --
//     in the issue gaps of the next tile's first 8 steps, 1/8 of the tile per step; the residual operands are requested two
//     steps ahead like every other operand (first two slices during the finished tile's own last two steps).
// Vector-memory operations of a wave complete in issue order and stores count like loads, so the loaders' one counted wait per
// step, vmcnt(2 * NLOAD), stays exactly as in k_conv6: at that point at least 2 * NLOAD younger operations have been issued
// (the operand requests of the two steps in between), and any epilogue load / store among them only makes the wait conservative.
// The last tile of a workgroup is finished by all eight waves as in k_conv6.
// Requirements (conv6_launch checks them, k_conv6 runs otherwise): f16 scheme, Cout % 64 == 0, nchunk even and >= 4, aligned
// tensors (vec4), no split-K.
// =====================================================================================================================
struct Tile6 { int ct, oh0, ow0, b; };

template <int CFG> __device__ __forceinline__ Tile6 tile6(const Conv6Args& A, int t) {
    Tile6 r;
    if (A.xcd) t = xcd_remap(t, A.ntiles);                          // (a workgroup's tiles t, t + gridDim.x, ... share t % 8: gridDim.x % 8 == 0 or gridDim.x == ntiles)
    r.ct = t % A.n_ct; t /= A.n_ct;
    const int th = t % A.tiles_h; t /= A.tiles_h;
    const int tw = t % A.tiles_w;
    r.b = t / A.tiles_w;
    r.oh0 = th * Cfg6<CFG>::TH; r.ow0 = tw * Cfg6<CFG>::TW;
    return r;
}

// STAMP: diagnostic build of the same kernel that records 100 MHz wall-clock stamps of four sampled workgroups (tools/conv6p_stamps.py)
#define P_STAMP(kind, idx)                                                                                       \
    do {                                                                                                         \
        if constexpr (STAMP) {                                                                                   \
            if (A.stamps && A.detail && (threadIdx.x & 255) == 0 && (idx) < 256) {                               \
                const int sw_ = (blockIdx.x == 0) ? 0 : (blockIdx.x == 37) ? 1 : (blockIdx.x == 101) ? 2 : (blockIdx.x == 200) ? 3 : -1; \
                if (sw_ >= 0) A.stamps[(sw_ * 10 + (kind)) * 256 + (idx)] = wall_clock64();                       \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)

template <int CFG, int NRES, bool STAMP>
__global__ __launch_bounds__(NT6) void k_conv6p(const Conv6Args A) {
    constexpr int SP = 2;
    constexpr int AST = ast6(SP);
    typedef Cfg6<CFG> C;
    constexpr int TH = C::TH, TW = C::TW, MW = C::MW, NCT = C::NCT;
    constexpr int IH = TH + 2, IW = TW + 2;
    constexpr int IHP = ((IH + 7) / 16) * 16 + 8;
    constexpr int NPX = IHP * (IW - 1) + IH;
    constexpr int NLP = IH * IW;
    constexpr int NBI = 2 * NLP;
    constexpr int NBQ = (NBI + 3 * NLD6 - 1) / (3 * NLD6);
    constexpr int NAQ = (AST + NLD6 - 1) / NLD6;
    static_assert(NAQ == 3 && NBQ == 1, "gwait() is written for 3 + 2 loads per step");
    constexpr int NLOAD = NAQ + 2;                                  // (BLOCKED tensors throughout: conv6_launch checks)
    constexpr int PXT = TH * TW;
    // epilogue: half-items (4 channels of a block at one pixel, 16 bytes; lane pairs = the two halves of a pixel, see k_conv6), 16 * PXT
    // per tile; a loader thread handles two per step: the same half at two pixels 128 apart (256-pixel tile) or in two blocks
    constexpr int NGS = NLD6 / PXT;                                 // channel blocks covered by the loader threads in one step
    constexpr int EPS = 8 / NGS;                                    // steps of the next tile that carry the epilogue = items per loader thread
    static_assert(NGS * PXT == NLD6 && EPS * NGS == 8 && EPS >= 4, "epilogue split");
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;                                     // [NABUF][AST]
    uint4* Bbuf = Abuf + NABUF * AST;                               // [2][SP][2 k-halves][NPX]
    float* ot = (float*)(Bbuf + 2 * SP * 2 * NPX);                  // [PXT][OTP] pixel-major output tile, NOT aliased: read while the next tile computes
    const int tid = threadIdx.x;
    const int nsteps = 3 * A.nchunk, ntiles = A.ntiles, tstride = gridDim.x;
    int tile = blockIdx.x;
    Tile6 last = tile6<CFG>(A, tile);                               // the tile whose output is in `ot` when the loop ends
    float tmaxp = 0.f;                                              // largest |output| this thread has stored (ACT_LOW)

    if (tid >= NT6 - NLD6) {
        // ------------------------------------------------------------------ loaders
        // A loader wave is INSTRUCTION-ISSUE bound (stamps: with 64-bit pointer arithmetic per request it needed 1.0-1.5 us per step
        // against 0.76 us of matrix work).  Every request is therefore a buffer instruction: one descriptor per tensor, the
        // per-lane part of the address in a loop-invariant VGPR, everything that moves (tile, chunk, step, epilogue slice) in the
        // 32-bit scalar offset; LDS addresses are loop-invariant VGPRs + immediates.
        const int lt = tid - (NT6 - NLD6);
        __builtin_amdgcn_s_setprio(2);
        const u32x4 srdW = make_srd(A.wp), srdI = make_srd(A.in), srdO = make_srd(A.out);
        const u32x4 srdR1 = make_srd(NRES > 0 ? (const void*)A.add1 : (const void*)A.out), srdR2 = make_srd(NRES > 1 ? (const void*)A.add2 : (const void*)A.out);
        constexpr unsigned ASTB = AST * 16;                         // bytes of A per step
        const unsigned plane4 = (unsigned)A.in_plane * 4u, oplane32 = (unsigned)A.out_plane * 32u;   // (bytes of a plane / of a block's plane)
        const unsigned chunkB = CK * plane4;                        // bytes between chunks of the input
        unsigned aoff[NAQ], boff[3][2];                             // per-lane byte offsets of this thread's requests
#pragma unroll
        for (int q = 0; q < NAQ; ++q) aoff[q] = (unsigned)((lt + NLD6 * q) * 16);      // (AST == NAQ * NLD6)
        static_assert(AST == NAQ * NLD6, "A requests");
        unsigned ldsB[3][2];                                        // LDS byte offset (inside one B buffer) of the half-items each part stores
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int q = 0; q < 2; ++q) {                           // half (lt & 1) of items part * 256 + (lt >> 1) and + 128, as in k_conv6
                int item = part * (NBQ * NLD6) + (lt >> 1) + (NLD6 / 2) * q;
                if (item >= NBI) item = 0;                          // (the last part is not full: surplus threads repeat item 0 -- same bytes, as in k_conv6)
                const int h2 = item / NLP, px = item - h2 * NLP;
                const int dw = px / IH, dh = px - dw * IH;
                boff[part][q] = (unsigned)(((size_t)h2 * A.in_plane + dw * A.in_hp + dh) * 32 + 16 * (lt & 1));
                ldsB[part][q] = (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + 8 * (lt & 1));
            }
        unsigned char* const ldsA = (unsigned char*)Abuf + lt * 16;                    // + buffer * ASTB + q * NLD6 * 16 (immediates)
        unsigned char* const ldsBb = (unsigned char*)Bbuf;
        // this thread's share of a tile's epilogue: half ehalf of pixels epx[q] in channel blocks egs[q] + NGS * j (j = step), q = 0, 1
        const int ehalf = lt & 1;
        int ew[2], eh[2];
        unsigned evoff[2];                                          // + scalar (tile, slice)
        const float* otp[2];                                        // + j * NGS * 8 (the next channel blocks of the same pixel)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = (lt >> 1) + (NLD6 / 2) * q, epx = idx % PXT, egs = idx / PXT;
            ew[q] = epx / TH; eh[q] = epx - ew[q] * TH;
            evoff[q] = (unsigned)(((size_t)egs * A.out_plane + (size_t)(ew[q] + 1) * A.out_hp + (eh[q] + 1)) * 32 + 16 * ehalf);
            otp[q] = ot + epx * OTP + egs * 8 + 4 * ehalf;
        }
        u32x4 ra0[NAQ], ra1[NAQ], ra2[NAQ];
        BRegs<true> rb0, rb1, rb2;
        f32x4 rr0[NRES > 0 ? NRES : 1][2], rr1[NRES > 0 ? NRES : 1][2], rr2[NRES > 0 ? NRES : 1][2];   // residual operands [operand][q], same rotation
        // scalar byte offsets of a tile inside the weights / the input / the output (and residual) tensors
        struct TOff { unsigned w, i, o, r1, r2; int oh0, ow0; };
        auto toff = [&](const Tile6& t) __attribute__((always_inline)) {
            TOff r;
            r.w = (unsigned)t.ct * (unsigned)A.nchunk_all * 3u * ASTB;
            r.i = (unsigned)((size_t)t.b * A.in_bs * 4 + ((size_t)t.ow0 * A.in_hp + t.oh0) * 32);
            const unsigned px = (unsigned)(((size_t)t.ct * 8 * A.out_plane + (size_t)t.ow0 * A.out_hp + t.oh0) * 32);
            r.o = (unsigned)((size_t)t.b * A.out_bs * 4) + px;
            r.r1 = (unsigned)((size_t)t.b * A.add1_bs * 4) + px;
            r.r2 = (unsigned)((size_t)t.b * A.add2_bs * 4) + px;
            r.oh0 = t.oh0; r.ow0 = t.ow0;
            return r;
        };
        TOff cur = toff(last), nx = cur, pv = cur;
        bool bad = false;
        // step / chunk indices are relative to the current tile; indices past its end address the next tile (or, after the last
        // tile, this one again: harmless re-reads into free buffers, as in k_conv6)
#define PLOAD_A(g_, ra_)                                                                                         \
        {                                                                                                        \
            const int gg_ = (g_);                                                                                \
            const unsigned so_ = (gg_ < nsteps) ? cur.w + (unsigned)gg_ * ASTB : nx.w + (unsigned)(gg_ - nsteps) * ASTB; \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) bload4(ra_[q], aoff[q], srdW, so_);                  \
        }
#define PSTORE_A(buf_, ra_)   /* buf_: compile-time A buffer */                                                  \
        {                                                                                                        \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) *(uint4*)(ldsA + (buf_) * ASTB + q * NLD6 * 16) = __builtin_bit_cast(uint4, ra_[q]); \
        }
#define PLOAD_B(c_, part_, rb_)                                                                                  \
        {                                                                                                        \
            const int cc_ = (c_);                                                                                \
            const unsigned so_ = (cc_ < A.nchunk) ? cur.i + (unsigned)cc_ * chunkB : nx.i + (unsigned)(cc_ - A.nchunk) * chunkB; \
            bload4f(rb_.q[0], boff[part_][0], srdI, so_); bload4f(rb_.q[1], boff[part_][1], srdI, so_);          \
        }
#define PSTORE_B(c_, part_, rb_)                                                                                 \
        {                                                                                                        \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
                unsigned char* bd = ldsBb + ((c_) & 1) * (SP * 2 * NPX * 16) + ldsB[part_][q];                   \
                uint2 s0, s1;                                                                                    \
                split_pair_h(rb_.q[q][0], rb_.q[q][1], s0.x, s1.x);                                              \
                split_pair_h(rb_.q[q][2], rb_.q[q][3], s0.y, s1.y);                                              \
                *(uint2*)bd = s0; *(uint2*)(bd + 2 * NPX * 16) = s1;                                             \
            }                                                                                                    \
        }
        // residual operands of epilogue slice j_ (channel blocks egs + NGS*j_) of tile t_: requested into set rr_.  Issued in EVERY step (a
        // step that has nothing to prefetch repeats slice 0 of the current tile): one unconditional instruction sequence, so the
        // destination registers of in-flight loads are never merged across branches (no copies of in-flight registers)
#define PREQ_RES(t_, j_, rr_)                                                                                    \
        if constexpr (NRES > 0) {                                                                                \
            const unsigned ko_ = (unsigned)((j_) * NGS) * oplane32;                                              \
            const unsigned so1_ = usgpr((t_).r1 + ko_), so2_ = (NRES > 1) ? usgpr((t_).r2 + ko_) : 0u;           \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
                const bool okhw_ = (t_).oh0 + eh[q] < A.H && (t_).ow0 + ew[q] < A.W;                             \
                const unsigned vo_ = okhw_ ? evoff[q] : 0u;                                                      \
                bload4f(rr_[0][q], vo_, srdR1, so1_);                                                            \
                if constexpr (NRES > 1) bload4f(rr_[1][q], vo_, srdR2, so2_);                                    \
            }                                                                                                    \
        }
        // epilogue slice j_ of tile t_ (its accumulators are in `ot`): LDS tile + residual operands (set rr_), ReLU, guard, store
#define PEPI(t_, j_, rr_)                                                                                        \
        {                                                                                                        \
            const unsigned so_ = usgpr((t_).o + (unsigned)((j_) * NGS) * oplane32);                              \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
                const bool okhw_ = (t_).oh0 + eh[q] < A.H && (t_).ow0 + ew[q] < A.W;                             \
                f32x4 x = *(const f32x4*)(otp[q] + (j_) * (NGS * 8));                                            \
                if constexpr (NRES > 0) x = x + rr_[0][q];                                                       \
                if constexpr (NRES > 1) x = x + rr_[1][q];                                                       \
                if (A.relu_out) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); } \
                if (okhw_) bstore4(x, evoff[q], srdO, so_);                                                      \
                {                                                                                                \
                    const float gm_ = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));   \
                    if (okhw_) { bad |= !(gm_ <= F16_RANGE); tmaxp = fmaxf(tmaxp, gm_); }      /* (stored values only, as in k_conv6) */ \
                }                                                                                                \
            }                                                                                                    \
        }
        // prologue of the first tile, as in k_conv6
        PLOAD_A(0, ra0) PLOAD_B(0, 0, rb0)
        PLOAD_A(1, ra1) PLOAD_B(0, 1, rb1)
        PLOAD_A(1, ra2) PLOAD_B(0, 2, rb2)
        gwait<2 * NLOAD>(ra0, rb0);
        PSTORE_A(0, ra0) PSTORE_B(0, 0, rb0)
        gwait<NLOAD>(ra1, rb1);
        PSTORE_A(1, ra1) PSTORE_B(0, 1, rb1)
        gwait<0>(ra2, rb2);
        PSTORE_B(0, 2, rb2)
        PLOAD_A(2, ra1) PLOAD_B(1, 0, rb1) PREQ_RES(cur, 0, rr1)    // (the residual sets in the steady-state order: operands, then residual)
        PLOAD_A(3, ra2) PLOAD_B(1, 1, rb2) PREQ_RES(cur, 0, rr2)
        lds_barrier6();                                             // barrier 0 of the first tile
        // Iteration g + k_ of the current tile (g = 3 * c0): requests A(g+k_+4), part (k_+2)%3 of B(c0 + (k_+2)/3 + 1) and the
        // residual operands of one epilogue slice into set rq; waits for set rs (requested two iterations ago); stores A(g+k_+2),
        // part k_ of B(c0+1); in steps 0..7 of every tile but the first runs epilogue slice g+k_ of the previous tile with the
        // residual operands of set rs.  Residual requests: steps 0..5 ask for slices 2..7 of the previous tile, the tile's last
        // two steps for slices 0 and 1 of the tile itself (consumed by steps 0 and 1 of the next tile).
#ifdef C6P_LOADER_IDLE  // (timing only: the matrix waves alone -- operands of the first steps stay in LDS, the loaders only keep the barriers)
#define PITER(k_, rs_a, rs_b, rs_r, rq_a, rq_b, rq_r) { lds_barrier6(); }
#else
#define PITER(k_, rs_a, rs_b, rs_r, rq_a, rq_b, rq_r)                                                            \
        {                                                                                                        \
            constexpr int part_ = (k_), part2_ = ((k_) + 2) % 3, dc2_ = ((k_) + 2) / 3;                         \
            const int gs_ = g + (k_);                                                                            \
            __builtin_amdgcn_s_setprio(2);                                                                       \
            PLOAD_A(gs_ + 4, rq_a)                                                                               \
            P_STAMP(7, sidx);                                                                                    \
            PLOAD_B(c0 + dc2_ + 1, part2_, rq_b)                                                                 \
            P_STAMP(8, sidx);                                                                                    \
            {                                                                                                    \
                const bool fromprev_ = have_prev && gs_ < EPS - 2;                                               \
                const TOff tq_ = fromprev_ ? pv : cur;                                                           \
                const int jq_ = fromprev_ ? gs_ + 2 : ((gs_ == nsteps - 1) ? 1 : 0);                             \
                PREQ_RES(tq_, jq_, rq_r)                                                                         \
            }                                                                                                    \
            __builtin_amdgcn_s_setprio(0);                                                                       \
            P_STAMP(2, sidx);                                                                                    \
            gwait<2 * (NLOAD + 2 * NRES)>(rs_a, rs_b);   /* exactly the requests issued since set rs: two iterations' operands and residuals */ \
            if constexpr (NRES > 0) { asm volatile("" : "+v"(rs_r[0][0]), "+v"(rs_r[0][1])); if constexpr (NRES > 1) asm volatile("" : "+v"(rs_r[1][0]), "+v"(rs_r[1][1])); } \
            P_STAMP(3, sidx);                                                                                    \
            PSTORE_A(((k_) + 2) % 3, rs_a) PSTORE_B(c0 + 1, part_, rs_b)   /* step g+k_+2 lives in A buffer (g+k_+2) % 3, g % 3 == 0 */ \
            P_STAMP(4, sidx);                                                                                    \
            if (have_prev && gs_ < EPS) PEPI(pv, gs_, rs_r)                                                      \
            P_STAMP(5, sidx);                                                                                    \
            lds_barrier6();                                                                                      \
            P_STAMP(6, sidx);                                                                                    \
            if constexpr (STAMP) ++sidx;                                                                         \
        }
#endif
        // ONE loop over the chunks of all tiles of this workgroup (no alternative code paths around in-flight registers)
        bool have_prev = false;
        int sidx = 0;                                               // (STAMP builds: running step number)
        bool has_next = tile + tstride < ntiles;
        if (has_next) nx = toff(tile6<CFG>(A, tile + tstride));
        for (int g = 0, c0 = 0;;) {
            PITER(0, ra1, rb1, rr1, ra0, rb0, rr0)
            PITER(1, ra2, rb2, rr2, ra1, rb1, rr1)
            PITER(2, ra0, rb0, rr0, ra2, rb2, rr2)
            g += 3; ++c0;
            if (c0 == A.nchunk) {                                   // tile boundary (scalar bookkeeping only)
                pv = cur;
                if (!has_next) break;
                tile += tstride;
                cur = nx;
                have_prev = true;
                g = 0; c0 = 0;
                has_next = tile + tstride < ntiles;
                if (has_next) nx = toff(tile6<CFG>(A, tile + tstride));
            }
        }
        last = tile6<CFG>(A, tile);
        gwait<0>(ra0, rb0); gwait<0>(ra1, rb1); gwait<0>(ra2, rb2);   // (requests past the end are still in flight)
        if constexpr (NRES > 0) {
#pragma unroll
            for (int q = 0; q < NRES; ++q) asm volatile("" : "+v"(rr0[q][0]), "+v"(rr0[q][1]), "+v"(rr1[q][0]), "+v"(rr1[q][1]), "+v"(rr2[q][0]), "+v"(rr2[q][1]));
        }
        if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
#undef PITER
#undef PLOAD_A
#undef PSTORE_A
#undef PLOAD_B
#undef PSTORE_B
#undef PREQ_RES
#undef PEPI
    } else {
        // ---------------------------------------------------------------------- MFMA waves
        const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h2 = lane >> 5;
        int pbh, pbw, m0;
        C::wave_map(wave, pbh, pbw, m0);
        const int pxl = (pbw + (li >> 3)) * IHP + pbh + (li & 7);
        f32x16 acc[MW][NCT], accl[MW][NCT];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
        lds_barrier6();                                             // barrier 0 of the first tile
        int sidx = 0;
        while (true) {
            for (int c = 0; c < A.nchunk; ++c) {
                const uint4* ab = Abuf + lane;
                const uint4* bb = Bbuf + (c & 1) * (SP * 2 * NPX) + h2 * NPX + pxl;
                u32x4 bf[2][NCT][SP], af[2][MW][SP];
                auto frag_a = [&](int T, int set, int m, int sp) __attribute__((always_inline)) {
                    const int kh = T / 3, kw = T - 3 * kh;
                    af[set][m][sp] = __builtin_bit_cast(u32x4, ab[kh * AST + ((kw * 2 + (m0 + m)) * SP + sp) * 64]);
                };
                auto frag_b = [&](int T, int set, int n, int sp) __attribute__((always_inline)) {
                    const int kh = T / 3, kw = T - 3 * kh;
                    bf[set][n][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * NPX + kw * IHP + kh + 8 * n]);
                };
                auto frags = [&](int T, int set) __attribute__((always_inline)) {
                    frag_a(T, set, 0, 0); frag_b(T, set, 0, 0); frag_a(T, set, 0, 1); frag_b(T, set, 0, 1);
#pragma unroll
                    for (int n = 1; n < NCT; ++n) { frag_b(T, set, n, 0); frag_b(T, set, n, 1); }
#pragma unroll
                    for (int m = 1; m < MW; ++m) { frag_a(T, set, m, 0); frag_a(T, set, m, 1); }
                };
                frags(0, 0);
                if (c == 0 && tile != (int)blockIdx.x) {            // a further tile: start from zero (the previous tile's sums are in `ot`)
#pragma unroll
                    for (int m = 0; m < MW; ++m)
#pragma unroll
                        for (int n = 0; n < NCT; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
                }
#pragma unroll
                for (int T = 0; T < 9; ++T) {
                    const int cu = T & 1;
                    if (T < 8) frags(T + 1, cu ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MW; ++m)
#pragma unroll
                        for (int n = 0; n < NCT; ++n) {
                            acc[m][n] = mfma_h(af[cu][m][0], bf[cu][n][0], acc[m][n]);
                            f32x16 l_ = accl[m][n];
                            l_ = mfma_h(af[cu][m][1], bf[cu][n][0], l_);
                            l_ = mfma_h(af[cu][m][0], bf[cu][n][1], l_);
                            accl[m][n] = l_;
                        }
                    if (T % 3 == 2) {
                        if (T == 8 && c == A.nchunk - 1) {
                            // the tile's last step: accumulators -> `ot` before the barrier that lets the loaders read it.  (The
                            // loaders finished reading the previous tile's `ot` in step 7 of this tile, several barriers ago.)
#pragma unroll
                            for (int n = 0; n < NCT; ++n)
#pragma unroll
                                for (int m = 0; m < MW; ++m)
#pragma unroll
                                    for (int rg = 0; rg < 4; ++rg) {
                                        f32x4 v;
#pragma unroll
                                        for (int j = 0; j < 4; ++j) v[j] = acc[m][n][4 * rg + j] * A.descale_hi + accl[m][n][4 * rg + j] * A.descale_lo;
                                        *(f32x4*)(ot + ((pbw + (li >> 3)) * TH + pbh + 8 * n + (li & 7)) * OTP + (m0 + m) * 32 + 8 * rg + 4 * h2) = v;
                                    }
                        }
                        P_STAMP(0, sidx);
                        lds_barrier6();
                        P_STAMP(1, sidx);
                        if constexpr (STAMP) ++sidx;
                    }
                }
            }
            tile += tstride;
            if (tile >= ntiles) break;
        }
        last = tile6<CFG>(A, tile - tstride);
    }

    // ---- the workgroup's last tile: all eight waves, as in k_conv6's BLOCKED epilogue (`ot` is complete: the loop's last barrier
    // follows its stores)
    {
        const int ct = last.ct, oh0 = last.oh0, ow0 = last.ow0, b = last.b;
        bool bad = false;
        constexpr int NHI = 16 * PXT, HQ = NHI / NT6;
        static_assert(NHI % NT6 == 0, "epilogue");
        unsigned off[HQ];
        f32x4 r1[HQ], r2[HQ];
#pragma unroll
        for (int k = 0; k < HQ; ++k) {
            const int e2 = k * NT6 + tid, half = e2 & 1, e = e2 >> 1;
            const int g = e / PXT, px = e - g * PXT, w = px / TH, h = px - w * TH;
            const int cb = ct * 8 + g, oh = oh0 + h, ow = ow0 + w;
            const bool ok = cb * 8 < A.Cout && oh < A.H && ow < A.W;
            off[k] = ok ? (unsigned)(((size_t)cb * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1)) * 8 + 4 * half) : ~0u;
            r1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; r2[k] = r1[k];
        }
        if constexpr (NRES > 0) {
#pragma unroll
            for (int k = 0; k < HQ; ++k) r1[k] = *(const f32x4*)(A.add1 + (size_t)b * A.add1_bs + ((off[k] != ~0u) ? off[k] : 8u));
        }
        if constexpr (NRES > 1) {
#pragma unroll
            for (int k = 0; k < HQ; ++k) r2[k] = *(const f32x4*)(A.add2 + (size_t)b * A.add2_bs + ((off[k] != ~0u) ? off[k] : 8u));
        }
#pragma unroll
        for (int k = 0; k < HQ; ++k) {
            const int e2 = k * NT6 + tid, half = e2 & 1, e = e2 >> 1;
            const int g = e / PXT, px = e - g * PXT;
            f32x4 x = *(const f32x4*)(ot + px * OTP + g * 8 + 4 * half);
            x = (x + r1[k]) + r2[k];
            if (A.relu_out) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
            if (off[k] != ~0u) store4(A.out + (size_t)b * A.out_bs + off[k], x, A.wt);
            const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
            if (off[k] != ~0u) { bad |= !(gm <= F16_RANGE); tmaxp = fmaxf(tmaxp, gm); }
        }
        if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
        act_report(A.am, tmaxp, NT6 / 64);
    }
}

// persistent form (k_conv6p): one workgroup per CU walks the launch's tiles; returns QMRI_OK and sets *done when it ran
template <int CFG, int NRES>
int launch6p_t(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1, const PTensor* add2,
               int relu_out) {
    typedef Cfg6<CFG> C;
    Conv6Args A{};
    A.in = in.fbase(); A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.fbase();   // (BLOCKED tensors: launch6p checks)
    A.add1 = add1 ? add1->fbase() : nullptr; A.add2 = add2 ? add2->fbase() : nullptr;
    A.in_blk = 1; A.out_blk = 1;
    A.Cout = L.Cout; A.W = in.W; A.H = in.H;
    A.in_hp = in.hp; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.hp; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.add1_bs = add1 ? (long)add1->Cal * add1->plane() : 0;
    A.add2_bs = add2 ? (long)add2->Cal * add2->plane() : 0;
    A.nchunk = L.nchunk6; A.nchunk_all = L.nchunk6; A.ksplit = 1; A.out_ks = 0; A.n_ct = L.n_ct6;
    A.tiles_h = (in.H + C::TH - 1) / C::TH; A.tiles_w = (in.W + C::TW - 1) / C::TW;
    A.ntiles = A.n_ct * A.tiles_h * A.tiles_w * B;
    A.relu_out = relu_out; A.vec4 = 1; A.wt = 1;
    A.xcd = qmri_knob(K_CONV_XCD);
    A.range_flag = ctx->net.d_range_flag;
    A.am = conv6_act_slot(ctx, true, L);
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    const int stamp_launch = qmri_knob(K_CONV_STAMP_LAUNCH);
    A.stamps = (unsigned long long*)ctx->net.d_stamps; A.launch_idx = g_conv6_launch_counter.fetch_add(1, std::memory_order_relaxed);
    A.detail = (A.stamps && A.launch_idx == stamp_launch) ? 1 : 0;
    if (!ctx->conv6p_attr[CFG][NRES]) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6p<CFG, NRES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6p_lds<CFG>()));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6p<CFG, NRES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6p_lds<CFG>()));
        ctx->conv6p_attr[CFG][NRES] = true;
    }
    const int grid = std::min(A.ntiles, ctx->conv_ncu);
    if (A.detail) {                                                 // diagnostic build of the same kernel (tools/conv6p_stamps.py)
        k_conv6p<CFG, NRES, true><<<dim3(grid), dim3(NT6), conv6p_lds<CFG>(), ctx->stream>>>(A);
        QMRI_HIP(ctx, hipGetLastError());
        return QMRI_OK;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_CONV3, conv_layer_flop(L, B, in.H, in.W)));
    if (e0) hipExtLaunchKernelGGL((k_conv6p<CFG, NRES, false>), dim3(grid), dim3(NT6), (std::uint32_t)conv6p_lds<CFG>(), ctx->stream, e0, e1, 0, A);
    else k_conv6p<CFG, NRES, false><<<dim3(grid), dim3(NT6), conv6p_lds<CFG>(), ctx->stream>>>(A);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

template <int CFG>
int launch6p(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1, const PTensor* add2,
             int relu_out, bool* done) {
    typedef Cfg6<CFG> C;
    *done = false;
    if (!qmri_knob(K_CONV_PERSIST) || L.sp6 != 2 || L.Cout % 64 != 0 || L.nchunk6 < 4 || L.nchunk6 % 2 != 0 || (add2 && !add1)) return QMRI_OK;
    if (!in.blk || !out.blk || (add1 && !add1->blk) || (add2 && !add2->blk)) return QMRI_OK;      // k_conv6p is written for BLOCKED tensors
    const bool same = (!add1 || (add1->h0 == out.h0 && add1->hp == out.hp && add1->plane() == out.plane())) &&
                      (!add2 || (add2->h0 == out.h0 && add2->hp == out.hp && add2->plane() == out.plane()));
    if (!same) return QMRI_OK;
    if (!ctx->conv_ncu) {
        hipDeviceProp_t prop;
        QMRI_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->conv_ncu = prop.multiProcessorCount;
    }
    const long ntiles = (long)L.n_ct6 * ((in.H + C::TH - 1) / C::TH) * ((in.W + C::TW - 1) / C::TW) * B;
    if (ntiles <= ctx->conv_ncu) return QMRI_OK;                   // at most one tile per CU: nothing to pipeline, k_conv6 is the same work
    *done = true;
    if (add2) return launch6p_t<CFG, 2>(ctx, L, B, in, out, add1, add2, relu_out);
    if (add1) return launch6p_t<CFG, 1>(ctx, L, B, in, out, add1, add2, relu_out);
    return launch6p_t<CFG, 0>(ctx, L, B, in, out, add1, add2, relu_out);
}

}  // namespace

int conv6p_try(qmri_ctx* ctx, int cfg, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1, const PTensor* add2,
               int relu_out, bool* done) {
    *done = false;
    if (cfg == 0) return launch6p<0>(ctx, L, B, in, out, add1, add2, relu_out, done);
    if (cfg == 1) return launch6p<1>(ctx, L, B, in, out, add1, add2, relu_out, done);
    return QMRI_OK;
}

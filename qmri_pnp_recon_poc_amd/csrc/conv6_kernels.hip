// conv6_kernels.hip -- the convolutions of the UNetRes / DRUNet denoiser on the 16-bit matrix cores with fp32-level accuracy.
//
// Reference semantics: denoiseImage_PnP_ADMM.m:1-117 runs the network in single precision; layers as in
// oracle/orc_net.c (Conv2d 3x3, stride 1, pad 1, no bias; optional ReLU; residual adds; 2x2 / stride-2 (transposed) convs).
//
// Every fp32 operand is split into SP 16-bit pieces and the piece products that matter are issued as 32x32x16 MFMAs with
// fp32 accumulation (32 cycles each, 16x the rate of v_mfma_f32_32x32x2_f32).  Two schemes, one kernel template:
//   SP = 2, "f16 x 3" (default): x = hi + lo'/2^11, hi = f16(x), lo' = f16((x - hi) * 2^11) (the scaling keeps lo' normal
//       whenever x is).  hi*hi goes into one accumulator, hi*lo' + lo'*hi into a second one that is folded in with 2^-11 in
//       the epilogue; lo*lo (2^-22) is dropped.  Per-operand error <= 2^-22 |x|: the result differs from an fp32 FMA chain
//       at the fp32 rounding level (tools/bf16x6_check.py: 2.2e-7 relative on K = 576, fp32 matmul 2.2e-7).  f16 carries
//       |x| <= 65504: weights are checked on the host, every epilogue raises Conv6Args::range_flag when an output is not
//       finite or above 6e4, and the callers then re-pack for SP = 3 and repeat (api_net.cpp: net_range_tripped).
//   SP = 3, "bf16 x 6" (knob conv_scheme = 3, and the fallback): x = x0 + x1 + x2 exactly (8 + 8 + 8 mantissa bits, no range
//       limit); of the nine piece products the six of order >= 2^-16,  w0 a0 + (w0 a1 + w1 a0) + (w1 a1 + w0 a2 + w2 a0),
//       are accumulated (8.6e-8 on the same tile).  Twice the matrix-core cycles of SP = 2.
//
// Implicit GEMM per workgroup: 64 output channels x (TH x TW) pixels, K = Cin*9 walked in chunks of 16 channels x 3 taps.
//   waves 0-3  MFMA: per tap SP A fragments per cout tile (weights, pre-split and pre-ordered on the host) and SP B
//              fragments per pixel block (activations) from LDS feed 3 (6) MFMAs per 32x32 tile
//   waves 4-7  loaders: weights global -> LDS (plain copy), activations global fp32 planes -> split -> LDS [pixel][8 ch],
//              requested two steps ahead and kept in registers for one
// Tensors stay fp32 padded planes in HBM (qmri_internal.h PTensor), so this kernel is interchangeable with k_conv.
#include "conv6_device.h"

std::atomic<int> g_conv6_launch_counter{0};

namespace {

// (Measured and removed: streaming the residual operand into an LDS tile during the last 8 steps of the loop, so that the epilogue
//  finds it on chip.  The loop is bound by the loader waves (tools/conv6p_stamps.py), so what the epilogue saved the loop lost:
//  634.9 vs 634.9 ADMM it/s, residual layers 21.3 us either way against 17.8 us for layers without a residual operand.)
template <int CFG, int SP, bool STAMP, bool INB>
__device__ __forceinline__ void conv6_body(const Conv6Args& A) {
    constexpr int AST = ast6(SP);
    typedef Cfg6<CFG> C;
    constexpr int TH = C::TH, TW = C::TW, MW = C::MW, NCT = C::NCT, MH = C::MH;
    constexpr int IH = TH + 2, IW = TW + 2;                         // input tile with halo
    constexpr int IHP = ((IH + 7) / 16) * 16 + 8;                   // its LDS row pitch, = 8 mod 16 entries: conflict-free ds_read_b128 of 8h x 4w blocks
    constexpr int NPX = IHP * (IW - 1) + IH;                        // LDS entries per (split, k-half) plane (the last row is not padded)
    constexpr int NLP = IH * IW;                                    // pixels actually loaded
    static_assert(IHP >= IH, "row pitch");
    constexpr int NBI = 2 * NLP;                                    // loader items of one chunk of B: (k-half, pixel)
    constexpr int NBQ = (NBI + 3 * NLD6 - 1) / (3 * NLD6);          // ... per loader thread and step (a chunk is spread over its 3 steps)
    constexpr int ASTH = AST / MH;                                  // uint4 of A this workgroup needs per step (its 32-row half, or all)
    constexpr int NAQ = (ASTH + NLD6 - 1) / NLD6;                   // ... per loader thread
    static_assert((NAQ == 5 || NAQ == 3 || NAQ == 2) && NBQ == 1, "gwait() is written for 5 / 3 / 2 loads of A and one item of B per step");
    // BLOCKED: half-items (16 bytes) per loader thread and step -- 2 * NBI of them per chunk over 3 steps x 256 threads: two on the
    // 256-pixel tile, one on the smaller ones (a request costs a loader wave ~0.05 us whether its lanes carry data or repeats)
    constexpr int NBH = (2 * NBI + 3 * NLD6 - 1) / (3 * NLD6);
    static_assert(NBH == 1 || NBH == 2, "half-items per step");
    constexpr int NBL = INB ? NBH : 8;                              // requests per step of B
    constexpr int NLOAD = NAQ + NBL;                                // vector-memory loads a loader thread issues per step
    typedef typename std::conditional<INB && NBH == 1, BRegs1, BRegs<INB>>::type BR;
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;                                     // [NABUF][AST]
    constexpr int PXT = TH * TW, PP = PXT + 4;                      // output tile in LDS: [64 cout][PP], aliases the B buffers
    uint4* Bbuf = Abuf + NABUF * AST;                                   // [2][SP splits][2 k-halves][NPX]  (8 channels = 16 B per entry)
    // The output tile aliases the operand buffers (SP == 3: the B buffers; SP == 2: from the start -- the loaders' last
    // stores into A precede the loop's last barrier, the tile is written after it).
    float* ot = (SP == 3) ? (float*)Bbuf : (float*)smem;
    static_assert(SP == 3 ? (64 * PP * 4 <= 2 * 3 * 2 * NPX * 16) : (64 * PP * 4 <= (NABUF * AST + 2 * SP * 2 * NPX) * 16), "output tile must fit the operand buffers");
    static_assert(SP == 3 ? (PXT * OTP * 4 <= 2 * 3 * 2 * NPX * 16) : (PXT * OTP * 4 <= (NABUF * AST + 2 * SP * 2 * NPX) * 16), "pixel-major output tile must fit the operand buffers");
    const int tid = threadIdx.x;
    int bid = A.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    // (Measured and removed, round 6: at the 28 x 28 level, where a workgroup reads more weight than activation bytes, the PIXEL tiles of one
    //  weight tile consecutive instead -- an XCD's L2 then shares weights, 1.2 MB read eight times, instead of activations: 871 / 875 against
    //  867 / 875 ADMM it/s, profiles/r06_c_ab_deep_level_weight_major_tile_order_not_kept.txt: the loaders do not wait for the Infinity Cache.)
    const int mh = (MH > 1) ? bid % MH : 0;                         // which 32-row half of the 64-row tile (MH = 2)
    if (MH > 1) bid /= MH;
    const int ct = bid % A.n_ct; bid /= A.n_ct;
    const int th = bid % A.tiles_h; bid /= A.tiles_h;
    const int tw = bid % A.tiles_w; bid /= A.tiles_w;
    const int ks = bid % A.ksplit;                                  // K slice of this workgroup (split-K layers)
    const int b = bid / A.ksplit;
    const int oh0 = th * TH, ow0 = tw * TW;
    const int nsteps = 3 * A.nchunk;

    if (tid >= NT6 - NLD6) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - (NT6 - NLD6);
        __builtin_amdgcn_s_setprio(2);                             // requests first: the MFMA waves have work queued anyway
        // A loader wave is instruction-issue bound (tools/conv6p_stamps.py), so every request is a buffer instruction: one
        // descriptor per tensor, the per-lane part of the address in a loop-invariant VGPR, whatever moves in the 32-bit scalar offset
        const u32x4 srdW = make_srd(A.wp), srdI = make_srd(A.in);
        constexpr unsigned ASTB = AST * 16;                         // bytes of A per step
        const unsigned plane4 = (unsigned)A.in_plane * 4u, chunkB = CK * plane4;
        const unsigned wbase = (unsigned)(((size_t)ct * A.nchunk_all + (size_t)ks * A.nchunk) * 3 * ASTB);   // steps of a cout tile are contiguous
        // halo origin = padded (oh0, ow0); a chunk (16 channels = two blocks of 8) is CK planes further in either format
        const unsigned ibase = INB ? (unsigned)(((size_t)b * A.in_bs + (size_t)ks * A.nchunk * CK * A.in_plane) * 4 + ((size_t)ow0 * A.in_hp + oh0) * 32)
                                   : (unsigned)(((size_t)b * A.in_bs + (size_t)ks * A.nchunk * CK * A.in_plane + (size_t)ow0 * A.in_hp + oh0) * 4);
        unsigned aoff[NAQ], boff[3][INB ? 2 : 8], ldsB[3][INB ? 2 : 1];   // loop-invariant byte offsets of this thread's requests / LDS stores
#pragma unroll
        for (int q = 0; q < NAQ; ++q) {                             // entry index in the step's A layout ((kw 2 + m) SP + sp) 64 + lane
            int j = lt + NLD6 * q;
            if (j >= ASTH) j = 0;
            if (MH > 1) { const int kw = j / (SP * 64); j += (kw + mh) * (SP * 64); }     // the entries with m == mh
            aoff[q] = (unsigned)(j * 16);
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            if constexpr (INB) {
                // half-items part * 256 NBH + 256 q + lt: half (lt & 1) of item (part NBH + q) * 128 + (lt >> 1)
#pragma unroll
                for (int q = 0; q < NBH; ++q) {
                    int item = (part * NBH + q) * (NLD6 / 2) + (lt >> 1);
                    if (item >= NBI) item = 0;
                    const int h2 = item / NLP, px = item - h2 * NLP;
                    const int dw = px / IH, dh = px - dw * IH;
                    boff[part][q] = (unsigned)((((size_t)h2) * A.in_plane + dw * A.in_hp + dh) * 32 + 16 * (lt & 1));
                    ldsB[part][q] = (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + 8 * (lt & 1));
                }
            } else {
                int item = part * (NBQ * NLD6) + lt;
                if (item >= NBI) item = 0;
                const int h2 = item / NLP, px = item - h2 * NLP;
                const int dw = px / IH, dh = px - dw * IH;
                const unsigned b0 = (unsigned)((((size_t)(h2 * 8)) * A.in_plane + dw * A.in_hp + dh) * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) boff[part][j] = b0 + (unsigned)j * plane4;   // the 8 channels differ by a plane
                ldsB[part][0] = (unsigned)((h2 * NPX + dw * IHP + dh) * 16);
            }
        }
        u32x4 ra0[NAQ], ra1[NAQ], ra2[NAQ];
        BR rb0, rb1, rb2;
        // Schedule.  Barrier g precedes compute step g.  Abuf[(g+1)&1] is free once step g-1 is over, i.e. after barrier g:
        // iteration g (between barriers g and g+1) stores A(g+1).  Bbuf[(c+1)&1] is free once chunk c-1 is over, i.e. after
        // barrier 3c: iterations 3c, 3c+1, 3c+2 store the three parts of B(c+1).  What an iteration stores was requested
        // two iterations earlier (three register sets in rotation), so a request has two whole steps to arrive.
        // (Requests past the end are clamped, not skipped: branch-free code lets the compiler count vmcnt exactly.)
#define LOAD_A(g_, ra_)                                                                                          \
        {                                                                                                        \
            const unsigned so_ = wbase + (unsigned)(((g_) < nsteps) ? (g_) : nsteps - 1) * ASTB;                 \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) bload4(ra_[q], aoff[q], srdW, so_);                  \
        }
#define STORE_A(buf_, ra_)     /* buf_ = step % NABUF, a compile-time constant; surplus threads repeat entry 0 (clamped request) */ \
        {                                                                                                        \
            uint4* ad = Abuf + (buf_) * AST;                                                                     \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) *(uint4*)((unsigned char*)ad + aoff[q]) = __builtin_bit_cast(uint4, ra_[q]); \
        }
#define LOAD_B(c_, part_, rb_)                                                                                   \
        {                                                                                                        \
            const unsigned so_ = ibase + (unsigned)(((c_) < A.nchunk) ? (c_) : A.nchunk - 1) * chunkB;           \
            if constexpr (INB) { _Pragma("unroll") for (int q = 0; q < NBH; ++q) bload4f(rb_.q[q], boff[part_][q], srdI, so_); } \
            else { _Pragma("unroll") for (int j = 0; j < 8; ++j) bload1(rb_.v[j], boff[part_][j], srdI, so_); }  \
        }
#define STORE_B(c_, part_, rb_)                                                                                  \
        if constexpr (INB) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q < NBH; ++q) {                                                    \
                unsigned char* bd = (unsigned char*)(Bbuf + ((c_) & 1) * (SP * 2 * NPX)) + ldsB[part_][q];       \
                uint2 s0, s1, s2;                /* 4 channels = 8 bytes of a 16-byte entry */                    \
                if constexpr (SP == 3) {                                                                         \
                    split_pair(rb_.q[q][0], rb_.q[q][1], s0.x, s1.x, s2.x);                                      \
                    split_pair(rb_.q[q][2], rb_.q[q][3], s0.y, s1.y, s2.y);                                      \
                } else {                                                                                         \
                    split_pair_h(rb_.q[q][0], rb_.q[q][1], s0.x, s1.x);                                          \
                    split_pair_h(rb_.q[q][2], rb_.q[q][3], s0.y, s1.y);                                          \
                }                                                                                                \
                *(uint2*)bd = s0;                /* split planes are 2*NPX entries apart */                       \
                *(uint2*)(bd + 2 * NPX * 16) = s1;                                                               \
                if constexpr (SP == 3) *(uint2*)(bd + 4 * NPX * 16) = s2;                                        \
            }                                                                                                    \
        } else {                                                                                                 \
            unsigned char* bd = (unsigned char*)(Bbuf + ((c_) & 1) * (SP * 2 * NPX)) + ldsB[part_][0];           \
            uint4 s0, s1, s2;                                                                                    \
            if constexpr (SP == 3) {                                                                             \
                split_pair(rb_.get(0), rb_.get(1), s0.x, s1.x, s2.x);                                            \
                split_pair(rb_.get(2), rb_.get(3), s0.y, s1.y, s2.y);                                            \
                split_pair(rb_.get(4), rb_.get(5), s0.z, s1.z, s2.z);                                            \
                split_pair(rb_.get(6), rb_.get(7), s0.w, s1.w, s2.w);                                            \
            } else {                                                                                             \
                split_pair_h(rb_.get(0), rb_.get(1), s0.x, s1.x);                                                \
                split_pair_h(rb_.get(2), rb_.get(3), s0.y, s1.y);                                                \
                split_pair_h(rb_.get(4), rb_.get(5), s0.z, s1.z);                                                \
                split_pair_h(rb_.get(6), rb_.get(7), s0.w, s1.w);                                                \
            }                                                                                                    \
            *(uint4*)bd = s0;                    /* split planes are 2*NPX entries apart */                       \
            *(uint4*)(bd + 2 * NPX * 16) = s1;                                                                   \
            if constexpr (SP == 3) *(uint4*)(bd + 4 * NPX * 16) = s2;                                            \
        }
        // prologue: all of B(chunk 0), A(0) and A(1) -- and, behind them in the same burst, what iterations 0 and 1 store (sets 1 and 2
        // of the rotation): ONE memory latency before the loop instead of two (the first two steps used to wait for requests issued
        // only after the first batch had arrived: 1.36 us for step 0 against 0.8 in steady state).  The first chunk's second and third
        // part travel in prologue-only registers.
        C6_STAMP(2, 0);
        {
            u32x4 pa1[NAQ], pa2[NAQ];
            BR pb1, pb2;
            LOAD_A(0, ra0) LOAD_B(0, 0, rb0)
            LOAD_A(1, pa1) LOAD_B(0, 1, pb1)
            LOAD_A(1, pa2) LOAD_B(0, 2, pb2)                        // (A again: keeps the request count per batch uniform)
            LOAD_A(2, ra1) LOAD_B(1, 0, rb1)                        // stored by iteration 0
            LOAD_A(3, ra2) LOAD_B(1, 1, rb2)                        // stored by iteration 1
            gwait<4 * NLOAD>(ra0, rb0);
            C6_STAMP(3, 0);
            STORE_A(0, ra0) STORE_B(0, 0, rb0)
            gwait<3 * NLOAD>(pa1, pb1);
            STORE_A(1, pa1) STORE_B(0, 1, pb1)
            gwait<2 * NLOAD>(pa2, pb2);
            STORE_B(0, 2, pb2)
        }
        C6_STAMP(1, 0);
        lds_barrier6();                                             // barrier 0: step 0 may start
        // iteration g stores A(g+2) and part g%3 of B(g/3+1) from set (g+1)%3 and requests what iteration g+2 stores,
        // A(g+4) and part (g+2)%3 of B((g+2)/3+1), into set g%3 (whose content iteration g-1 stored).  At the wait the
        // requests of this and of the previous iteration may stay in flight: vmcnt(2*NLOAD).
#ifdef C6_LOADER_IDLE   // (timing only: k_conv6's matrix waves alone after the prologue)
#define ITER(k_, rs_a, rs_b, rq_a, rq_b) { lds_barrier6(); }
#else
#define ITER(k_, rs_a, rs_b, rq_a, rq_b)   /* iteration g + k_, g = 3*c0 */                                       \
        {                                                                                                        \
            constexpr int part_ = (k_), part2_ = ((k_) + 2) % 3, dc2_ = ((k_) + 2) / 3;                         \
            __builtin_amdgcn_s_setprio(2);         /* requests first ... */                                       \
            LOAD_A(g + (k_) + 4, rq_a) LOAD_B(c0 + dc2_ + 1, part2_, rq_b)                                       \
            __builtin_amdgcn_s_setprio(0);         /* ... the split arithmetic only in the MFMA waves' issue gaps */ \
            C6_STAMP(2, g + (k_) + 1);                                                                           \
            gwait<2 * NLOAD>(rs_a, rs_b);                                                                        \
            C6_STAMP(3, g + (k_) + 1);                                                                           \
            STORE_A(((k_) + 2) % NABUF, rs_a) STORE_B(c0 + 1, part_, rs_b)   /* step g+k_+2, g % 3 == 0 */        \
            C6_STAMP(1, g + (k_) + 1);                                                                           \
            lds_barrier6();                                                                                      \
        }
#endif
        for (int g = 0, c0 = 0; g < nsteps; g += 3, ++c0) {
            ITER(0, ra1, rb1, ra0, rb0)
            ITER(1, ra2, rb2, ra1, rb1)
            ITER(2, ra0, rb0, ra2, rb2)
        }
        gwait<0>(ra0, rb0); gwait<0>(ra1, rb1); gwait<0>(ra2, rb2);   // (clamped requests past the end are still in flight)
#undef ITER
#undef LOAD_A
#undef STORE_A
#undef LOAD_B
#undef STORE_B
    } else {
    // ---------------------------------------------------------------------- MFMA waves
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h2 = lane >> 5;
    int pbh, pbw, m0;
    C::wave_map(wave, pbh, pbw, m0);
    m0 += mh;                                                       // (MH = 2: this workgroup's half of the 64-row tile)
    const int pxl = (pbw + (li >> 3)) * IHP + pbh + (li & 7);       // LDS entry of this lane's pixel at tap (0,0), pixel block 0
    f32x16 acc[MW][NCT];
    f32x16 accl[SP == 2 ? MW : 1][SP == 2 ? NCT : 1];              // f16 scheme: the cross terms hi*lo' + lo'*hi, 2^11 too large
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; if constexpr (SP == 2) accl[m][n][r] = 0.f; }

    if constexpr (STAMP) { if (A.stamps && tid == 0 && blockIdx.x == 0) { A.stamps[8192 + (A.launch_idx & 127) * 4] = wall_clock64(); A.stamps[8192 + (A.launch_idx & 127) * 4 + 3] = (unsigned long long)(CFG * 1000 + nsteps); } }
    C6_STAMP(0, 0);
    lds_barrier6();                                                 // barrier 0
    for (int c = 0; c < A.nchunk; ++c) {
        // One chunk = 9 taps = 3 steps (kh = 0, 1, 2; a step's taps are kw = 0, 1, 2), fully unrolled.  Fragments of tap T+1
        // are requested before the MFMAs of tap T (register double buffer, alternating with T) -- also across the two
        // barriers inside the chunk: the next step's weights were published one barrier earlier (three A buffers) and the
        // chunk's activations are complete.  Only the chunk's first tap waits for its fragments after a barrier.
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
            const int kh = T / 3, kw = T - 3 * kh;
            if constexpr (SP == 2) {
                // LDS returns data in request order: request in the order the MFMAs consume, so that the first products
                // of the next tap wait for two fragments, not for all of them
                frag_a(T, set, 0, 0); frag_b(T, set, 0, 0); frag_a(T, set, 0, 1); frag_b(T, set, 0, 1);
#pragma unroll
                for (int n = 1; n < NCT; ++n) { frag_b(T, set, n, 0); frag_b(T, set, n, 1); }
#pragma unroll
                for (int m = 1; m < MW; ++m) { frag_a(T, set, m, 0); frag_a(T, set, m, 1); }
                return;
            }
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) bf[set][n][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * NPX + kw * IHP + kh + 8 * n]);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) af[set][m][sp] = __builtin_bit_cast(u32x4, ab[kh * AST + ((kw * 2 + (m0 + m)) * SP + sp) * 64]);
        };
        frags(0, 0);
#pragma unroll
        for (int T = 0; T < 9; ++T) {
            const int cur = T & 1;
            if (T < 8) frags(T + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);              // keep the requests above the MFMAs they overlap with
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NCT; ++n) {
                    if constexpr (SP == 3) {
                        f32x16 a_ = acc[m][n];                      // smallest terms first
                        a_ = mfma_b(af[cur][m][2], bf[cur][n][0], a_);
                        a_ = mfma_b(af[cur][m][0], bf[cur][n][2], a_);
                        a_ = mfma_b(af[cur][m][1], bf[cur][n][1], a_);
                        a_ = mfma_b(af[cur][m][1], bf[cur][n][0], a_);
                        a_ = mfma_b(af[cur][m][0], bf[cur][n][1], a_);
                        a_ = mfma_b(af[cur][m][0], bf[cur][n][0], a_);
                        acc[m][n] = a_;
                    } else {
                        acc[m][n] = mfma_h(af[cur][m][0], bf[cur][n][0], acc[m][n]);
                        f32x16 l_ = accl[m][n];
                        l_ = mfma_h(af[cur][m][1], bf[cur][n][0], l_);
                        l_ = mfma_h(af[cur][m][0], bf[cur][n][1], l_);
                        accl[m][n] = l_;
                    }
                }
            if (T % 3 == 2) {
                const int g = 3 * c + T / 3;
                C6_STAMP(0, g + 1);
                if constexpr (STAMP) {
                    if (A.stamps && A.detail && tid == 0 && (blockIdx.x % 13) == 0 && blockIdx.x / 13 < 8 && g + 65 < 128)
                        A.stamps[((blockIdx.x / 13) * 4 + 0) * 128 + g + 65] = __builtin_readcyclecounter();
                }
                lds_barrier6();                                     // barrier g+1
            }
        }
    }
    C6_STAMP(0, nsteps + 1);
    if constexpr (STAMP) { if (A.stamps && tid == 0 && blockIdx.x == 0) A.stamps[8192 + (A.launch_idx & 127) * 4 + 1] = wall_clock64(); }

    // ---- accumulators -> LDS tile ot[cout][pixel] (the B buffers are free now).  C/D layout: col = lane&31 (pixel),
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (A.out_blk) {                                                // (uniform) pixel-major tile: see OTP
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = acc[m][n][4 * rg + j];
                        if constexpr (SP == 2) v[j] = v[j] * A.descale_hi + accl[m][n][4 * rg + j] * A.descale_lo;      // (powers of two: exact)
                    }
                    *(f32x4*)(ot + ((pbw + (li >> 3)) * TH + pbh + 8 * n + (li & 7)) * OTP + (m0 + m) * 32 + 8 * rg + 4 * h2) = v;
                }
    } else {
#pragma unroll
    for (int n = 0; n < NCT; ++n)
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (m0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h2;
                float v = acc[m][n][r];
                if constexpr (SP == 2) v = v * A.descale_hi + accl[m][n][r] * A.descale_lo;      // (powers of two: exact)
                ot[co * PP + (pbw + (li >> 3)) * TH + pbh + 8 * n + (li & 7)] = v;
            }
    }
    }   // MFMA waves

    // ---- all eight waves: residual adds, ReLU, stores.  Interior rows start on a 128-byte line (PTensor), tiles on a multiple
    // of 8 in h: with H % 4 == 0 every group of four consecutive h is one aligned float4.  The residual operands are requested
    // before the barrier that publishes the LDS tile: one memory latency, overlapped.
    {
        const bool has1 = A.add1 != nullptr, has2 = A.add2 != nullptr;
        bool bad = false;
        float tmax = 0.f;
        if (A.out_blk) {
            // BLOCKED output (and residual operands): the 8 channels of a block at one pixel are 32 contiguous bytes.  A thread takes
            // HALF of such an item (4 channels, 16 bytes), lane pairs take the two halves of one pixel and consecutive pairs consecutive
            // pixels (h fastest): a wave's store / residual request covers whole 512-byte runs (a lane storing both halves would
            // write every line in two half-filled pieces -- measured 2 us slower per 224 x 224 layer), and the LDS reads of a 32-lane
            // group fall on 32 different banks (4 channels further = 16 banks further, PP = 4 mod 32).
            constexpr int NHI = 16 * PXT, HQ = NHI / NT6;           // half-items of the tile; per thread
            static_assert(NHI % NT6 == 0, "epilogue");
            unsigned off[HQ];                                       // float offset of the half-item inside one image, ~0u = outside
            f32x4 r1[HQ], r2[HQ];
#pragma unroll
            for (int k = 0; k < HQ; ++k) {
                const int e2 = k * NT6 + tid, half = e2 & 1, e = e2 >> 1;
                const int g = e / PXT, px = e - g * PXT, w = px / TH, h = px - w * TH;
                const int cb = ct * 8 + g, oh = oh0 + h, ow = ow0 + w;
                const bool ok = cb * 8 < A.Cout && oh < A.H && ow < A.W && (MH == 1 || (g >> 2) == mh);   // (MH = 2: only this workgroup's four blocks)
                off[k] = ok ? (unsigned)(((size_t)cb * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1)) * 8 + 4 * half) : ~0u;
                r1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; r2[k] = r1[k];
            }
            if (has1) {
#pragma unroll
                for (int k = 0; k < HQ; ++k) r1[k] = *(const f32x4*)(A.add1 + (size_t)b * A.add1_bs + ((off[k] != ~0u) ? off[k] : 8u));
            }
            if (has2) {
#pragma unroll
                for (int k = 0; k < HQ; ++k) r2[k] = *(const f32x4*)(A.add2 + (size_t)b * A.add2_bs + ((off[k] != ~0u) ? off[k] : 8u));
            }
            lds_barrier6();                                         // the output tile is complete
#pragma unroll
            for (int k = 0; k < HQ; ++k) {
                const int e2 = k * NT6 + tid, half = e2 & 1, e = e2 >> 1;
                const int g = e / PXT, px = e - g * PXT;
                f32x4 x = *(const f32x4*)(ot + px * OTP + g * 8 + 4 * half);
                x = (x + r1[k]) + r2[k];
                if (A.relu_out) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
                if (off[k] != ~0u) store4(A.out + (size_t)b * A.out_bs + off[k], x, A.wt);
                if constexpr (SP == 2) {
                    const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
                    // (only what is stored counts: rows of the LDS tile this workgroup did not compute -- the other 32-row half, MH = 2 -- hold whatever the
                    //  kernel that had this LDS before left there; with another context's kernels in between that is no longer this network's own weights,
                    //  and the guard raised a false overflow: tests/test_gpu_net.py, two contexts on one device)
                    if (off[k] != ~0u) { bad |= !(gm <= F16_RANGE); tmax = fmaxf(tmax, gm); }                      // (also NaN)
                }
            }
        } else if (A.vec4) {
            constexpr int NG = 64 * PXT / 4, GQ = NG / NT6;         // float4 groups of the tile; per thread
            static_assert(NG % NT6 == 0 && TH % 4 == 0, "epilogue");
            unsigned off[GQ];                                       // element offset inside one image (fits 32 bits), ~0u = outside
            f32x4 r1[GQ], r2[GQ];
#pragma unroll
            for (int k = 0; k < GQ; ++k) {
                const int e = k * NT6 + tid;
                const int co = e / (PXT / 4), rem = e - co * (PXT / 4), w = rem / (TH / 4), h = 4 * (rem - w * (TH / 4));
                const int cog = ct * 64 + co, oh = oh0 + h, ow = ow0 + w;
                const bool ok = cog < A.Cout && oh < A.H && ow < A.W && (MH == 1 || (co >> 5) == mh);
                off[k] = ok ? (unsigned)((size_t)cog * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1)) : ~0u;
                r1[k] = f32x4{0.f, 0.f, 0.f, 0.f}; r2[k] = r1[k];
            }
            if (has1) {                                             // (uniform branches around batches of loads: all in flight together)
#pragma unroll
                for (int k = 0; k < GQ; ++k) r1[k] = *(const f32x4*)(A.add1 + (size_t)b * A.add1_bs + ((off[k] != ~0u) ? off[k] : 1u));
            }
            if (has2) {
#pragma unroll
                for (int k = 0; k < GQ; ++k) r2[k] = *(const f32x4*)(A.add2 + (size_t)b * A.add2_bs + ((off[k] != ~0u) ? off[k] : 1u));
            }
            lds_barrier6();                                         // the output tile is complete
#pragma unroll
            for (int k = 0; k < GQ; ++k) {
                const int e = k * NT6 + tid;
                const int co = e / (PXT / 4), rem = e - co * (PXT / 4);
                f32x4 x = (*(const f32x4*)(ot + co * PP + 4 * rem) + r1[k]) + r2[k];
                if (A.relu_out) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
                if (off[k] != ~0u) store4(A.out + (size_t)ks * A.out_ks + (size_t)b * A.out_bs + off[k], x, A.wt);
                if constexpr (SP == 2) {
                    const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
                    // (only what is stored counts: with one 32-row half computed, the other rows of the LDS tile are whatever the operand buffers held)
                    if (off[k] != ~0u) { bad |= !(gm <= F16_RANGE); tmax = fmaxf(tmax, gm); }                      // (also NaN)
                }
            }
        } else {
            constexpr int NE = 64 * PXT, EQ = NE / NT6;             // tile elements; elements per thread
            static_assert(NE % NT6 == 0, "epilogue");
            unsigned off[EQ];
            float r1[EQ], r2[EQ];
#pragma unroll
            for (int k = 0; k < EQ; ++k) {
                const int e = k * NT6 + tid;
                const int co = e / PXT, rem = e - co * PXT, w = rem / TH, h = rem - w * TH;
                const int cog = ct * 64 + co, oh = oh0 + h, ow = ow0 + w;
                const bool ok = cog < A.Cout && oh < A.H && ow < A.W && (MH == 1 || (co >> 5) == mh);
                off[k] = ok ? (unsigned)((size_t)cog * A.out_plane + (size_t)(ow + 1) * A.out_hp + (oh + 1)) : ~0u;
                r1[k] = 0.f; r2[k] = 0.f;
            }
            if (has1) {
#pragma unroll
                for (int k = 0; k < EQ; ++k) r1[k] = A.add1[(size_t)b * A.add1_bs + ((off[k] != ~0u) ? off[k] : 0u)];
            }
            if (has2) {
#pragma unroll
                for (int k = 0; k < EQ; ++k) r2[k] = A.add2[(size_t)b * A.add2_bs + ((off[k] != ~0u) ? off[k] : 0u)];
            }
            lds_barrier6();                                         // the output tile is complete
#pragma unroll
            for (int k = 0; k < EQ; ++k) {
                const int e = k * NT6 + tid;
                const int co = e / PXT, rem = e - co * PXT;
                float x = (ot[co * PP + rem] + r1[k]) + r2[k];
                if (A.relu_out) x = fmaxf(x, 0.f);
                if (off[k] != ~0u) A.out[(size_t)ks * A.out_ks + (size_t)b * A.out_bs + off[k]] = x;
                if constexpr (SP == 2) { if (off[k] != ~0u) { bad |= !(fabsf(x) <= F16_RANGE); tmax = fmaxf(tmax, fabsf(x)); } }
            }
        }
        if constexpr (SP == 2) {
            if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
            act_report(A.am, tmax, NT6 / 64);
        }
    }
    C6_STAMP(0, nsteps + 2);
    if constexpr (STAMP) { if (A.stamps && tid == 0 && blockIdx.x == 0) A.stamps[8192 + (A.launch_idx & 127) * 4 + 2] = wall_clock64(); }
}

template <int CFG, int SP, bool INB, bool STAMP = false> __global__ __launch_bounds__(NT6) void k_conv6(const Conv6Args A) { conv6_body<CFG, SP, STAMP, INB>(A); }

template <int CFG, int SP>
int launch6(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
            const PTensor* add2, int relu_out, int ksplit = 1, float* partial = nullptr, long out_ks = 0, const ProfSpan* span = nullptr) {
    typedef Cfg6<CFG> C;
    Conv6Args A;
    A.in = in.fbase(); A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.fbase();
    A.add1 = add1 ? add1->fbase() : nullptr; A.add2 = add2 ? add2->fbase() : nullptr;
    A.in_blk = in.blk ? 1 : 0; A.out_blk = (out.blk && !partial) ? 1 : 0;       // (split-K partial sums are planar scratch)
    if ((add1 && add1->blk != out.blk) || (add2 && add2->blk != out.blk) || (out.blk && L.Cout % 8 != 0)) {
        qmri_set_error(ctx, "conv layer %d: residual operands and output must share one tensor format (blocked needs Cout %% 8 == 0)", L.index);
        return QMRI_ERR_STATE;
    }
    A.Cout = L.Cout; A.W = in.W; A.H = in.H;
    A.in_hp = in.hp; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.hp; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.add1_bs = add1 ? (long)add1->Cal * add1->plane() : 0;
    A.add2_bs = add2 ? (long)add2->Cal * add2->plane() : 0;
    A.nchunk = L.nchunk6 / ksplit; A.nchunk_all = L.nchunk6; A.ksplit = ksplit; A.out_ks = out_ks; A.n_ct = L.n_ct6;
    if (partial) { A.out = partial + (out.h0 - 1); A.add1 = A.add2 = nullptr; relu_out = 0; }   // raw partial sums; k_conv6_reduce finishes the layer
    A.tiles_h = (in.H + C::TH - 1) / C::TH; A.tiles_w = (in.W + C::TW - 1) / C::TW;
    A.relu_out = relu_out;
    A.vec4 = (in.H % 4 == 0 && out.h0 % 4 == 0 && out.hp % 4 == 0 && (!add1 || (add1->h0 == out.h0 && add1->hp == out.hp)) &&
              (!add2 || (add2->h0 == out.h0 && add2->hp == out.hp))) ? 1 : 0;
    A.range_flag = ctx->net.d_range_flag;
    A.am = conv6_act_slot(ctx, SP == 2 && !partial, L);             // (split-K partial sums are reported by k_conv6_reduce)
    A.wt = qmri_knob(K_CONV_WT);
    A.xcd = qmri_knob(K_CONV_XCD);
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    A.stamps = (unsigned long long*)ctx->net.d_stamps;
    const int stamp_launch = qmri_knob(K_CONV_STAMP_LAUNCH);
    A.launch_idx = g_conv6_launch_counter.fetch_add(1, std::memory_order_relaxed);
    A.detail = (stamp_launch < 0 || A.launch_idx == stamp_launch) ? 1 : 0;
    const size_t lds = conv6_lds<CFG>(SP);
    if (!ctx->conv6_attr[CFG][SP - 2]) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->conv6_attr[CFG][SP - 2] = true;
    }
    const int grid = C::MH * A.n_ct * A.tiles_h * A.tiles_w * ksplit * B;
    hipEvent_t e0 = nullptr, e1 = nullptr;                          // profile level 2 only: the launch's own dispatch timestamps
    if (span && span->on) e0 = span->start;                         // (split-K: the layer's pair ends on its reduce launch)
    else if (!span) QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_CONV3, conv_layer_flop(L, B, in.H, in.W)));
    if constexpr (SP == 2 && CFG < 2) {
        if (A.stamps) {                                             // knob conv_stamps: the diagnostic instantiation (tools/conv6_stamps.py)
            if (in.blk) {
                QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                k_conv6<CFG, SP, true, true><<<dim3(grid), dim3(NT6), lds, ctx->stream>>>(A);
            } else {
                QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                k_conv6<CFG, SP, false, true><<<dim3(grid), dim3(NT6), lds, ctx->stream>>>(A);
            }
            QMRI_HIP(ctx, hipGetLastError());
            return QMRI_OK;
        }
    }
    if (in.blk) {
        if (e0) hipExtLaunchKernelGGL((k_conv6<CFG, SP, true>), dim3(grid), dim3(NT6), (std::uint32_t)lds, ctx->stream, e0, e1, 0, A);
        else k_conv6<CFG, SP, true><<<dim3(grid), dim3(NT6), lds, ctx->stream>>>(A);
    } else {
        if (e0) hipExtLaunchKernelGGL((k_conv6<CFG, SP, false>), dim3(grid), dim3(NT6), (std::uint32_t)lds, ctx->stream, e0, e1, 0, A);
        else k_conv6<CFG, SP, false><<<dim3(grid), dim3(NT6), lds, ctx->stream>>>(A);
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

template <int CFG>
int launch6(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
            const PTensor* add2, int relu_out, int ksplit = 1, float* partial = nullptr, long out_ks = 0, const ProfSpan* span = nullptr) {
    if constexpr (CFG < 2) {
        if (ksplit == 1 && !partial) {
            bool done = false;
            QMRI_TRY(conv6p_try(ctx, CFG, L, B, in, out, add1, add2, relu_out, &done));
            if (done) return QMRI_OK;
        }
    }
    return (L.sp6 == 2) ? launch6<CFG, 2>(ctx, L, B, in, out, add1, add2, relu_out, ksplit, partial, out_ks, span)
                        : launch6<CFG, 3>(ctx, L, B, in, out, add1, add2, relu_out, ksplit, partial, out_ks, span);
}

// split-K layers: out = relu(sum_k partial_k + add1 + add2), partial sums added in slice order.  VEC: four consecutive h per thread
// as aligned float4 (interior rows start on a 128-byte line, H % 4 == 0) -- the scalar form was launch- and latency-bound at
// 5.5 us for 0.8 M elements.
template <bool VEC>
__global__ __launch_bounds__(256) void k_conv6_reduce(const float* __restrict__ part, int ksplit, long out_ks, float* __restrict__ out,
                                                        const float* __restrict__ add1, const float* __restrict__ add2, long add1_bs,
                                                        long add2_bs, long out_bs, int Cout, int H, int W, int hp, int plane, int relu,
                                                        long total, unsigned* range_flag, ActMax am) {
    constexpr int V = VEC ? 4 : 1;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * V;     // first element of this thread (h fastest)
    float gm = 0.f;
    if (i < total) {                                                // (no early return: act_report reduces across the whole wave)
        const int h = (int)(i % H);
        long r = i / H;
        const int w = (int)(r % W); r /= W;
        const int c = (int)(r % Cout);
        const long b = r / Cout;
        const long o = (long)c * plane + (long)(w + 1) * hp + (h + 1);
        if constexpr (VEC) {
            f32x4 v = *(const f32x4*)(part + b * out_bs + o);
            for (int k = 1; k < ksplit; ++k) v = v + *(const f32x4*)(part + (long)k * out_ks + b * out_bs + o);
            if (add1) v = v + *(const f32x4*)(add1 + b * add1_bs + o);
            if (add2) v = v + *(const f32x4*)(add2 + b * add2_bs + o);
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            store4(out + b * out_bs + o, v, 1);
            gm = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            if (range_flag && !(gm <= F16_RANGE)) atomicOr(range_flag, 1u);
        } else {
            float v = 0.f;
            for (int k = 0; k < ksplit; ++k) v += part[(long)k * out_ks + b * out_bs + o];
            if (add1) v += add1[b * add1_bs + o];
            if (add2) v += add2[b * add2_bs + o];
            if (relu) v = fmaxf(v, 0.f);
            out[b * out_bs + o] = v;
            gm = fabsf(v);
            if (range_flag && !(gm <= F16_RANGE)) atomicOr(range_flag, 1u);
        }
    }
    act_report(am, gm, 4);
}

// The same for BLOCKED output / residual tensors (the partial sums are planar scratch): one half-item (4 channels of a block at one
// pixel, 16 bytes) per thread, lane pairs = the two halves of a pixel, h fastest -- the partial reads are coalesced per channel, the
// stores contiguous.  Same order of additions as above.
__global__ __launch_bounds__(256) void k_conv6_reduce_blk(const float* __restrict__ part, int ksplit, long out_ks, float* __restrict__ out,
                                                            const float* __restrict__ add1, const float* __restrict__ add2, long add1_bs,
                                                            long add2_bs, long out_bs, int Cout, int H, int W, int hp, int plane, int relu,
                                                            long total_half_items, unsigned* range_flag, ActMax am) {
    const long i2 = (long)blockIdx.x * 256 + threadIdx.x;
    float gm = 0.f;
    if (i2 < total_half_items) {
        const int half = (int)(i2 & 1);
        const long i = i2 >> 1;
        const int h = (int)(i % H);
        long r = i / H;
        const int w = (int)(r % W); r /= W;
        const int nblk = Cout / 8;
        const int g = (int)(r % nblk);
        const long b = r / nblk;
        const long px = (long)(w + 1) * hp + (h + 1);
        const float* pp = part + b * out_bs + (long)(8 * g + 4 * half) * plane + px;
        f32x4 x;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = pp[(long)j * plane];
        for (int k = 1; k < ksplit; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] += pp[(long)k * out_ks + (long)j * plane];
        }
        const long bo = ((long)g * plane + px) * 8 + 4 * half;
        if (add1) x = x + *(const f32x4*)(add1 + b * add1_bs + bo);
        if (add2) x = x + *(const f32x4*)(add2 + b * add2_bs + bo);
        if (relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
        store4(out + b * out_bs + bo, x, 1);
        gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
        if (range_flag && !(gm <= F16_RANGE)) atomicOr(range_flag, 1u);
    }
    act_report(am, gm, 4);
}

// end of a forward pass (f16 scheme): per layer, the largest |output| over the slots its kernels reported (act_check_layer, conv6_act.h).
// The PnP-ADMM loop does not launch this: the kernel that follows its forward pass (k_dual_fwd_h) does the same work in its first workgroups.
__global__ __launch_bounds__(256) void k_act_check(ActCheckArgs a) {
    __shared__ float red[4];
    act_check_layer(a, blockIdx.x, red);
}

}  // namespace


// f16 scheme: allocate the |output| slots and the calibrated magnitudes (once per network); reduce the slots after the last layer
int conv6_act_begin(qmri_ctx* ctx, int nlayers) {
    NetPlan& net = ctx->net;
    net.act_on = false;
    if (net.sp6 != 2) return QMRI_OK;
    if (!net.d_act_slots) {
        net.act_cap = nlayers;
        QMRI_HIP(ctx, hipMalloc((void**)&net.d_act_slots, (size_t)net.act_cap * ACT_MAXSLOT * sizeof(float)));
        QMRI_HIP(ctx, hipMalloc((void**)&net.d_act_count, (size_t)net.act_cap * sizeof(int)));
        QMRI_HIP(ctx, hipMalloc((void**)&net.d_act_ref, (size_t)net.act_cap * sizeof(float)));
        QMRI_HIP(ctx, hipMemsetAsync(net.d_act_count, 0, (size_t)net.act_cap * sizeof(int), ctx->stream));
        QMRI_HIP(ctx, hipMemsetAsync(net.d_act_ref, 0, (size_t)net.act_cap * sizeof(float), ctx->stream));   // (0: nothing calibrated, nothing trips)
    }
    net.act_on = true;
    return QMRI_OK;
}
int conv6_act_end(qmri_ctx* ctx) {
    NetPlan& net = ctx->net;
    if (!net.act_on) return QMRI_OK;
    net.act_on = false;
    const bool verbose = qmri_knob(K_VERBOSE) != 0;
    if (verbose) {                                                  // diagnostic: the per-layer maxima of this forward pass
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int> cnt(net.act_cap);
        QMRI_HIP(ctx, hipMemcpy(cnt.data(), net.d_act_count, cnt.size() * sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "libqmri: largest |output| per layer%s:", net.act_record ? " (calibration probe)" : "");
        for (int l = 0; l < net.act_cap; ++l) {
            std::vector<float> v(std::max(cnt[l], 1), 0.f);
            QMRI_HIP(ctx, hipMemcpy(v.data(), net.d_act_slots + (size_t)l * ACT_MAXSLOT, (size_t)cnt[l] * sizeof(float), hipMemcpyDeviceToHost));
            fprintf(stderr, " %.3g", *std::max_element(v.begin(), v.end()));
        }
        fprintf(stderr, "\n");
    }
    const ActCheckArgs ac = {net.d_act_slots, net.d_act_count, net.d_act_ref, net.act_record ? 1 : 0, net.d_range_flag,
                             (net.act_cap + 1 <= net.h_range_words) ? net.h_range_flag : nullptr, net.act_cap};
    if (net.act_defer && !net.act_record) {                         // the caller's next kernel finishes the layers (qmri_pnp_admm_dev)
        net.act_pending = ac;
        net.act_pending_valid = true;
        return QMRI_OK;
    }
    k_act_check<<<dim3(net.act_cap), dim3(256), 0, ctx->stream>>>(ac);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

bool conv6_enabled() {
    const bool on = qmri_knob(K_CONV_F32) <= 0;
    return on;
}

// operand splitting of the matrix-core path: 2 = f16 x 3 products (default), 3 = bf16 x 6 products (knob conv_scheme = 3)
int conv6_default_sp() {
    const int sp = (qmri_knob(K_CONV_SCHEME) == 3) ? 3 : 2;
    return sp;
}

// The f16 scheme carries |w| up to the f16 maximum; larger weights (or non-finite ones) need the bf16 scheme.
bool conv6_weights_fit_f16(const float* w, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!(std::fabs(w[i]) <= F16_RANGE)) return false;
    return true;
}

// Weights (Conv2d OIHW) -> pre-split A fragments:
//   uint4 index = ((((ct64*nchunk + chunk)*9 + tap)*2 + m)*SP + split)*64 + lane ; the uint4 holds 8 pieces, element j:
//   row = ct64*64 + m*32 + (lane&31),  ci = chunk*16 + 8*(lane>>5) + j,  tap = kh*3 + kw
void conv6_plan_pack(ConvLayer& L, const float* w, std::vector<uint16_t>& packed) {
    L.nchunk6 = (L.Cin + CK - 1) / CK;
    L.n_ct6 = (L.Cout + 63) / 64;
    const int SP = L.sp6;
    const float wscale = conv6_weight_scale(L, w, (size_t)L.Cout * L.Cin * 9);
    packed.assign((size_t)L.n_ct6 * L.nchunk6 * 9 * 2 * SP * 64 * 8, 0);
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
                            uint16_t h[3];
                            host_split(SP, v, h, wscale);
                            const size_t base = ((((size_t)ct * L.nchunk6 + chunk) * 9 + tap) * 2 + m) * SP;
                            for (int sp = 0; sp < SP; ++sp) packed[((base + sp) * 64 + lane) * 8 + j] = h[sp];
                        }
}

// The same packing on the device (round 6: qmri_set_denoiser spent 0.5 s splitting and ordering 32.6 M weights on one host thread -- four times the
// reconstruction it serves).  One thread per (cout tile, chunk, tap, m, lane) entry gathers its 8 weights from the flat blob in device memory and
// writes the SP uint4 of its pieces; entries outside the layer's channels are zeros, as conv6_plan_pack leaves them.  Same bits (tested).
namespace {
__global__ __launch_bounds__(256) void k_pack6_w(const float* __restrict__ w, uint4* __restrict__ out, int Cin, int Cout, int nchunk, long nent, int SP, float scale) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= nent) return;
    const int lane = (int)(e & 63), m = (int)((e >> 6) & 1);
    long r = e >> 7;
    const int tap = (int)(r % 9); r /= 9;
    const int chunk = (int)(r % nchunk);
    const int ct = (int)(r / nchunk);
    const int row = ct * 64 + m * 32 + (lane & 31), ci0 = chunk * CK + 8 * (lane >> 5);
    unsigned short h[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ci0 + j;
        const float v = (row < Cout && ci < Cin) ? w[((size_t)row * Cin + ci) * 9 + tap] : 0.f;
        dev_split(SP, v, h[j], scale);
    }
    const long base = ((((long)ct * nchunk + chunk) * 9 + tap) * 2 + m) * SP;
    for (int sp = 0; sp < SP; ++sp) {
        uint4 o;
        o.x = h[0][sp] | ((unsigned)h[1][sp] << 16); o.y = h[2][sp] | ((unsigned)h[3][sp] << 16);
        o.z = h[4][sp] | ((unsigned)h[5][sp] << 16); o.w = h[6][sp] | ((unsigned)h[7][sp] << 16);
        out[(base + sp) * 64 + lane] = o;
    }
}
}  // namespace

// d_w: this layer's weights in the flat blob on the device; wmax: its largest |w| (the scale's input).  Allocates L.wp6.
int conv6_pack_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w, float wmax) {
    L.nchunk6 = (L.Cin + CK - 1) / CK;
    L.n_ct6 = (L.Cout + 63) / 64;
    const float wscale = conv6_scale_from_max(L, wmax);
    const long nent = (long)L.n_ct6 * L.nchunk6 * 9 * 2 * 64;
    QMRI_HIP(ctx, hipMalloc(&L.wp6, (size_t)nent * L.sp6 * sizeof(uint4)));
    k_pack6_w<<<dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, ctx->stream>>>(d_w, (uint4*)L.wp6, L.Cin, L.Cout, L.nchunk6, nent, L.sp6, wscale);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int conv6_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                 const PTensor* add2, int relu_out) {
    // the largest pixel tile that still gives most CUs a workgroup (one workgroup per CU is the design point)
    auto ntiles = [&](int th, int tw) { return (long)L.n_ct6 * ((in.H + th - 1) / th) * ((in.W + tw - 1) / tw) * B; };
    // (tiles may overhang the image -- the kernel masks its stores -- as long as the padded area stays below 1.35 x the image)
    auto waste = [&](int th, int tw) { return (double)(((in.H + th - 1) / th) * th) * (((in.W + tw - 1) / tw) * tw) / ((double)in.H * in.W); };
    if (ntiles(16, 16) >= 160 && waste(16, 16) <= 1.35) return launch6<0>(ctx, L, B, in, out, add1, add2, relu_out);
    if (ntiles(16, 8) >= 160 && waste(16, 8) <= 1.35) return launch6<1>(ctx, L, B, in, out, add1, add2, relu_out);
    // Small feature maps with many channels (the 28 x 28 x 512 level): a 64-pixel tile would re-read the layer's weights
    // once per tile (16 x 14 MB); instead keep the 256-pixel tile and split K over workgroups, then add the partial
    // outputs in slice order (deterministic) in a second, elementwise kernel.
    const bool splitk_on = qmri_knob(K_CONV_SPLITK) != 0;
    // 56 x 56 level: tile config of the split-K variant (0, 1), 2 = no split.  With blocked tensors the unsplit 64-pixel tiles (196
    // workgroups x 48 steps, no reduce launch) win: 696 vs 672 ADMM it/s on one box (round 2; with planar tensors split-K = 2 on
    // 128-pixel tiles + a reduce kernel was the faster form)
    // (3 = 128 pixels x 32 output channels, two workgroups per 64-row weight tile: half the weight bytes through LDS per MFMA of the
    //  64-pixel tile -- these levels' steps are bound by LDS traffic, 4 fragment reads per 3 MFMAs and a full weight step written per
    //  9 MFMAs of a wave: 743 -> 756 ADMM it/s with 3 / 3 / K over 4.  Measured and not kept on the way: one barrier per chunk instead
    //  of per step (six A buffers, whole chunks requested two ahead): 18.3 vs 18.2 us per launch, the barriers are not the bound.)
    const int mid_cfg = qmri_knob(K_CONV_MIDCFG);
    const int deep_cfg_g = qmri_knob(K_CONV_DEEPCFG);              // (28 x 28 level, see below)
    if (splitk_on && L.nchunk6 >= 16) {
        // candidate: the 256-pixel tile (28 x 28 level) or the 128-pixel tile (56 x 56 level), K split so that about one
        // workgroup per CU results and every workgroup still walks >= 4 chunks
        // 28 x 28 level: tile config and largest K split.  Measured with blocked tensors on one box (ADMM it/s, two runs each):
        // 256-pixel tiles x 8 slices (the round-1 choice) 701.6 | x 4: 695 | 128-pixel x 4: 715 | x 2: 695 | 64-pixel x 2: 717.6 | x 1: 679;
        // then, on another box: 64-pixel x 2 (and 64-pixel unsplit at 56 x 56) 743 | configuration 3 at both levels, K over 2: 751.6 | over 4: 756
        const int deep_cfg = deep_cfg_g;
        const int deep_ks = qmri_knob(K_CONV_DEEPKS);
        const bool deep = in.H <= 32;
        const int cfg = deep ? deep_cfg : mid_cfg;
        const long nt = (cfg == 0) ? ntiles(16, 16) : (cfg == 1) ? ntiles(16, 8) : (cfg == 2) ? ntiles(8, 8) : 2 * ntiles(16, 8);
        int ksplit = 1;
        while ((deep || cfg < 2) && cfg >= 0 && ksplit * 2 * nt <= 256 && ksplit * 2 <= (deep ? deep_ks : 8) && L.nchunk6 % (ksplit * 2) == 0 &&
               L.nchunk6 / (ksplit * 2) >= 4)
            ksplit *= 2;
        if (ksplit > 1 && in.H <= 64) {
            const long out_ks = (long)B * out.Cal * out.plane();
            const size_t need = (size_t)ksplit * out_ks + 8192;
            NetPlan& net = ctx->net;
            if (net.c6part_floats < need) {
                if (net.d_c6part) { QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream)); QMRI_HIP(ctx, hipFree(net.d_c6part)); net.d_c6part = nullptr; }
                QMRI_HIP(ctx, hipMalloc((void**)&net.d_c6part, need * sizeof(float)));
                net.c6part_floats = need;
            }
            // profile level 2: ONE unit per layer -- from the convolution's start to the end of the reduce kernel that completes it
            ProfSpan span;
            QMRI_TRY(qmri_prof_pair(ctx, &span.start, &span.stop, PROF_CONV3, conv_layer_flop(L, B, in.H, in.W)));
            span.on = span.start != nullptr;
            if (cfg == 0) QMRI_TRY(launch6<0>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks, &span));
            else if (cfg == 1) QMRI_TRY(launch6<1>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks, &span));
            else if (cfg == 2) QMRI_TRY(launch6<2>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks, &span));
            else QMRI_TRY(launch6<3>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks, &span));
            const long total = (long)B * L.Cout * in.H * in.W;
            const bool vec = in.H % 4 == 0 && out.h0 % 4 == 0 && out.hp % 4 == 0 && (!add1 || (add1->h0 == out.h0 && add1->hp == out.hp)) &&
                             (!add2 || (add2->h0 == out.h0 && add2->hp == out.hp));
            if ((add1 && add1->blk != out.blk) || (add2 && add2->blk != out.blk) || (out.blk && L.Cout % 8 != 0)) {
                qmri_set_error(ctx, "conv layer %d: residual operands and output must share one tensor format", L.index);
                return QMRI_ERR_STATE;
            }
#define REDUCE_ARGS (const float*)(net.d_c6part + (out.h0 - 1)), ksplit, out_ks, out.fbase(), (const float*)(add1 ? add1->fbase() : nullptr),            \
                (const float*)(add2 ? add2->fbase() : nullptr), add1 ? (long)add1->Cal * add1->plane() : 0L, add2 ? (long)add2->Cal * add2->plane() : 0L,   \
                (long)out.Cal * out.plane(), L.Cout, in.H, in.W, out.hp, (int)out.plane(), relu_out, total,                                           \
                (L.sp6 == 2) ? ctx->net.d_range_flag : (unsigned*)nullptr, conv6_act_slot(ctx, L.sp6 == 2, L)
            hipEvent_t const r1 = span.on ? span.stop : nullptr;
            if (out.blk) {
                const long total_items = total / 4;             // half-items
                hipExtLaunchKernelGGL(k_conv6_reduce_blk, dim3((unsigned)((total_items + 255) / 256)), dim3(256), 0, ctx->stream, nullptr, r1, 0,
                    (const float*)(net.d_c6part + (out.h0 - 1)), ksplit, out_ks, out.fbase(), (const float*)(add1 ? add1->fbase() : nullptr),
                    (const float*)(add2 ? add2->fbase() : nullptr), add1 ? (long)add1->Cal * add1->plane() : 0L, add2 ? (long)add2->Cal * add2->plane() : 0L,
                    (long)out.Cal * out.plane(), L.Cout, in.H, in.W, out.hp, (int)out.plane(), relu_out, total_items,
                    (L.sp6 == 2) ? ctx->net.d_range_flag : (unsigned*)nullptr, conv6_act_slot(ctx, L.sp6 == 2, L));
            } else if (vec) hipExtLaunchKernelGGL(k_conv6_reduce<true>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, ctx->stream, nullptr, r1, 0, REDUCE_ARGS);
            else hipExtLaunchKernelGGL(k_conv6_reduce<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, nullptr, r1, 0, REDUCE_ARGS);
#undef REDUCE_ARGS
            QMRI_HIP(ctx, hipGetLastError());
            return QMRI_OK;
        }
    }
    if (L.nchunk6 >= 16 && L.Cout % 64 == 0 && ((in.H <= 32) ? deep_cfg_g : mid_cfg) == 3) return launch6<3>(ctx, L, B, in, out, add1, add2, relu_out);
    return launch6<2>(ctx, L, B, in, out, add1, add2, relu_out);
}

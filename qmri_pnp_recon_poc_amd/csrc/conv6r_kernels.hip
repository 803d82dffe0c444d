// conv6r_kernels.hip -- k_conv6r: a run of 3x3 layers of the full-resolution level as ONE launch with LDS-resident tiles (f16 x 3 scheme;
// scheme and shared device code: conv6_kernels.hip, conv6_device.h).  Reference semantics: the ResBlocks of UNetRes (basicblock.py:211-223,
// network_unet.py:106-117) with the head, the tail and the level's down-sampling convolution where they ride in the launch.
#include "conv6_device.h"

namespace {

// =====================================================================================================================
// k_conv6r : the ResBlocks of the full-resolution level (64 channels) in ONE launch, every workgroup's tile RESIDENT in LDS.
//
// A 3x3 layer launched alone (k_conv6, one slice) is a serial chain: first operands 2 us -> 12 steps at the matrix cores' sustained rate
// (9.4 us) -> epilogue stores 2.3 - 4 us -> kernel boundary 2 us; and the tile a workgroup writes is, but for a one-pixel ring, the tile
// the same workgroup reads in the next layer.  Here a workgroup keeps its 18 x 18 x 64 input tile in LDS as f16 pieces (109 KB next to
// the 36 KB of weight buffers) for all layers of a run of ResBlocks (basicblock.py:211-223) -- with the network's head in front and the level's
// down-sampling convolution behind on the down path, the network's tail behind on the up path, where those apply:
//   * the loop (k_conv6's: same fragments, same MFMA order) reads activations from the resident tile; the loader waves stream WEIGHTS
//     only, straight across layer ends (the next layer's first steps are in LDS before the previous layer's epilogue starts);
//   * the epilogue stays in the matrix waves' registers (descale, + block input, ReLU, range check, f16 split) and writes the pieces IN
//     PLACE into the resident tile (the input is dead once the loop is over); ResBlock outputs also go to memory as fp32 (they are
//     residual operands -- read back by this workgroup alone -- and the run's result), the ReLU intermediates never leave the chip;
//   * the one-pixel ring comes from the eight neighbouring workgroups through memory, as TAGGED GRANULES: the loader waves publish the
//     tile's edges (columns w = 0 / 15, rows h = 0 / 15, four corner pixels: 368 triples of three 16-byte LDS entries) as 8-byte words
//     {3 x f16, 16-bit tag}, four 16-byte sc1 stores per triple, into the tile's part of the exchange buffer of the layer's parity; the
//     matrix waves (idle between two layers) pause, request the matching segments of the eight neighbours, check the eight tags of each
//     triple, write its three entries into the ring and ask again for what was not complete.  No counter, no drain of the stores, no
//     barrier between publish and fetch: a granule is its own flag (MI355X_MICROARCH.md, valid forms, R2).  Two buffers in turn make it
//     race-free: a tile overwrites its layer-l edges at layer l + 2, which it reaches only after its neighbours published layer l + 1,
//     i.e. after they consumed layer l.  The tag is a running count the host never resets.  At the image border nothing is fetched: the
//     ring there keeps the zeros of the run input's halo.
// The launch is a small layer program (Conv6rArgs: per layer the chunks of its input, what its epilogue does, the tensors involved).
// All 196 workgroups must be resident at once (one per CU by LDS size; the host checks tiles <= CUs).  A fetch that is not complete after
// R_SPIN_MAX attempts raises bit 2 of the range flag and the wave runs on without waiting (its workgroup goes on publishing, so nobody
// waits for IT); the host then repeats the call with one launch per layer and keeps this path off (api_net.cpp net_range_tripped).
// Arithmetic, operand order and rounding are those of k_conv6: results are bit-identical (tests/test_gpu_net.py).  DESIGN.md section 5.1
// has the measurements and what was tried on the way.
// =====================================================================================================================
constexpr int R_MAXL = 10;                                  // layers per launch (2 nb ResBlock layers + the network's head or tail)
// what a layer of the launch does behind its loop (Conv6rArgs::kind)
constexpr int R_RELU = 1;                                   // ReLU
constexpr int R_ADD = 2;                                    // + radd[l] (an fp32 BLOCKED tensor, this tile's pixels: a ResBlock's input)
constexpr int R_SKIP = 4;                                   // + skip, after it
constexpr int R_STORE = 8;                                  // the output goes to sdst[l] as fp32 (BLOCKED): a later layer's residual operand
constexpr int R_STORE_WT = 16;                              // ... written through: the run's result
constexpr int R_KEEP = 32;                                  // a layer follows: pieces in place into the resident tile, ring exchange
constexpr int R_DOWN = 128;                                 // the level's strided convolution behind the ResBlocks (2x2 / stride 2, 64 -> 128: k_conv6s DOWN) from the resident tile: no ring needed
constexpr unsigned R_DOWN_STEPB = 2 * 2 * 2 * 64 * 16;           // ... bytes of one of its weight steps (k_conv6s: 2 planes x 2 row tiles x 2 pieces x 64 lanes x 16 B)
constexpr int R_LOCAL = 256;                                // with R_KEEP: the next layer needs no ring (R_DOWN follows): pieces in place, no exchange
constexpr int R_TAIL = 64;                                  // the network's last layer (<= 16 output channels): first 32-row tile of the weights only, PLANAR fp32 output
constexpr int R_IH = 18, R_IW = 18, R_IHP = 24;             // input tile with ring; LDS row pitch (= 8 mod 16 entries, as in k_conv6)
constexpr int R_NPX = R_IHP * (R_IW - 1) + R_IH;            // LDS entries per (split, k-half) plane
constexpr int R_CHUNK = 2 * 2 * R_NPX;                      // ... per 16-channel chunk: [split][k-half][R_NPX]
constexpr int R_AST = ast6(2);
constexpr int R_SPIN_MAX = 1 << 16;
constexpr int R_SEGT = 86, R_CORT = 6;                      // triples per edge segment (16 pixels x 16 entries, padded) / per corner pixel
constexpr int R_NTRI = 4 * R_SEGT + 4 * R_CORT;             // triples (64 bytes each) a tile publishes per layer
constexpr size_t conv6r_lds() { return (size_t)(NABUF * R_AST + 4 * R_CHUNK) * 16; }

struct Conv6rArgs {
    const float* src; const float* skip;                    // fbase of the run's input (BLOCKED 64 channels, or the PLANAR network input: in_planar) and of the skip tensor (or null)
    const float* radd[R_MAXL]; float* sdst[R_MAXL];         // per layer: the operand R_ADD adds, where R_STORE / R_STORE_WT store (BLOCKED fp32 tensors of the level's geometry)
    float* dn_out; int dn_hp, dn_plane;                     // R_DOWN: the BLOCKED output tensor of the next level (fbase), its pitch and plane (elements)
    float* out; int out_hp, out_plane, out_c;               // R_TAIL: the PLANAR output tensor (fbase), its pitch and plane (elements), its channels (<= 16)
    int in_planar, in_plane;                                // the run's first layer is the network's head: src = the PLANAR input (16 channels allocated), its plane (elements)
    int nch[R_MAXL], kind[R_MAXL];                          // 16-channel chunks of the layer's input (1: the head; 4), R_* flags
    unsigned char* xbuf; size_t xbuf_half;                  // exchange buffer [2 layer parities][tiles][R_NTRI][64 bytes]; bytes per parity
    const uint4* wp[R_MAXL];
    float dh[R_MAXL], dl[R_MAXL];                           // descale of the layer's packed weights (Conv6Args::descale_hi / _lo)
    int am_layer[R_MAXL];                                   // row of the |output| report, -1: none
    int nlayers, hp, plane, tiles_h, tiles_w, xcd;
    int drop;                                               // test hook: tile (0, 0) publishes nothing
    int delay;                                              // s_sleep(1) units (64 clocks) between E2 and the first fetch attempt
    unsigned epoch;                                         // layers published before this launch: layer l of this launch tags its granules (epoch + l + 1) mod 2^16
    unsigned* range_flag; float* am_slots; int* am_count;
    unsigned long long* stamps;                             // diagnostic instantiation only
};

template <int N> __device__ __forceinline__ void gwait_a(u32x4 (&a)[3]) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N) : "memory"); }
// (scalar base per channel block + one per-lane offset + an immediate for the pixel block: no per-request address registers)
// (the s_nop behind every store: a VMEM store of more than 8 bytes needs wait states before its data registers are written again, and the hazard
//  recognizer does not look inside inline asm -- without it the next value's arithmetic corrupts the store)
template <int IMM> __device__ __forceinline__ void gload4r_sc1(f32x4& dst, unsigned off, const void* base) { asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(dst) : "v"(off), "s"(base), "n"(IMM) : "memory"); }
template <int IMM> __device__ __forceinline__ void gstore4r(unsigned off, f32x4 x, void* base) { asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base), "n"(IMM) : "memory"); }
template <int IMM> __device__ __forceinline__ void gstore4r_sc1(unsigned off, f32x4 x, void* base) { asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" ::"v"(off), "v"(x), "s"(base), "n"(IMM) : "memory"); }

// triple t of a tile's 368 -> (segment, index inside it); first triple of a segment; LDS byte offset of entry kind k (channel block, piece) at tile position (dw, dh)
__device__ __forceinline__ void r_tri_decode(int t, int& seg, int& j) {
    if (t < 4 * R_SEGT) { seg = t / R_SEGT; j = t - seg * R_SEGT; } else { seg = 4 + (t - 4 * R_SEGT) / R_CORT; j = (t - 4 * R_SEGT) - (seg - 4) * R_CORT; }
}
__device__ __forceinline__ int r_seg_base(int seg) { return seg < 4 ? seg * R_SEGT : 4 * R_SEGT + (seg - 4) * R_CORT; }
__device__ __forceinline__ unsigned r_ent_lds(int k, int dw, int dh) {
    const int cb = k >> 1, sp = k & 1;
    return (unsigned)(((cb >> 1) * R_CHUNK + sp * 2 * R_NPX + (cb & 1) * R_NPX + dw * R_IHP + dh) * 16);
}

// (STAMP: diagnostic instantiation, knob res_stamps -- tools/conv6r_stamps.py: 100 MHz phase stamps of four workgroups, [wg][matrix wave 0 / loader wave 0][layer][8])
#define R_STAMP(role, k)                                                                                         \
    do {                                                                                                         \
        if constexpr (STAMP) {                                                                                   \
            if (A.stamps && (threadIdx.x & 255) == 0 && blockIdx.x % 50 == 0 && blockIdx.x / 50 < 4)             \
                A.stamps[(((blockIdx.x / 50) * 2 + (role)) * R_MAXL + l) * 8 + (k)] = wall_clock64();            \
        }                                                                                                        \
    } while (0)
template <bool STAMP>
__global__ __launch_bounds__(NT6) void k_conv6r(const Conv6rArgs A) {
    constexpr int SP = 2, AST = R_AST, IHP = R_IHP, NPX = R_NPX, NAQ = 3;
    static_assert(AST == NAQ * NLD6, "one weight step = three 16-byte entries per loader thread");
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;                                     // [NABUF][AST]
    uint4* Bt = Abuf + NABUF * AST;                                 // [4 chunks][SP][2 k-halves][NPX]: the resident tile
    const int tid = threadIdx.x;
    const int bid = A.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int th = bid % A.tiles_h, tw = bid / A.tiles_h;
    const int oh0 = th * 16, ow0 = tw * 16;
    const int nl = A.nlayers;
    const unsigned plane32 = (unsigned)A.plane * 32u;               // bytes between channel blocks
    const unsigned tile0 = (unsigned)((ow0 * A.hp + oh0) * 32);     // byte offset of the tile's ring origin (relative to fbase)

    if (tid >= NT6 - NLD6) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - (NT6 - NLD6);
        __builtin_amdgcn_s_setprio(2);
        constexpr unsigned ASTB = AST * 16;
        unsigned aoff[NAQ];
#pragma unroll
        for (int q = 0; q < NAQ; ++q) aoff[q] = (unsigned)((lt + NLD6 * q) * 16);
        // the weight stream: the steps of all layers in a row (3 per 16-channel chunk), requested four steps ahead of the step the matrix waves are in
        int rq_l = 0, rq_s = 0, rq_n = 3 * A.nch[0];
        unsigned rq_stride = ASTB;                                  // bytes between the steps of the layer being requested (R_DOWN: 8 KB steps; its third entry repeats the first)
        u32x4 srdW = make_srd(A.wp[0]);
#define R_REQ(ra_)                                                                                               \
        {                                                                                                        \
            const unsigned so_ = (unsigned)rq_s * rq_stride;                                                     \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) bload4(ra_[q], (q == 2 && rq_stride != ASTB) ? aoff[0] : aoff[q], srdW, so_); \
            if (++rq_s == rq_n) {                                                                                \
                if (rq_l + 1 < nl) { ++rq_l; rq_s = 0; rq_n = 3 * A.nch[rq_l]; srdW = make_srd(A.wp[rq_l]); rq_stride = (A.kind[rq_l] & R_DOWN) ? R_DOWN_STEPB : ASTB; } \
                else rq_s = rq_n - 1;                /* past the end: the last step again (stored where nobody reads) */ \
            }                                                                                                    \
        }
#define R_STORE_A(buf_, ra_)                                                                                     \
        {                                                                                                        \
            uint4* ad = Abuf + (buf_) * AST;                                                                     \
            _Pragma("unroll") for (int q = 0; q < NAQ; ++q) *(uint4*)((unsigned char*)ad + aoff[q]) = __builtin_bit_cast(uint4, ra_[q]); \
        }
        u32x4 ra0[NAQ], ra1[NAQ], ra2[NAQ];
        {
            u32x4 pa1[NAQ];
            R_REQ(ra0) R_REQ(pa1) R_REQ(ra1) R_REQ(ra2)             // steps 0 .. 3
            if (!A.in_planar) {
                // the whole input tile with its ring, fp32 -> pieces: 8 channel blocks x 324 pixels x 2 halves, 21 per thread, all requested at once (one
                // memory latency; the matrix waves wait for this anyway and the loader waves have the registers)
                constexpr int NHALF = 8 * R_IH * R_IW * 2, NPRO = (NHALF + NLD6 - 1) / NLD6;
                f32x4 v[NPRO];
                unsigned lo[NPRO];
#pragma unroll
                for (int k = 0; k < NPRO; ++k) {
                    const int idx = lt + NLD6 * k;
                    const bool valid = idx < NHALF;
                    const int idc = valid ? idx : 0, half = idc & 1, item = idc >> 1;
                    const int cb = item / (R_IH * R_IW), px = item - cb * (R_IH * R_IW), dw = px / R_IH, dh = px - dw * R_IH;
                    gload4r(v[k], (unsigned)cb * plane32 + (unsigned)((dw * A.hp + dh) * 32 + 16 * half) + tile0, A.src);
                    lo[k] = valid ? (unsigned)(((cb >> 1) * R_CHUNK + (cb & 1) * NPX + dw * IHP + dh) * 16 + 8 * half) : ~0u;
                }
                static_assert(NPRO == 21, "the wait below names 21 registers");
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                               "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]), "+v"(v[18]), "+v"(v[19]), "+v"(v[20])::"memory");
#pragma unroll
                for (int k = 0; k < NPRO; ++k) {
                    uint2 s0, s1;
                    split_pair_h(v[k][0], v[k][1], s0.x, s1.x);
                    split_pair_h(v[k][2], v[k][3], s0.y, s1.y);
                    if (lo[k] != ~0u) {
                        *(uint2*)((unsigned char*)Bt + lo[k]) = s0;
                        *(uint2*)((unsigned char*)Bt + lo[k] + 2 * NPX * 16) = s1;
                    }
                }
            } else {
                // the network's head: the tile's first chunk from the PLANAR fp32 input (16 channels allocated, those beyond in_nc zero: PTensor): 2 k-halves x
                // 324 pixels, an item = 8 channels of a pixel = 8 requests one plane apart -> one hi and one lo' entry; the other three chunks are zeroed --
                // the head's epilogue writes their interior, the ring fetch their ring, and at the image border the ring must read as zero
                constexpr int NIT = 2 * R_IH * R_IW, NQ = (NIT + NLD6 - 1) / NLD6;
                static_assert(NQ == 3, "the wait below names 24 registers");
                float v[NQ][8];
                unsigned lo[NQ];
                const unsigned pl4 = (unsigned)A.in_plane * 4u, t0 = (unsigned)((ow0 * A.hp + oh0) * 4);
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    const int idx = lt + NLD6 * k;
                    const bool valid = idx < NIT;
                    const int idc = valid ? idx : 0, kh = idc / (R_IH * R_IW), px = idc - kh * (R_IH * R_IW), dw = px / R_IH, dh = px - dw * R_IH;
#pragma unroll
                    for (int j = 0; j < 8; ++j) gload1(v[k][j], (unsigned)(8 * kh + j) * pl4 + (unsigned)((dw * A.hp + dh) * 4) + t0, A.src);
                    lo[k] = valid ? (unsigned)((kh * NPX + dw * IHP + dh) * 16) : ~0u;
                }
                {                                                   // chunks 1 .. 3 := 0 while the requests are in flight
                    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                    for (int i = lt; i < 3 * R_CHUNK; i += NLD6) Bt[R_CHUNK + i] = z;
                }
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[0][4]), "+v"(v[0][5]), "+v"(v[0][6]), "+v"(v[0][7]), "+v"(v[1][0]),
                               "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[1][4]), "+v"(v[1][5]), "+v"(v[1][6]), "+v"(v[1][7]), "+v"(v[2][0]), "+v"(v[2][1]),
                               "+v"(v[2][2]), "+v"(v[2][3]), "+v"(v[2][4]), "+v"(v[2][5]), "+v"(v[2][6]), "+v"(v[2][7])::"memory");
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    uint4 s0, s1;
                    split_pair_h(v[k][0], v[k][1], s0.x, s1.x);
                    split_pair_h(v[k][2], v[k][3], s0.y, s1.y);
                    split_pair_h(v[k][4], v[k][5], s0.z, s1.z);
                    split_pair_h(v[k][6], v[k][7], s0.w, s1.w);
                    if (lo[k] != ~0u) {
                        *(uint4*)((unsigned char*)Bt + lo[k]) = s0;
                        *(uint4*)((unsigned char*)Bt + lo[k] + 2 * NPX * 16) = s1;
                    }
                }
            }
            gwait_a<0>(ra0); gwait_a<0>(pa1);
            R_STORE_A(0, ra0) R_STORE_A(1, pa1)
        }
        // The ring exchange, per thread and layer: <= 2 TRIPLES to publish and <= 2 to fetch.  A triple = three 16-byte LDS entries (an entry = the 8 hi
        // or the 8 lo' pieces of one channel block at one pixel) = 24 f16 values = eight 8-byte granules {3 x f16, 16-bit tag} = four 16-byte stores.
        // A tile publishes eight segments -- its columns w = 0 / 15, its rows h = 0 / 15 (16 pixels x 16 entries, padded to 86 triples) and its four
        // corner pixels (16 entries, 6 triples) -- into its own 368 x 64 bytes of the exchange buffer of the layer's parity; a tile fetches the
        // matching segments of its eight neighbours (its left ring column = the left neighbour's column w = 15, ...).  Offsets: ~0u = none.
        unsigned p_lds[2][3], p_x[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int t = lt + NLD6 * q;
            const bool valid = t < R_NTRI;
            int seg, j;
            r_tri_decode(valid ? t : 0, seg, j);
            const int npx = seg < 4 ? 16 : 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {                           // segment `seg` of this tile, pixel pp at interior position (w, h)
                const int E = 3 * j + i, Ec = (E < npx * 16) ? E : 0, pp = Ec >> 4, k = Ec & 15;
                const int w = (seg == 0) ? 0 : (seg == 1) ? 15 : (seg == 2 || seg == 3) ? pp : (seg < 6 ? 0 : 15);
                const int h = (seg == 0 || seg == 1) ? pp : (seg == 2) ? 0 : (seg == 3) ? 15 : ((seg & 1) ? 15 : 0);
                p_lds[q][i] = r_ent_lds(k, w + 1, h + 1);
            }
            p_x[q] = valid ? (unsigned)((tw * A.tiles_h + th) * (R_NTRI * 64) + t * 16) : ~0u;   // (quarter i of a triple: + i * R_NTRI * 16 -- consecutive lanes, consecutive 16 bytes)
        }
        lds_barrier6();                                             // barrier 0: step 0 may start
#define R_ITER(k_, rs_, rq_)     /* stores step g + k_ + 2 into its buffer, requests step g + k_ + 4 */          \
        {                                                                                                        \
            __builtin_amdgcn_s_setprio(2);                                                                       \
            R_REQ(rq_)                                                                                           \
            __builtin_amdgcn_s_setprio(0);                                                                       \
            gwait_a<2 * NAQ>(rs_);                                                                               \
            R_STORE_A(((k_) + 2) % NABUF, rs_)                                                                   \
            lds_barrier6();                                                                                      \
        }
#pragma unroll 1
        for (int l = 0; l < nl; ++l) {
#pragma unroll 1
            for (int g = 0; g < 3 * A.nch[l]; g += 3) {
                R_ITER(0, ra1, ra0)
                R_ITER(1, ra2, ra1)
                R_ITER(2, ra0, ra2)
            }
            R_STAMP(1, 0);
            if (A.kind[l] & R_TAIL) {                               // the network's output: out_c planes of this tile from the matrix waves' LDS copy
                lds_barrier6();
                const float* tl = (const float*)Bt;
                const unsigned opx = (unsigned)(((ow0 + (lt >> 4) + 1) * A.out_hp + (oh0 + (lt & 15)) + 1) * 4), opl = (unsigned)A.out_plane * 4u;
                for (int c = 0; c < A.out_c; ++c) {
                    const float x = tl[c * 256 + lt];
                    asm volatile("global_store_dword %0, %1, %2" ::"v"(opx + (unsigned)c * opl), "v"(x), "s"(A.out) : "memory");
                }
                break;
            }
            if (!(A.kind[l] & R_KEEP)) break;                      // (the last layer)
            lds_barrier6();                                         // E2: the matrix waves have written this layer's output into the tile
            if (A.kind[l] & R_LOCAL) continue;                      // (the next layer reads no ring)
            R_STAMP(1, 1);
            const unsigned tag = (A.epoch + (unsigned)l + 1u) & 0xFFFFu, thi = tag << 16;
            unsigned char* xb = A.xbuf + (size_t)(l & 1) * A.xbuf_half;
            if (!(A.drop && th == 0 && tw == 0)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (p_x[q] == ~0u) continue;
                    unsigned d[12];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const uint4 e = *(const uint4*)((const unsigned char*)Bt + p_lds[q][i]);
                        d[4 * i] = e.x; d[4 * i + 1] = e.y; d[4 * i + 2] = e.z; d[4 * i + 3] = e.w;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                   // two granules per store: {F0 F1 | F2 tag} {F3 F4 | F5 tag}
                        u32x4 g;
                        g[0] = d[3 * i];
                        g[1] = (d[3 * i + 1] & 0xFFFFu) | thi;
                        g[2] = (d[3 * i + 1] >> 16) | (d[3 * i + 2] << 16);
                        g[3] = (d[3 * i + 2] >> 16) | thi;
                        asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(p_x[q] + (unsigned)(i * (R_NTRI * 16))), "v"(g), "s"(xb) : "memory");
                    }
                }
            }
            R_STAMP(1, 2);
            R_STAMP(1, 5);
            lds_barrier6();                                         // E3: the tile is the next layer's input
            R_STAMP(1, 6);
        }
        gwait_a<0>(ra0); gwait_a<0>(ra1); gwait_a<0>(ra2);          // (clamped requests past the end are still in flight)
#undef R_ITER
#undef R_REQ
#undef R_STORE_A
        return;
    }
    // ---------------------------------------------------------------------- MFMA waves (tile configuration 0: 64 cout x 64 pixels each, side by side in w)
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h2 = lane >> 5;
    const int pbw = 4 * wave;
    const int pxl = (pbw + (li >> 3)) * IHP + (li & 7);             // LDS entry of this lane's pixel at tap (0,0), pixel block 0
    // byte offset (relative to fbase) of this lane's 4 channels of block 0 at its pixel of pixel block 0; + 8 rows per pixel block, + plane32 per channel block
    const unsigned gpx = (unsigned)(((ow0 + pbw + (li >> 3) + 1) * A.hp + (oh0 + (li & 7)) + 1) * 32 + 16 * h2);
    bool dead = false;                                              // a fetch timed out: no more waiting in this wave
    lds_barrier6();                                                 // barrier 0
#pragma unroll 1
    for (int l = 0; l < nl; ++l) {
        R_STAMP(0, 0);
        // (One scalar base per tensor and per-request offsets made in the epilogue itself: a scalar base per channel block costs 48 SGPRs across the
        //  layer loop -- hipcc then spills SGPRs into VGPR lanes, and a v_readlane reload directly in front of an inline-asm VMEM instruction is a
        //  hazard its recognizer does not see: the first version of this faulted on a garbage address.  The empty asm keeps the offsets out of the
        //  loop-invariant code that would pin 16 VGPRs instead (it stands behind the loop).  tools/audit_conv6_isa.py checks both.)
        f32x16 acc[2][2], accl[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
        const int kind = A.kind[l], nch = A.nch[l];
        // the loop of k_conv6 on the resident tile (MWL = 32-row tiles of the weights a wave multiplies)
#define R_LOOP(MWL)                                                                                              \
        _Pragma("unroll 1") for (int c = 0; c < nch; ++c) {                                                      \
            const uint4* ab = Abuf + lane;                                                                       \
            const uint4* bb = Bt + c * R_CHUNK + h2 * NPX + pxl;                                                 \
            u32x4 bf[2][2][SP], af[2][MWL][SP];                                                                  \
            auto frag_a = [&](int T, int set, int m, int sp) __attribute__((always_inline)) {                    \
                const int kh = T / 3, kw = T - 3 * kh;                                                           \
                af[set][m][sp] = __builtin_bit_cast(u32x4, ab[kh * AST + ((kw * 2 + m) * SP + sp) * 64]);        \
            };                                                                                                   \
            auto frag_b = [&](int T, int set, int n, int sp) __attribute__((always_inline)) {                    \
                const int kh = T / 3, kw = T - 3 * kh;                                                           \
                bf[set][n][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * NPX + kw * IHP + kh + 8 * n]);            \
            };                                                                                                   \
            auto frags = [&](int T, int set) __attribute__((always_inline)) {      /* in the order the MFMAs consume (k_conv6) */ \
                frag_a(T, set, 0, 0); frag_b(T, set, 0, 0); frag_a(T, set, 0, 1); frag_b(T, set, 0, 1);          \
                frag_b(T, set, 1, 0); frag_b(T, set, 1, 1);                                                      \
                if constexpr (MWL == 2) { frag_a(T, set, 1, 0); frag_a(T, set, 1, 1); }                          \
            };                                                                                                   \
            frags(0, 0);                                                                                         \
            _Pragma("unroll") for (int T = 0; T < 9; ++T) {                                                      \
                const int cur = T & 1;                                                                           \
                if (T < 8) frags(T + 1, cur ^ 1);                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                _Pragma("unroll") for (int m = 0; m < MWL; ++m)                                                  \
                    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                              \
                        acc[m][n] = mfma_h(af[cur][m][0], bf[cur][n][0], acc[m][n]);                             \
                        f32x16 l_ = accl[m][n];                                                                  \
                        l_ = mfma_h(af[cur][m][1], bf[cur][n][0], l_);                                           \
                        l_ = mfma_h(af[cur][m][0], bf[cur][n][1], l_);                                           \
                        accl[m][n] = l_;                                                                         \
                    }                                                                                            \
                if (T % 3 == 2) lds_barrier6();     /* end of a step (the last one: every wave is done with the tile) */ \
            }                                                                                                    \
        }
        if (kind & R_DOWN) {
            // The level's down-sampling convolution (Conv2d k = 2, s = 2, 64 -> 128; basicblock.py downsample_strideconv) on the resident tile: the GEMM
            // of k_conv6s<DOWN> -- 8 x 8 output pixels, a step = (16-channel chunk, kw) with the two kh as planes, wave = (32-row tile m0, pixel block),
            // accl += lo x hi, hi x lo; acc += hi x hi per plane -- walked for the two 64-row weight tiles in turn (9 steps each, the ninth all zero:
            // the packed layout of k_conv6s), same order, same bits.  Output: 128 channels x 64 pixels of the next level, straight from the registers.
            const int m0 = wave & 1, pbd = 4 * (wave >> 1);
            const int pxd = (2 * (pbd + (li >> 3)) + 1) * IHP + 2 * (li & 7) + 1;      // LDS entry of input pixel (2 oh, 2 ow) of this lane's output pixel
            const unsigned dpl = (unsigned)A.dn_plane * 32u;
            const unsigned dgo = (unsigned)((((ow0 >> 1) + pbd + (li >> 3) + 1) * A.dn_hp + ((oh0 >> 1) + (li & 7)) + 1) * 32 + 16 * h2);
            const float dh_ = A.dh[l], dl_ = A.dl[l];
            float gmax = 0.f;
            bool bad = false;
            int abi = 0;                                            // A buffer of the step (the steps of all layers rotate through three)
#pragma unroll 1
            for (int ct = 0; ct < 2; ++ct) {
                f32x16 dacc, daccl;
#pragma unroll
                for (int r = 0; r < 16; ++r) { dacc[r] = 0.f; daccl[r] = 0.f; }
#pragma unroll 1
                for (int g = 0; g < 9; ++g) {
                    const int c = (g >> 1) < 4 ? (g >> 1) : 3, kw = g & 1;     // (the ninth step's weights are zero: any chunk)
                    const uint4* ab = Abuf + abi * AST + lane;
                    const uint4* bb = Bt + c * R_CHUNK + h2 * NPX + pxd + kw * IHP;
                    u32x4 bf[2][SP], af[2][SP];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int sp = 0; sp < SP; ++sp) {
                            bf[t][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * NPX + t]);
                            af[t][sp] = __builtin_bit_cast(u32x4, ab[((t * 2 + m0) * SP + sp) * 64]);
                        }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        daccl = mfma_h(af[t][1], bf[t][0], daccl);
                        daccl = mfma_h(af[t][0], bf[t][1], daccl);
                        dacc = mfma_h(af[t][0], bf[t][0], dacc);
                    }
                    abi = (abi == 2) ? 0 : abi + 1;
                    lds_barrier6();
                }
                unsigned go = dgo + (unsigned)(ct * 8 + m0 * 4) * dpl;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 x;
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = __builtin_fmaf(daccl[4 * rg + j], dl_, dacc[4 * rg + j] * dh_);
                    const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
                    bad |= !(gm <= F16_RANGE);
                    gmax = fmaxf(gmax, gm);
                    gstore4r_sc1<0>(go, x, A.dn_out);
                    go += dpl;
                }
            }
            if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
            act_report(ActMax{A.am_slots, A.am_count, A.am_layer[l]}, gmax, 4);
            R_STAMP(0, 1); R_STAMP(0, 2); R_STAMP(0, 3);
            break;
        }
        if (kind & R_TAIL) { R_LOOP(1) } else { R_LOOP(2) }          // (uniform; two copies of the code: a predicate inside the taps costs registers the loop does not have)
#undef R_LOOP
        unsigned gpx_l = gpx;
        int pxl_l = pxl, h2_l = h2;
        asm volatile("" : "+v"(gpx_l), "+v"(pxl_l), "+v"(h2_l));           // (behind the loop: what is derived from them is then made here, not kept across the loop)
        if (kind & R_TAIL) {
            // The network's last layer (64 -> out_c <= 16 channels, no ReLU, no operand): channels 0 .. 15 of the first 32-row tile go through LDS
            // (the tile is dead: [channel][w][h] fp32, 16 KB) to the loader waves, which store the out_c planes of the PLANAR output -- their side of
            // the kernel has the scalar registers for it, this side has not.
            R_STAMP(0, 1);
            R_STAMP(0, 2);
            float* tl = (float*)Bt;
            const float dh_ = A.dh[l], dl_ = A.dl[l];
            float gmax = 0.f;
            bool bad = false;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = 8 * rg + 4 * h2 + j;
                        const float x = __builtin_fmaf(accl[0][n][4 * rg + j], dl_, acc[0][n][4 * rg + j] * dh_);
                        tl[(ch * 16 + pbw + (li >> 3)) * 16 + 8 * n + (li & 7)] = x;
                        const float xa = (ch < A.out_c) ? fabsf(x) : 0.f;
                        bad |= !(xa <= F16_RANGE);
                        gmax = fmaxf(gmax, xa);
                    }
            if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
            act_report(ActMax{A.am_slots, A.am_count, A.am_layer[l]}, gmax, 4);
            lds_barrier6();                                         // the loader waves store it
            R_STAMP(0, 3);
            break;
        }
        // ---- epilogue in registers.  C/D layout: column = lane & 31 = pixel, rows 8 rg + 4 h2 + j = output channels: one lane holds
        // four consecutive channels (half a channel block: cb = 4 m + rg, half h2) of its pixel per (m, n, rg).  Straight-line forms per
        // kind of layer, packed fp32 arithmetic where gfx950 has it.
        R_STAMP(0, 1);
        const f32x2 dh2 = {A.dh[l], A.dh[l]}, dl2 = {A.dl[l], A.dl[l]};
        auto pair = [&](int m, int n, int r) __attribute__((always_inline)) { return f32x2{acc[m][n][r], acc[m][n][r + 1]}; };
        auto pairl = [&](int m, int n, int r) __attribute__((always_inline)) { return f32x2{accl[m][n][r], accl[m][n][r + 1]}; };
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 x = __builtin_elementwise_fma(pairl(m, n, r), dl2, pair(m, n, r) * dh2);      // (powers of two: exact; the sum rounds once, as in k_conv6)
                    acc[m][n][r] = x[0]; acc[m][n][r + 1] = x[1];
                }
        // R_ADD: a ResBlock's second conv adds the block's input, requested once the descaled sums have freed the second accumulator set (requesting it a
        // chunk earlier needs 32 registers the loop does not have: 212 bytes of scratch per lane; right after the loop: 60).  sc1 loads of what this
        // workgroup itself stored two layers ago (or of the run's input): they bypass this CU's L1, which may hold the lines from the previous read of
        // the same addresses.  R_SKIP adds the skip tensor after it (UNetRes.forward, network_unet.py:106-117; requesting both together costs 226 spilled
        // registers: not done).  The order is k_conv6's: (x + block input) + skip.
        f32x4 res[2][2][4];
#define R_REQ_OPERAND(rp_)                                                                                       \
        {                                                                                                        \
            const float* rp__ = (rp_);                                                                           \
            unsigned go = gpx_l;                     /* (a running offset: eight multiples of the plane stride cost eight more scalar registers) */ \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int rg = 0; rg < 4; ++rg) {                                               \
                    gload4r_sc1<0>(res[m][0][rg], go, rp__); gload4r_sc1<256>(res[m][1][rg], go, rp__);          \
                    go += plane32;                                                                               \
                }                                                                                                \
        }
#define R_ADD_OPERAND()         /* x += the requested operand's values at this lane's pixels */                   \
        {                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)"                                                                    \
                         : "+v"(res[0][0][0]), "+v"(res[0][0][1]), "+v"(res[0][0][2]), "+v"(res[0][0][3]), "+v"(res[0][1][0]), "+v"(res[0][1][1]), \
                           "+v"(res[0][1][2]), "+v"(res[0][1][3]), "+v"(res[1][0][0]), "+v"(res[1][0][1]), "+v"(res[1][0][2]), "+v"(res[1][0][3]), \
                           "+v"(res[1][1][0]), "+v"(res[1][1][1]), "+v"(res[1][1][2]), "+v"(res[1][1][3])::"memory");     \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                        \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                    \
                    _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                          \
                        const f32x2 x = pair(m, n, r) + f32x2{res[m][n][r >> 2][r & 3], res[m][n][r >> 2][(r & 3) + 1]}; \
                        acc[m][n][r] = x[0]; acc[m][n][r + 1] = x[1];                                            \
                    }                                                                                            \
        }
        if (kind & R_ADD) { R_REQ_OPERAND(A.radd[l]) R_ADD_OPERAND() }      // (uniform branches)
        if (kind & R_SKIP) { R_REQ_OPERAND(A.skip) R_ADD_OPERAND() }
#undef R_REQ_OPERAND
#undef R_ADD_OPERAND
        R_STAMP(0, 2);
        float gmax = 0.f;                                           // largest |output| of this lane
        bool bad = false;
        // RELU; STORE 0 none / 1 plain (this workgroup reads it back, sc1) / 2 written through (the run's result); KEEP: the next layer's operand, in place
        auto finish = [&](auto relu_c, auto store_c, auto keep_c) __attribute__((always_inline)) {
            constexpr bool RELU = decltype(relu_c)::value, KEEP = decltype(keep_c)::value;
            constexpr int STORE = decltype(store_c)::value;
            float* sd = A.sdst[l];
            unsigned go = gpx_l;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x4 x;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { x[j] = acc[m][n][4 * rg + j]; if constexpr (RELU) x[j] = fmaxf(x[j], 0.f); }
                        const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
                        bad |= !(gm <= F16_RANGE);                  // (also NaN)
                        gmax = fmaxf(gmax, gm);
                        if constexpr (KEEP) {
                            uint2 s0, s1;
                            split_pair_h(x[0], x[1], s0.x, s1.x);
                            split_pair_h(x[2], x[3], s0.y, s1.y);
                            unsigned char* bd = (unsigned char*)Bt + (((2 * m + (rg >> 1)) * R_CHUNK + (rg & 1) * NPX + pxl_l + IHP + 1 + 8 * n) * 16 + 8 * h2_l);
                            *(uint2*)bd = s0;
                            *(uint2*)(bd + 2 * NPX * 16) = s1;
                        }
                        if constexpr (STORE != 0) {
                            if constexpr (STORE == 2) { if (n) gstore4r_sc1<256>(go, x, sd); else gstore4r_sc1<0>(go, x, sd); } else { if (n) gstore4r<256>(go, x, sd); else gstore4r<0>(go, x, sd); }
                        }
                    }
                    go += plane32;
                }
        };
        typedef std::true_type T1; typedef std::false_type T0;
        typedef std::integral_constant<int, 0> S0; typedef std::integral_constant<int, 1> S1; typedef std::integral_constant<int, 2> S2;
        if ((kind & R_RELU) && (kind & R_KEEP)) finish(T1{}, S0{}, T1{});                                 // a ResBlock's first conv
        else if ((kind & R_STORE) && (kind & R_KEEP)) finish(T0{}, S1{}, T1{});                                  // ... its second one (and the head)
        else if (kind & R_STORE_WT) finish(T0{}, S2{}, T0{});                                                   // the run's result
        else finish(T0{}, S0{}, T1{});                                                                          // the layer in front of the tail: nobody else reads it
        if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
        act_report(ActMax{A.am_slots, A.am_count, A.am_layer[l]}, gmax, 4);
        R_STAMP(0, 3);
        if (!(kind & R_KEEP)) break;                                // (the last layer)
        lds_barrier6();                                             // E2
        if (kind & R_LOCAL) continue;                               // (the next layer reads no ring)
        R_STAMP(0, 4);
        {
                // this thread's <= 2 triples of the ring, worked out again for every layer (eight registers less across the loop; fetched by the matrix waves, which have nothing else to do between two layers -- and, unlike the loader
            // waves, no stores of their own in front of the requests): ring segment `seg` <- neighbour (dtw, dth), its segment ns
            int tid_l = tid;
            asm volatile("" : "+v"(tid_l));                        // (per layer, on purpose: as loop invariants the eight offsets are spilled)
            unsigned c_lds[2][3], c_x[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t = tid_l + 256 * q;
                const bool valid = t < R_NTRI;
                int seg, j;
                r_tri_decode(valid ? t : 0, seg, j);
                const int npx = seg < 4 ? 16 : 1;
                const int dtw = (seg == 0 || seg == 4 || seg == 5) ? -1 : (seg == 1 || seg == 6 || seg == 7) ? 1 : 0;
                const int dth = (seg == 2 || seg == 4 || seg == 6) ? -1 : (seg == 3 || seg == 5 || seg == 7) ? 1 : 0;
                const int ns = (seg < 4) ? (seg ^ 1) : 11 - seg;
                const bool have = valid && tw + dtw >= 0 && tw + dtw < A.tiles_w && th + dth >= 0 && th + dth < A.tiles_h;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int E = 3 * j + i, pp = E >> 4, k = E & 15;
                    const int dw = (dtw < 0) ? 0 : (dtw > 0) ? 17 : pp + 1, dh = (dth < 0) ? 0 : (dth > 0) ? 17 : pp + 1;
                    c_lds[q][i] = (have && E < npx * 16) ? r_ent_lds(k, dw, dh) : ~0u;
                }
                c_x[q] = have ? (unsigned)(((tw + dtw) * A.tiles_h + th + dth) * (R_NTRI * 64) + (r_seg_base(ns) + j) * 16) : ~0u;
            }
            const unsigned tag = (A.epoch + (unsigned)l + 1u) & 0xFFFFu, thi = tag << 16;
            const unsigned char* xb = A.xbuf + (size_t)(l & 1) * A.xbuf_half;
            // the neighbours publish about now and their stores need ~1 us to be visible: requests sent at once only find old tags -- and 196 x 256
            // lanes re-reading 17 KB each slow the stores they wait for (measured per forward: 1117 us without the pause, 1094 - 1104 with 32 - 48 units)
            for (int i = 0; i < A.delay; ++i) __builtin_amdgcn_s_sleep(1);     // (timed against the 100 MHz clock instead -- the chip's own varies between boxes -- it was slower at every setting: the clock reads of 784 waves are traffic of their own)
            {
                bool pend[2] = {c_x[0] != ~0u && !dead, c_x[1] != ~0u && !dead};
                bool ok = false;
                for (int spin = 0; spin < R_SPIN_MAX; ++spin) {
                    u32x4 g[2][4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (pend[q]) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(g[q][i]) : "v"(c_x[q] + (unsigned)(i * (R_NTRI * 16))), "s"(xb) : "memory");
                        }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[0][2]), "+v"(g[0][3]), "+v"(g[1][0]), "+v"(g[1][1]), "+v"(g[1][2]), "+v"(g[1][3])::"memory");
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (pend[q]) {
                            unsigned bad = 0;
#pragma unroll
                            for (int i = 0; i < 4; ++i) bad |= (g[q][i][1] ^ thi) | (g[q][i][3] ^ thi);
                            if ((bad >> 16) == 0) {                 // all eight granules carry this layer's tag: the three entries are complete
                                unsigned d[12];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    d[3 * i] = g[q][i][0];
                                    d[3 * i + 1] = (g[q][i][1] & 0xFFFFu) | (g[q][i][2] << 16);
                                    d[3 * i + 2] = (g[q][i][2] >> 16) | (g[q][i][3] << 16);
                                }
#pragma unroll
                                for (int i = 0; i < 3; ++i)
                                    if (c_lds[q][i] != ~0u) *(uint4*)((unsigned char*)Bt + c_lds[q][i]) = make_uint4(d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
                                pend[q] = false;
                            }
                        }
                    if (!__any(pend[0] || pend[1])) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && !dead) { dead = true; if (lane == 0 && A.range_flag) atomicOr(A.range_flag, 4u); }
            }
        }
        R_STAMP(0, 5);
        lds_barrier6();                                             // E3
        R_STAMP(0, 6);
    }
}

}  // namespace

// A run of the full-resolution level's 3x3 layers as ONE launch with resident tiles (k_conv6r, Conv6rRun in qmri_internal.h).  *done = false: not
// eligible, nothing launched.
int conv6r_try(qmri_ctx* ctx, const Conv6rRun& run, int B, bool* done) {
    *done = false;
    NetPlan& net = ctx->net;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;       // (a captured launch would replay stale tags: never under a stream capture)
    const bool capturing = hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const int nres = run.nres, nl = nres + (run.head ? 1 : 0) + (run.tail ? 1 : 0) + (run.down ? 1 : 0);
    if (!qmri_knob(K_CONV_RESIDENT) || capturing || net.res_off || !net.d_res_xbuf || B != 1 || nres < 2 || nl > R_MAXL || (nres & 1) || net.d_stamps) return QMRI_OK;
    if (!run.res || !run.src || !run.cur || (run.head && (!run.head_in || run.skip)) || (run.tail && !run.tail_out) || (run.down && (!run.down_out || run.tail || run.skip))) return QMRI_OK;
    auto is3 = [](const ConvLayer& L) { return (L.kind == CONV_3X3 || L.kind == CONV_3X3N) && L.sp6 == 2 && L.wp6 && L.n_ct6 == 1; };
    for (int l = 0; l < nres; ++l) {
        const ConvLayer& L = run.res[l];
        if (!is3(L) || L.Cin != 64 || L.Cout != 64 || L.nchunk6 != 4) return QMRI_OK;
    }
    const PTensor &src = *run.src, &cur = *run.cur;
    const PTensor* ts[3] = {&src, &cur, run.skip};
    for (const PTensor* t : ts) {
        if (!t) continue;
        if (!t->p || !t->blk || t->Cal < 64 || t->H != src.H || t->W != src.W || t->hp != src.hp || t->h0 != src.h0) return QMRI_OK;
    }
    if (run.head) {                                                 // in_nc -> 64 from the PLANAR network input: one 16-channel chunk
        const ConvLayer& L = *run.head;
        const PTensor& in = *run.head_in;
        if (!is3(L) || L.Cout != 64 || L.nchunk6 != 1 || L.Cin > 16 || !in.p || in.blk || in.Cal < 16 || in.H != src.H || in.W != src.W || in.hp != src.hp ||
            in.h0 != src.h0 || src.p == cur.p) return QMRI_OK;
    }
    if (run.tail) {                                                 // 64 -> out_nc <= 16 to the PLANAR network output
        const ConvLayer& L = *run.tail;
        const PTensor& out = *run.tail_out;
        if (!is3(L) || L.Cin != 64 || L.nchunk6 != 4 || L.Cout > 16 || !out.p || out.blk || out.H != src.H || out.W != src.W ||
            (size_t)out.Cal * out.plane() * 4 >= ((size_t)1 << 31)) return QMRI_OK;
    }
    if (run.down) {                                                 // 64 -> 128, 2x2 / stride 2, to the next level's BLOCKED tensor (the packed weights of k_conv6s DOWN: 2 tiles x 9 steps)
        const ConvLayer& L = *run.down;
        const PTensor& out = *run.down_out;
        if (L.kind != CONV_DOWN || L.sp6 != 2 || !L.wp6 || L.Cin != 64 || L.Cout != 128 || L.nchunk6 != 9 || L.n_ct6 != 2 || !out.p || !out.blk || out.Cal < 128 ||
            out.H * 2 != src.H || out.W * 2 != src.W || (size_t)out.Cal * out.plane() * 4 >= ((size_t)1 << 31)) return QMRI_OK;
    }
    if (src.H % 16 || src.W % 16) return QMRI_OK;
    if ((size_t)src.Cal * src.plane() * 4 >= ((size_t)1 << 31)) return QMRI_OK;       // (32-bit byte offsets)
    if (!ctx->conv_ncu) {
        hipDeviceProp_t prop;
        QMRI_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->conv_ncu = prop.multiProcessorCount;
    }
    const int tiles_h = src.H / 16, tiles_w = src.W / 16, tiles = tiles_h * tiles_w;
    if (tiles > ctx->conv_ncu || tiles != net.res_tiles) return QMRI_OK;               // every workgroup must be resident: one per CU
    Conv6rArgs A{};
    A.src = run.head ? run.head_in->fbase() : src.fbase();
    A.in_planar = run.head ? 1 : 0; A.in_plane = run.head ? (int)run.head_in->plane() : 0;
    A.skip = run.skip ? run.skip->fbase() : nullptr;
    if (run.down) { A.dn_out = run.down_out->fbase(); A.dn_hp = run.down_out->hp; A.dn_plane = (int)run.down_out->plane(); }
    if (run.tail) { A.out = run.tail_out->fbase(); A.out_hp = run.tail_out->hp; A.out_plane = (int)run.tail_out->plane(); A.out_c = run.tail->Cout; }
    int l = 0;
    auto put = [&](const ConvLayer& L, int kind, const float* radd, float* sdst) {
        A.wp[l] = reinterpret_cast<const uint4*>(L.wp6);
        A.dh[l] = L.w6_descale; A.dl[l] = L.w6_descale * (1.f / LO_SCALE);
        A.am_layer[l] = conv6_act_slot(ctx, true, L).layer;
        A.nch[l] = L.nchunk6; A.kind[l] = kind; A.radd[l] = radd; A.sdst[l] = sdst;
        ++l;
    };
    if (run.head) put(*run.head, R_STORE | R_KEEP, nullptr, src.fbase());             // x1 = m_head(x0): also the up path's skip tensor
    for (int r = 0; r < nres; ++r) {
        const bool last = r == nres - 1;
        if (!(r & 1)) put(run.res[r], R_RELU | R_KEEP, nullptr, nullptr);
        else if (!last) put(run.res[r], R_ADD | R_STORE | R_KEEP, (r == 1) ? src.fbase() : cur.fbase(), cur.fbase());
        else if (run.tail) put(run.res[r], R_ADD | (run.skip ? R_SKIP : 0) | R_KEEP, (r == 1) ? src.fbase() : cur.fbase(), nullptr);   // only the tail reads it
        else if (run.down) put(run.res[r], R_ADD | R_KEEP | R_LOCAL, (r == 1) ? src.fbase() : cur.fbase(), nullptr);               // only the down conv reads it, and no ring of it
        else put(run.res[r], R_ADD | (run.skip ? R_SKIP : 0) | R_STORE_WT, (r == 1) ? src.fbase() : cur.fbase(), cur.fbase());
    }
    if (run.tail) put(*run.tail, R_TAIL, nullptr, nullptr);
    if (run.down) { put(*run.down, R_DOWN, nullptr, nullptr); A.nch[l - 1] = 6; }      // (18 steps of 8 KB)
    for (int k = l; k < R_MAXL; ++k) { A.wp[k] = A.wp[l - 1]; A.nch[k] = A.nch[l - 1]; A.am_layer[k] = -1; }
    // Tags are 16 bits on the wire ((epoch + l + 1) mod 2^16) and the epoch is a running count: a granule can only be mistaken for a current one if
    // it survived, un-rewritten, for 2^16 tags.  What rules that out is that EVERY launch rewrites EVERY slot: a publishing layer l (R_KEEP without
    // R_LOCAL) rewrites all slots of parity l & 1 of every tile, so a launch must publish on both parities -- a layer program that does not is not
    // run in this form.  (The test hook's silent tile leaves tags of the launch before: at most two launches old, never 2^16 tags.)
    int pub[2] = {0, 0};
    for (int k = 0; k < l; ++k) if ((A.kind[k] & R_KEEP) && !(A.kind[k] & R_LOCAL)) pub[k & 1] += 1;
    if (!pub[0] || !pub[1]) return QMRI_OK;
    A.nlayers = nl; A.hp = src.hp; A.plane = (int)src.plane(); A.tiles_h = tiles_h; A.tiles_w = tiles_w;
    A.xbuf = net.d_res_xbuf; A.xbuf_half = (size_t)tiles * R_NTRI * 64;
    A.xcd = qmri_knob(K_CONV_XCD);
    const int delay = std::max(0, std::min(4096, qmri_knob(K_RES_DELAY)));
    A.epoch = net.res_epoch; A.drop = net.res_drop; A.delay = delay;
    A.range_flag = net.d_range_flag; A.am_slots = net.d_act_slots; A.am_count = net.d_act_count;
    if (!ctx->conv6r_attr) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6r<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6r_lds()));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6r<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6r_lds()));
        ctx->conv6r_attr = true;
    }
    g_conv6_launch_counter.fetch_add(nl, std::memory_order_relaxed);
    const int stamp_which = qmri_knob(K_RES_STAMPS);               // 1: the up path's launch, 2: the down path's (the one with the head)
    A.stamps = (stamp_which == 2) == (run.head != nullptr) ? (unsigned long long*)net.d_res_stamps : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // (profile level 2: ONE unit for the launch, with the work of every layer inside it -- ResBlock layers, head, tail, down-sampling convolution)
    double flop = 0.0;
    for (int r = 0; r < nres; ++r) flop += conv_layer_flop(run.res[r], 1, src.H, src.W);
    if (run.head) flop += conv_layer_flop(*run.head, 1, src.H, src.W);
    if (run.tail) flop += conv_layer_flop(*run.tail, 1, src.H, src.W);
    if (run.down) flop += conv_layer_flop(*run.down, 1, src.H, src.W);
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_CONV3, flop));
    if (A.stamps) k_conv6r<true><<<dim3(tiles), dim3(NT6), conv6r_lds(), ctx->stream>>>(A);
    else if (e0) hipExtLaunchKernelGGL((k_conv6r<false>), dim3(tiles), dim3(NT6), (std::uint32_t)conv6r_lds(), ctx->stream, e0, e1, 0, A);
    else k_conv6r<false><<<dim3(tiles), dim3(NT6), conv6r_lds(), ctx->stream>>>(A);
    QMRI_HIP(ctx, hipGetLastError());
    net.res_epoch += (unsigned)(nl - 1);                            // one tag per layer but the last; a running count (see the note on tags above)
    *done = true;
    return QMRI_OK;
}

size_t conv6r_xbuf_bytes(int tiles) { return (size_t)2 * tiles * R_NTRI * 64; }

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
//   SP = 3, "bf16 x 6" (QMRI_CONV_SCHEME=bf16x6, and the fallback): x = x0 + x1 + x2 exactly (8 + 8 + 8 mantissa bits, no range
//       limit); of the nine piece products the six of order >= 2^-16,  w0 a0 + (w0 a1 + w1 a0) + (w1 a1 + w0 a2 + w2 a0),
//       are accumulated (8.6e-8 on the same tile).  Twice the matrix-core cycles of SP = 2.
//
// Implicit GEMM per workgroup: 64 output channels x (TH x TW) pixels, K = Cin*9 walked in chunks of 16 channels x 3 taps.
//   waves 0-3  MFMA: per tap SP A fragments per cout tile (weights, pre-split and pre-ordered on the host) and SP B
//              fragments per pixel block (activations) from LDS feed 3 (6) MFMAs per 32x32 tile
//   waves 4-7  loaders: weights global -> LDS (plain copy), activations global fp32 planes -> split -> LDS [pixel][8 ch],
//              requested two steps ahead and kept in registers for one
// Tensors stay fp32 padded planes in HBM (qmri_internal.h PTensor), so this kernel is interchangeable with k_conv.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "qmri_internal.h"
#include "conv6_act.h"
#include <hip/hip_ext.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (register arrays of HIP's uint4 struct are not promoted out of scratch)

constexpr int NT6 = 512;         // threads per workgroup: 4 MFMA waves + 4 loader waves
constexpr int NLD6 = 256;        // loader threads
constexpr int CK = 16;           // input channels per chunk = K of one MFMA
// SP = pieces an fp32 operand is split into: 3 (bf16 x 6 products) or 2 (f16 x 3 products), see the header comment
constexpr int ast6(int SP) { return 3 * 2 * SP * 64; }   // uint4 per step of A: 3 taps x 2 cout tiles x SP splits x 64 lanes
constexpr float LO_SCALE = 2048.f;    // f16 scheme: the low piece is stored as (x - hi) * 2^11, so it is normal whenever x is

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b + 8 share one, MI355X_MICROARCH.md), each with its own L2.  Tiles
// that are neighbours in memory -- the cout tiles of one pixel tile read the same activations, vertically adjacent pixel tiles share
// their halo rows' 128-byte lines -- are consecutive in tile order, so the default order puts them on eight different L2s: measured
// 28.3 MB fetched per 224 x 224 layer for 16.3 MB of input (rocprofv3 FETCH_SIZE).  The remap gives every XCD one contiguous range
// of tiles (bijective for any n; placement only ever changes speed).
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct ActMax { float* slots; int* count; int layer; };   // where a launch reports its largest |output| (see ACT_LOW); layer < 0: it does not

struct Conv6Args {
    const float* in; const uint4* wp; float* out; const float* add1; const float* add2;
    int Cout, W, H;
    int in_hp, in_plane; long in_bs;          // padded row pitch, plane size, batch stride (elements)
    int out_hp, out_plane; long out_bs, add1_bs, add2_bs;
    int nchunk, n_ct, tiles_h, tiles_w, relu_out;
    int vec4;                     // epilogue may use aligned float4 accesses (H % 4 == 0 and line-aligned tensors)
    int wt;                       // write-through (sc1) output stores, see store4()
    int xcd;                      // XCD-aware tile order, see xcd_remap()
    int in_blk, out_blk;          // the input / the output (and with it the residual operands) is a BLOCKED tensor [c/8][w][h][8] (see BRegs)
    int lowhalf;                  // Cfg6<3> only: launch the first 32-row half of every 64-row tile alone (Cout <= 32)
    int in_pcs, out_pcs;          // ... whose 32-byte items hold the f16 PIECES of the 8 channels, [8 x hi][8 x lo'], instead of 8 floats (see pieces8)
    int nchunk_all, ksplit;       // split-K: this launch covers nchunk of the layer's nchunk_all chunks per workgroup, ksplit workgroups per tile
    long out_ks;                  // split-K: elements between the partial outputs of consecutive K slices
    int ntiles;                   // k_conv6p: tiles of the launch (n_ct * tiles_h * tiles_w * B)
    int launch_idx, detail;       // diagnostic: running launch number; record the per-step stamps of this launch
    unsigned* range_flag;         // f16 scheme: set to 1 when an output leaves the range the next layer's f16 split can carry
    ActMax am;                    // f16 scheme: where this launch reports the largest |output| (see ACT_LOW)
    float descale_hi, descale_lo; // f16 scheme: the layer's weights are packed times 2^k (largest |w| in [1, 2)): 2^-k and 2^-k / 2^11
    unsigned long long* stamps;   // diagnostic (QMRI_CONV_STAMPS): [16 workgroups][2 roles][128] barrier-arrival times, 100 MHz clock
};

// Workgroup tile = 64 output channels x (TH x TW) pixels.  A wave owns MW cout tiles (32 rows) x NCT pixel blocks of
// 8h x 4w (stacked in h); wave_map gives its first pixel block and first cout tile.
template <int CFG> struct Cfg6;
template <> struct Cfg6<0> {     // 256 px, waves side by side in w, 64 cout x 64 px each
    static constexpr int TH = 16, TW = 16, MW = 2, NCT = 2, MH = 1;
    static __device__ __forceinline__ void wave_map(int wave, int& pbh, int& pbw, int& m0) { pbh = 0; pbw = 4 * wave; m0 = 0; }
};
template <> struct Cfg6<1> {     // 128 px, waves 2 x 2, 64 cout x 32 px each
    static constexpr int TH = 16, TW = 8, MW = 2, NCT = 1, MH = 1;
    static __device__ __forceinline__ void wave_map(int wave, int& pbh, int& pbw, int& m0) { pbh = 8 * (wave & 1); pbw = 4 * (wave >> 1); m0 = 0; }
};
template <> struct Cfg6<2> {     // 64 px, waves = 2 cout halves x 2 pixel blocks, 32 cout x 32 px each
    static constexpr int TH = 8, TW = 8, MW = 1, NCT = 1, MH = 1;
    static __device__ __forceinline__ void wave_map(int wave, int& pbh, int& pbw, int& m0) { pbh = 0; pbw = 4 * (wave >> 1); m0 = wave & 1; }
};
template <> struct Cfg6<3> {     // 128 px x 32 cout: the workgroup takes ONE 32-row half (MH = 2 workgroups per 64-row tile) of the weights --
                                 // half the weight bytes per MFMA of the 64-pixel tile, for the deep levels where the step is bound by them
    static constexpr int TH = 16, TW = 8, MW = 1, NCT = 1, MH = 2;
    static __device__ __forceinline__ void wave_map(int wave, int& pbh, int& pbw, int& m0) { pbh = 8 * (wave & 1); pbw = 4 * (wave >> 1); m0 = 0; }
};
// Output tile in LDS for BLOCKED output tensors: PIXEL-major, ot[pixel][OTP] with the tile's 64 output channels of a pixel contiguous (round 3).
// In the MFMA C/D layout a lane's four consecutive registers are four consecutive output channels of one pixel, and an epilogue thread's
// half-item is four consecutive channels of one pixel: one 16-byte LDS access on either side instead of four 4-byte ones (channel-major
// ot[cout][pixel] needed 128 ds_write_b32 per matrix wave and tile; it stays the layout of PLANAR outputs, whose threads take four pixels of a
// channel).  Pitch 68 floats: 16-byte aligned, 8-lane write groups on 32 distinct banks.
constexpr int OTP = 68;
constexpr int NABUF = 3;         // LDS buffers of A (one step each): step g lives in buffer g % 3 = its kh; a step's weights are complete one
                                 // barrier before the step starts, so the MFMA waves can request its first fragments across that barrier

// (STAMP: diagnostic instantiation only -- the production kernels carry no stamp code)
#define C6_STAMP(role, k)                                                                        \
    do {                                                                                         \
        if constexpr (STAMP) {                                                                   \
            if (A.stamps && A.detail && (threadIdx.x & 255) == 0 && (blockIdx.x % 13) == 0 && blockIdx.x / 13 < 8 && (k) < 128)   \
                A.stamps[((blockIdx.x / 13) * 4 + (role)) * 128 + (k)] = wall_clock64();        \
        }                                                                                        \
    } while (0)

__device__ __forceinline__ void lds_barrier6() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ f32x16 mfma_b(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
#ifdef C6_NO_MFMA       // (timing only, with -DQMRI_TIMING_ONLY: the fragments are still read -- the operands stay live)
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }

// Loader waves request their operands with inline-asm loads and wait with hand-counted s_waitcnt: hipcc's own counter
// insertion drains vmcnt almost completely at the loop header, which exposes a full memory latency per step.  A loader
// wave issues no other vector-memory instruction, loads complete in issue order, and every wait names the registers it
// releases ("+v"), so no consumer can be scheduled above it.
// (scalar base + 32-bit per-lane byte offset: the offsets are loop invariant, the base advances per step)
__device__ __forceinline__ void gload4(u32x4& dst, unsigned off, const void* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory"); }
// Buffer forms: descriptor (4 SGPRs) per tensor, per-lane byte offset (VGPR) + scalar byte offset (SGPR): a request costs the
// wave ONE instruction plus whatever the scalar offset costs, instead of a 64-bit pointer per request.  num_records = 2^32 - 1:
// the range check sees only the per-lane offset; the tensors here are far below 4 GB (conv6_launch checks).
__device__ __forceinline__ u32x4 make_srd(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    u32x4 r;
    r[0] = (unsigned)v; r[1] = (unsigned)(v >> 32) & 0xFFFFu; r[2] = 0xFFFFFFFFu; r[3] = 0x00020000u;
    return r;
}
// a wave-uniform 32-bit value the compiler may hold in a VGPR -> SGPR (the "s" operands below); the s_nop covers the 5 wait
// states between a VALU write of an SGPR and a vector-memory instruction reading it
__device__ __forceinline__ unsigned usgpr(unsigned v) {
    unsigned r = __builtin_amdgcn_readfirstlane(v);
    asm volatile("s_nop 4" : "+s"(r));
    return r;
}
__device__ __forceinline__ void bload4(u32x4& dst, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void bload4f(f32x4& dst, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void bload4f_o16(f32x4& dst, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void bstore4_o16(f32x4 x, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:16 sc1\n\ts_nop 1" ::"v"(x), "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void bload1(float& dst, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void bstore4(f32x4 x, unsigned voff, u32x4 srd, unsigned soff) { asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(x), "v"(voff), "s"(srd), "s"(soff) : "memory"); }
__device__ __forceinline__ void gload4r(f32x4& dst, unsigned off, const void* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void gload1(float& dst, unsigned off, const void* base) { asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory"); }
// One loader item of the B operand = 8 input channels of one pixel.  PLANAR tensors ([c][w][h], h fastest): eight 4-byte requests, one
// per channel plane.  BLOCKED tensors ([c/8][w][h][8], the interior format of the matrix-core schemes): the eight
// channels are 32 contiguous bytes, two 16-byte requests -- a vector-memory instruction costs a loader wave 25-60 cycles of issue
// whatever its width, and the loader waves' issue time bounds the loop (tools/conv6p_stamps.py).
template <bool INB> struct BRegs;
template <> struct BRegs<false> { float v[8]; __device__ __forceinline__ float get(int j) const { return v[j]; } };
template <> struct BRegs<true> { f32x4 q[2]; };   // two HALF-items (4 channels, 16 bytes) of different pixels: lane pairs take the two halves
                                                   // of one pixel, so a wave's request covers contiguous runs (as the epilogue's stores do)
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[5], BRegs<false>& b) {
    asm volatile("s_waitcnt vmcnt(%13)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]),
                   "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7])
                 : "n"(N)
                 : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[3], BRegs<false>& b) {
    asm volatile("s_waitcnt vmcnt(%11)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]),
                   "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7])
                 : "n"(N)
                 : "memory");
}
// the small tiles (128 / 64 pixels) need only ONE half-item per loader thread and step
struct BRegs1 { f32x4 q[1]; };
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[2], BRegs1& b) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b.q[0]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[2], BRegs<false>& b) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]), "+v"(b.v[3]), "+v"(b.v[4]), "+v"(b.v[5]), "+v"(b.v[6]), "+v"(b.v[7])
                 : "n"(N)
                 : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[5], BRegs1& b) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b.q[0]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[3], BRegs1& b) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b.q[0]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[5], BRegs<true>& b) {
    asm volatile("s_waitcnt vmcnt(%7)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b.q[0]), "+v"(b.q[1]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void gwait(u32x4 (&a)[3], BRegs<true>& b) {
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b.q[0]), "+v"(b.q[1]) : "n"(N) : "memory");
}

// Output stores.  A plain store leaves its line dirty in the XCD's L2, and the end-of-kernel release then writes all of them back
// before the next (dependent) kernel may start: 12.8 MB per layer at the 224 x 224 level, i.e. a kernel boundary of 3.3 us instead
// of the 1.7-1.9 us of a boundary with nothing dirty (MI355X_MICROARCH.md, price list row "boundary").  Write-through (sc1) stores
// send the bytes to memory as they are issued -- while other workgroups still compute -- and leave nothing for the boundary.
__device__ __forceinline__ void store4(float* p, f32x4 x, int wt) {
#ifdef C6_NO_STORES     // (timing only: k_conv6's output stores dropped)
    asm volatile("" :: "v"(x), "v"(p)); return;
#endif
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
    else *(f32x4*)p = x;
}

// wave-uniform pointer, guaranteed to live in SGPRs (the "s" operands of the loads above)
template <typename T> __device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    // v_readfirstlane writes an SGPR from the VALU; a VMEM instruction reading it needs 5 wait states, which the hazard
    // recognizer cannot provide for operands of inline asm
    asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return (const T*)(((unsigned long long)hi << 32) | lo);
}

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

// x = hi + lo' / 2^11 with hi = f16(x), lo' = f16((x - hi) * 2^11): 22 significant bits plus the sign of lo'
// (x - hi is exact in fp32; lo' rounds at 2^-22 |x|); two values packed per dword, low half = first
// gfx950: v_cvt_pk_f16_f32 rounds and packs two values in one instruction; x - hi is taken as fma(hi, -1, x) so that it becomes one
// v_fma_mix_f32 reading the f16 half directly (exact either way): 6 VALU instructions per pair instead of 12.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h(float xa, float xb, unsigned& p0, unsigned& p1) {
#ifdef C6_NO_SPLIT
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){xa, xb}, f16x2)); p1 = p0 ^ 0x03ff03ffu; return;   // (a lo piece that toggles like a real one: zeros would let the matrix cores run cooler and clock higher)      // (timing only: one instruction instead of the split; finite values -- garbage trips the range guard and the run repeats with bf16 pieces)
#endif
    const f16x2 hi = __builtin_convertvector((f32x2){xa, xb}, f16x2);
    p0 = __builtin_bit_cast(unsigned, hi);
    // 2^11 (x - hi) = fma(hi, -2^11, 2^11 x): every step exact (x - hi is representable, the factor a power of two), one v_fma_mix_f32 per value
    // reading the f16 half in place (hipcc, left alone, converts, subtracts and multiplies: twice the instructions)
    const f32x2 xs = (f32x2){xa, xb} * (f32x2){LO_SCALE, LO_SCALE};
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "s"(-LO_SCALE), "v"(xs[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "s"(-LO_SCALE), "v"(xs[1]));
    const f16x2 lo = __builtin_convertvector((f32x2){ra, rb}, f16x2);
    p1 = __builtin_bit_cast(unsigned, lo);
}
// PIECES tensors (round 4).  The intermediate tensor of a ResBlock (t = relu(conv1(a)), basicblock.py:211-223) has exactly one reader, the
// block's second convolution, and that reader wants it as f16 pieces: so the producer's epilogue stores the pieces -- the 32 bytes of a
// BLOCKED item hold [8 channels x hi][8 channels x lo'] instead of 8 floats, same addresses, same allocation, same zero halo -- and the
// consumer's loader copies two 16-byte half-items into the two split planes of its LDS buffer without touching them (no conversion: 12 vector
// instructions and one LDS store less per loader thread and step, on the waves whose issue slots bound the loop).  The pieces are what
// split_pair_h() makes of the stored value either way, so results are bit-identical to the fp32-tensor form.  Tensors that are also residual
// or skip operands stay fp32 (hi + lo' carries 22 bits, a residual needs all 24).
// hi (lo = 0) or scaled-low (lo = 1) pieces of the 8 channels c0[0..3], c1[0..3] of one pixel, packed as the loader's split planes want them
__device__ __forceinline__ f32x4 pieces8(const f32x4& c0, const f32x4& c1, int lo) {
    unsigned h[4], l[4];
    split_pair_h(c0[0], c0[1], h[0], l[0]); split_pair_h(c0[2], c0[3], h[1], l[1]);
    split_pair_h(c1[0], c1[1], h[2], l[2]); split_pair_h(c1[2], c1[3], h[3], l[3]);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = lo ? l[i] : h[i];
    return __builtin_bit_cast(f32x4, r);
}
// the value of the neighbouring lane (lane ^ 1) on the VALU: lane pairs hold the two 4-channel halves of one pixel's 8-channel block
__device__ __forceinline__ float lane_xor1(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    asm volatile("" : "+v"(r));      // (one move per value: hipcc otherwise merged the four moves of a vector into one -- and read element 0 four times)
    return __int_as_float(r);
}
__device__ __forceinline__ f32x4 pair_swap(const f32x4& x) {
    const float a = x[0], b = x[1], c = x[2], d = x[3];
    f32x4 y;
    y[0] = lane_xor1(a); y[1] = lane_xor1(b); y[2] = lane_xor1(c); y[3] = lane_xor1(d);
    return y;
}
constexpr float F16_RANGE = 60000.f;  // |activation| above this cannot be split (f16 max 65504): reported through range_flag
__device__ __forceinline__ void act_report(const ActMax& am, float tmax, int waves_per_block) {
    if (am.layer < 0) return;
    tmax = __builtin_bit_cast(float, wave_max_bits(tmax));
    const int wave = threadIdx.x >> 6;
    const long slot = (long)blockIdx.x * waves_per_block + wave;
    if ((threadIdx.x & 63) == 0 && slot < ACT_MAXSLOT) am.slots[(size_t)am.layer * ACT_MAXSLOT + slot] = tmax;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long n = (long)gridDim.x * waves_per_block;
        am.count[am.layer] = (int)(n < ACT_MAXSLOT ? n : ACT_MAXSLOT);
    }
}

// (Measured and removed: streaming the residual operand into an LDS tile during the last 8 steps of the loop, so that the epilogue
//  finds it on chip.  The loop is bound by the loader waves (tools/conv6p_stamps.py), so what the epilogue saved the loop lost:
//  634.9 vs 634.9 ADMM it/s, residual layers 21.3 us either way against 17.8 us for layers without a residual operand.)
template <int CFG, int SP, bool STAMP, bool INB, bool INP = false>    // INP: the input is a PIECES tensor (pieces8)
__device__ __forceinline__ void conv6_body(const Conv6Args& A) {
    static_assert(!INP || (INB && SP == 2), "pieces tensors are blocked tensors of the f16 scheme");
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
    // which 32-row half of the 64-row tile (MH = 2); A.lowhalf: only the first half exists (a layer with <= 32 output channels, the 64 -> 10
    // tail: half the matrix work and half the weight bytes of the full tile)
    const int mh = (MH > 1 && !A.lowhalf) ? bid % MH : 0;
    if (MH > 1 && !A.lowhalf) bid /= MH;
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
                    // (PIECES: half 0 = the 8 hi pieces = the whole 16-byte entry of split plane 0, half 1 = the lo' pieces = split plane 1)
                    ldsB[part][q] = INP ? (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + (lt & 1) * (2 * NPX * 16))
                                        : (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + 8 * (lt & 1));
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
        if constexpr (INP) {                     /* pieces as stored: a copy */                                  \
            _Pragma("unroll") for (int q = 0; q < NBH; ++q)                                                      \
                *(f32x4*)((unsigned char*)(Bbuf + ((c_) & 1) * (SP * 2 * NPX)) + ldsB[part_][q]) = rb_.q[q];     \
        } else if constexpr (INB) {                                                                              \
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
                f32x4 xs = x;
                if constexpr (SP == 2) {
                    if (A.out_pcs) {                                // (uniform) PIECES output: this lane's half-item = the hi (half 0) or lo' (half 1) pieces of all 8 channels
                        const f32x4 y = pair_swap(x);               // the other half of the pixel's block
                        xs = half ? pieces8(y, x, 1) : pieces8(x, y, 0);
                    }
                }
                if (off[k] != ~0u) store4(A.out + (size_t)b * A.out_bs + off[k], xs, A.wt);
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
template <int CFG> __global__ __launch_bounds__(NT6) void k_conv6i(const Conv6Args A) { conv6_body<CFG, 2, false, true, true>(A); }     // ... reading a PIECES tensor

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

template <int CFG, int NRES, bool STAMP, bool INP = false>     // INP: the input is a PIECES tensor (pieces8)
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
                ldsB[part][q] = INP ? (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + (lt & 1) * (2 * NPX * 16))     // (PIECES: as in k_conv6)
                                    : (unsigned)((h2 * NPX + dw * IHP + dh) * 16 + 8 * (lt & 1));
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
        if constexpr (INP) {                     /* pieces as stored: a copy */                                  \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) *(f32x4*)(ldsBb + ((c_) & 1) * (SP * 2 * NPX * 16) + ldsB[part_][q]) = rb_.q[q]; \
        } else {                                                                                                 \
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
                f32x4 xs_ = x;                                                                                   \
                if (A.out_pcs) {                 /* (uniform) PIECES output, as in k_conv6 */                     \
                    const f32x4 y_ = pair_swap(x);                                                               \
                    xs_ = ehalf ? pieces8(y_, x, 1) : pieces8(x, y_, 0);                                         \
                }                                                                                                \
                if (okhw_) bstore4(xs_, evoff[q], srdO, so_);                                                    \
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
            f32x4 xs = x;
            if (A.out_pcs) {                                        // (uniform) PIECES output, as in k_conv6
                const f32x4 y = pair_swap(x);
                xs = half ? pieces8(y, x, 1) : pieces8(x, y, 0);
            }
            if (off[k] != ~0u) store4(A.out + (size_t)b * A.out_bs + off[k], xs, A.wt);
            const float gm = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
            if (off[k] != ~0u) { bad |= !(gm <= F16_RANGE); tmaxp = fmaxf(tmaxp, gm); }
        }
        if (bad && A.range_flag) atomicOr(A.range_flag, 1u);
        act_report(A.am, tmaxp, NT6 / 64);
    }
}

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

// (STAMP: diagnostic instantiation, QMRI_RES_STAMPS=1 -- tools/conv6r_stamps.py: 100 MHz phase stamps of four workgroups, [wg][matrix wave 0 / loader wave 0][layer][8])
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

template <int CFG> constexpr size_t conv6_lds(int SP) {
    return (size_t)(NABUF * ast6(SP) + 2 * SP * 2 * ((((Cfg6<CFG>::TH + 2 + 7) / 16) * 16 + 8) * (Cfg6<CFG>::TW + 1) + Cfg6<CFG>::TH + 2)) * 16;
}

template <int CFG> constexpr size_t conv6p_lds() {
    return conv6_lds<CFG>(2) + (size_t)(Cfg6<CFG>::TH * Cfg6<CFG>::TW) * OTP * 4;
}

// where a launch of layer L reports (slot row = the layer's index in the network, so that the rows mean the same for every batch
// size and tile choice -- with split-K the reduce kernel reports for the layer)
static ActMax conv6_act_slot(qmri_ctx* ctx, bool reports, const ConvLayer& L) {
    NetPlan& net = ctx->net;
    ActMax am{net.d_act_slots, net.d_act_count, -1};
    if (reports && net.act_on && net.d_act_slots && L.index >= 0 && L.index < net.act_cap) am.layer = L.index;
    return am;
}

static std::atomic<int> g_launch_counter{0};     // diagnostic: running number of k_conv6 launches (all configurations, all contexts)

template <int CFG, int SP>
int launch6(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
            const PTensor* add2, int relu_out, int ksplit = 1, float* partial = nullptr, long out_ks = 0) {
    typedef Cfg6<CFG> C;
    Conv6Args A;
    A.in = in.fbase(); A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.fbase();
    A.add1 = add1 ? add1->fbase() : nullptr; A.add2 = add2 ? add2->fbase() : nullptr;
    A.in_blk = in.blk ? 1 : 0; A.out_blk = (out.blk && !partial) ? 1 : 0;       // (split-K partial sums are planar scratch)
    A.in_pcs = in.pcs ? 1 : 0; A.out_pcs = (out.pcs && !partial) ? 1 : 0;       // (... and the reduce kernel writes the pieces)
    if ((in.pcs && (!in.blk || SP != 2)) || (out.pcs && (!out.blk || SP != 2 || add1 || add2)) || (add1 && add1->pcs) || (add2 && add2->pcs)) {
        qmri_set_error(ctx, "conv layer %d: a PIECES tensor must be a blocked tensor of the f16 scheme, never a residual operand or a layer output with one", L.index);
        return QMRI_ERR_STATE;
    }
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
    static const int wt_stores = getenv("QMRI_CONV_WT") ? atoi(getenv("QMRI_CONV_WT")) : 1;
    A.wt = wt_stores;
    static const int xcd_order = getenv("QMRI_CONV_XCD") ? atoi(getenv("QMRI_CONV_XCD")) : 1;
    A.xcd = xcd_order;
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    A.stamps = (unsigned long long*)ctx->net.d_stamps;
    static const int stamp_launch = getenv("QMRI_CONV_STAMP_LAUNCH") ? atoi(getenv("QMRI_CONV_STAMP_LAUNCH")) : -1;
    A.launch_idx = g_launch_counter.fetch_add(1, std::memory_order_relaxed);
    A.detail = (stamp_launch < 0 || A.launch_idx == stamp_launch) ? 1 : 0;
    const size_t lds = conv6_lds<CFG>(SP);
    if (!ctx->conv6_attr[CFG][SP - 2]) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6<CFG, SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->conv6_attr[CFG][SP - 2] = true;
    }
    // (a layer with at most 32 output channels on the 32-row configuration: the second half of its one 64-row tile is all padding)
    A.lowhalf = (C::MH > 1 && L.Cout <= 32 && !partial) ? 1 : 0;
    const int grid = (A.lowhalf ? 1 : C::MH) * A.n_ct * A.tiles_h * A.tiles_w * ksplit * B;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (L.Cin >= 64 && L.Cout >= 64) QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1));                  // (profile level 2 only)
    if constexpr (SP == 2 && CFG < 2) {
        if (A.stamps) {                                             // QMRI_CONV_STAMPS: the diagnostic instantiation (tools/conv6_stamps.py)
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
    if constexpr (SP == 2) {
        if (in.pcs) {
            if (!ctx->conv6i_attr[CFG]) {
                QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6i<CFG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ctx->conv6i_attr[CFG] = true;
            }
            if (e0) hipExtLaunchKernelGGL((k_conv6i<CFG>), dim3(grid), dim3(NT6), (std::uint32_t)lds, ctx->stream, e0, e1, 0, A);
            else k_conv6i<CFG><<<dim3(grid), dim3(NT6), lds, ctx->stream>>>(A);
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

// persistent form (k_conv6p): one workgroup per CU walks the launch's tiles; returns QMRI_OK and sets *done when it ran
template <int CFG, int NRES>
int launch6p_t(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1, const PTensor* add2,
               int relu_out) {
    typedef Cfg6<CFG> C;
    Conv6Args A{};
    A.in = in.fbase(); A.wp = reinterpret_cast<const uint4*>(L.wp6); A.out = out.fbase();   // (BLOCKED tensors: launch6p checks)
    A.add1 = add1 ? add1->fbase() : nullptr; A.add2 = add2 ? add2->fbase() : nullptr;
    A.in_blk = 1; A.out_blk = 1;
    A.in_pcs = in.pcs ? 1 : 0; A.out_pcs = out.pcs ? 1 : 0;
    if ((out.pcs && NRES > 0) || (add1 && add1->pcs) || (add2 && add2->pcs)) {
        qmri_set_error(ctx, "conv layer %d: a PIECES tensor is never a residual operand or a layer output with one", L.index);
        return QMRI_ERR_STATE;
    }
    A.Cout = L.Cout; A.W = in.W; A.H = in.H;
    A.in_hp = in.hp; A.in_plane = (int)in.plane(); A.in_bs = (long)in.Cal * in.plane();
    A.out_hp = out.hp; A.out_plane = (int)out.plane(); A.out_bs = (long)out.Cal * out.plane();
    A.add1_bs = add1 ? (long)add1->Cal * add1->plane() : 0;
    A.add2_bs = add2 ? (long)add2->Cal * add2->plane() : 0;
    A.nchunk = L.nchunk6; A.nchunk_all = L.nchunk6; A.ksplit = 1; A.out_ks = 0; A.n_ct = L.n_ct6;
    A.tiles_h = (in.H + C::TH - 1) / C::TH; A.tiles_w = (in.W + C::TW - 1) / C::TW;
    A.ntiles = A.n_ct * A.tiles_h * A.tiles_w * B;
    A.relu_out = relu_out; A.vec4 = 1; A.wt = 1;
    static const int xcd_order = getenv("QMRI_CONV_XCD") ? atoi(getenv("QMRI_CONV_XCD")) : 1;
    A.xcd = xcd_order;
    A.range_flag = ctx->net.d_range_flag;
    A.am = conv6_act_slot(ctx, true, L);
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    static const int stamp_launch = getenv("QMRI_CONV_STAMP_LAUNCH") ? atoi(getenv("QMRI_CONV_STAMP_LAUNCH")) : -1;
    A.stamps = (unsigned long long*)ctx->net.d_stamps; A.launch_idx = g_launch_counter.fetch_add(1, std::memory_order_relaxed);
    A.detail = (A.stamps && A.launch_idx == stamp_launch) ? 1 : 0;
    if (!ctx->conv6p_attr[CFG][NRES]) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6p<CFG, NRES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6p_lds<CFG>()));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6p<CFG, NRES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6p_lds<CFG>()));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6p<CFG, NRES, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6p_lds<CFG>()));
        ctx->conv6p_attr[CFG][NRES] = true;
    }
    const int grid = std::min(A.ntiles, ctx->conv_ncu);
    if (in.pcs) {                                                   // (no diagnostic build of this form: the stamps run on fp32 tensors)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (L.Cin >= 64 && L.Cout >= 64) QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1));
        if (e0) hipExtLaunchKernelGGL((k_conv6p<CFG, NRES, false, true>), dim3(grid), dim3(NT6), (std::uint32_t)conv6p_lds<CFG>(), ctx->stream, e0, e1, 0, A);
        else k_conv6p<CFG, NRES, false, true><<<dim3(grid), dim3(NT6), conv6p_lds<CFG>(), ctx->stream>>>(A);
        QMRI_HIP(ctx, hipGetLastError());
        return QMRI_OK;
    }
    if (A.detail) {                                                 // diagnostic build of the same kernel (tools/conv6p_stamps.py)
        k_conv6p<CFG, NRES, true><<<dim3(grid), dim3(NT6), conv6p_lds<CFG>(), ctx->stream>>>(A);
        QMRI_HIP(ctx, hipGetLastError());
        return QMRI_OK;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (L.Cin >= 64 && L.Cout >= 64) QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1));
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
    static const int persist = getenv("QMRI_CONV_PERSIST") ? atoi(getenv("QMRI_CONV_PERSIST")) : 1;
    if (!persist || L.sp6 != 2 || L.Cout % 64 != 0 || L.nchunk6 < 4 || L.nchunk6 % 2 != 0 || (add2 && !add1)) return QMRI_OK;
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

template <int CFG>
int launch6(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
            const PTensor* add2, int relu_out, int ksplit = 1, float* partial = nullptr, long out_ks = 0) {
    if constexpr (CFG < 2) {
        if (ksplit == 1 && !partial) {
            bool done = false;
            QMRI_TRY(launch6p<CFG>(ctx, L, B, in, out, add1, add2, relu_out, &done));
            if (done) return QMRI_OK;
        }
    }
    return (L.sp6 == 2) ? launch6<CFG, 2>(ctx, L, B, in, out, add1, add2, relu_out, ksplit, partial, out_ks)
                        : launch6<CFG, 3>(ctx, L, B, in, out, add1, add2, relu_out, ksplit, partial, out_ks);
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
                                                            long total_half_items, unsigned* range_flag, ActMax am, int out_pcs) {
    const long i2 = (long)blockIdx.x * 256 + threadIdx.x;
    float gm = 0.f;
    // (out_pcs: the lane pairs exchange their halves -- every lane of a pair must be here, so an odd tail is handled by its even lane's bound)
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
        f32x4 xs = x;
        if (out_pcs) {                                              // PIECES output (pieces8): total_half_items is even, both lanes of a pair are active
            const f32x4 y = pair_swap(x);
            xs = half ? pieces8(y, x, 1) : pieces8(x, y, 0);
        }
        store4(out + b * out_bs + bo, xs, 1);
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

inline uint16_t host_bf16(float x) {                               // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u; std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float host_bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

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
    static const bool verbose = getenv("QMRI_ACT_VERBOSE") != nullptr;
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
    static const bool on = !(getenv("QMRI_CONV_F32") && atoi(getenv("QMRI_CONV_F32")) > 0);
    return on;
}

// operand splitting of the matrix-core path: 2 = f16 x 3 products (default), 3 = bf16 x 6 products (QMRI_CONV_SCHEME=bf16x6)
int conv6_default_sp() {
    static const int sp = (getenv("QMRI_CONV_SCHEME") && !strcmp(getenv("QMRI_CONV_SCHEME"), "bf16x6")) ? 3 : 2;
    return sp;
}

// The f16 scheme carries |w| up to the f16 maximum; larger weights (or non-finite ones) need the bf16 scheme.
bool conv6_weights_fit_f16(const float* w, size_t n) {
    for (size_t i = 0; i < n; ++i) if (!(std::fabs(w[i]) <= F16_RANGE)) return false;
    return true;
}

namespace {
// pieces of one weight in the layer's scheme (L.sp6): bf16 triple, or f16 (hi, (w - hi) * 2^11)
inline void host_split(int sp, float v, uint16_t (&h)[3], float scale = 1.f) {
    v *= scale;                                                    // (f16 scheme: a power of two, exact)
    if (sp == 3) {
        h[0] = host_bf16(v);
        const float r1 = v - host_bf16_to_f(h[0]);
        h[1] = host_bf16(r1);
        const float r2 = r1 - host_bf16_to_f(h[1]);
        h[2] = host_bf16(r2);
    } else {
        const _Float16 hi = (_Float16)v;                           // round to nearest even, as v_cvt_f16_f32
        const _Float16 lo = (_Float16)((v - (float)hi) * LO_SCALE);
        std::memcpy(&h[0], &hi, 2); std::memcpy(&h[1], &lo, 2); h[2] = 0;
    }
}
}  // namespace

// f16 scheme: the layer's weights are packed times 2^k with the largest |w| in [1, 2), and the epilogue multiplies by 2^-k -- both
// exact.  An f16 piece below 6.1e-5 is subnormal and carries an absolute, not a relative error; scaling keeps a layer of
// uniformly small weights (say 1e-5) as accurate as any other.  Returns the factor and records its inverse in the layer.
static float conv6_weight_scale(ConvLayer& L, const float* w, size_t n) {
    L.w6_descale = 1.f;
    if (L.sp6 != 2) return 1.f;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int e = 0;
    (void)std::frexp(mx, &e);                                      // mx = f * 2^e, f in [0.5, 1)
    const int k = std::min(60, std::max(-60, 1 - e));              // mx * 2^k in [1, 2)
    L.w6_descale = std::ldexp(1.f, -k);
    return std::ldexp(1.f, k);
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
    static const int wt_stores = getenv("QMRI_CONV_WT") ? atoi(getenv("QMRI_CONV_WT")) : 1;
    A.wt = wt_stores;
    static const int xcd_order = getenv("QMRI_CONV_XCD") ? atoi(getenv("QMRI_CONV_XCD")) : 1;
    A.xcd = xcd_order;
    A.descale_hi = L.w6_descale; A.descale_lo = L.w6_descale * (1.f / LO_SCALE);
    A.tiles_h = (A.GH + STH - 1) / STH; A.tiles_w = (A.GW + STW - 1) / STW;
    const int grid = A.n_ct * A.tiles_h * A.tiles_w * B;
#define LAUNCH6S(KIND_, SP_)                                                                                     \
    {                                                                                                            \
        if (in.blk) k_conv6s<KIND_, SP_, true><<<dim3(grid), dim3(NT6), conv6s_lds(SP_), ctx->stream>>>(A);      \
        else k_conv6s<KIND_, SP_, false><<<dim3(grid), dim3(NT6), conv6s_lds(SP_), ctx->stream>>>(A);            \
    }
    if (L.sp6 == 2) { if (up) LAUNCH6S(1, 2) else LAUNCH6S(0, 2) }
    else { if (up) LAUNCH6S(1, 3) else LAUNCH6S(0, 3) }
#undef LAUNCH6S
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int conv6_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                 const PTensor* add2, int relu_out) {
    // the largest pixel tile that still gives most CUs a workgroup (one workgroup per CU is the design point)
    auto ntiles = [&](int th, int tw) { return (long)L.n_ct6 * ((in.H + th - 1) / th) * ((in.W + tw - 1) / tw) * B; };
    // (tiles may overhang the image -- the kernel masks its stores -- as long as the padded area stays below 1.35 x the image)
    auto waste = [&](int th, int tw) { return (double)(((in.H + th - 1) / th) * th) * (((in.W + tw - 1) / tw) * tw) / ((double)in.H * in.W); };
    // A layer with <= 32 output channels (the 64 -> 10 tail) can run on the 32-row configuration, first half only (QMRI_CONV_TAIL32=1): half the
    // matrix work and half the weight bytes of the full 64-row tile.  Measured (profiles/r04_p_*): no faster -- 812 / 806 against 814 / 818 ADMM
    // it/s, 14.0 against 14.0 - 14.2 slices/s: the tail's launch is its loader and its boundary, not its matrix work.  OFF by default.
    static const bool tail32 = getenv("QMRI_CONV_TAIL32") && atoi(getenv("QMRI_CONV_TAIL32")) != 0;
    if (tail32 && L.sp6 == 2 && L.Cout <= 32 && L.n_ct6 == 1 && L.nchunk6 >= 4 && ntiles(16, 8) >= 160 && waste(16, 8) <= 1.35)
        return launch6<3>(ctx, L, B, in, out, add1, add2, relu_out);
    if (ntiles(16, 16) >= 160 && waste(16, 16) <= 1.35) return launch6<0>(ctx, L, B, in, out, add1, add2, relu_out);
    if (ntiles(16, 8) >= 160 && waste(16, 8) <= 1.35) return launch6<1>(ctx, L, B, in, out, add1, add2, relu_out);
    // Small feature maps with many channels (the 28 x 28 x 512 level): a 64-pixel tile would re-read the layer's weights
    // once per tile (16 x 14 MB); instead keep the 256-pixel tile and split K over workgroups, then add the partial
    // outputs in slice order (deterministic) in a second, elementwise kernel.
    static const bool splitk_on = !(getenv("QMRI_CONV_SPLITK") && atoi(getenv("QMRI_CONV_SPLITK")) == 0);
    // 56 x 56 level: tile config of the split-K variant (0, 1), 2 = no split.  With blocked tensors the unsplit 64-pixel tiles (196
    // workgroups x 48 steps, no reduce launch) win: 696 vs 672 ADMM it/s on one box (round 2; with planar tensors split-K = 2 on
    // 128-pixel tiles + a reduce kernel was the faster form)
    // (3 = 128 pixels x 32 output channels, two workgroups per 64-row weight tile: half the weight bytes through LDS per MFMA of the
    //  64-pixel tile -- these levels' steps are bound by LDS traffic, 4 fragment reads per 3 MFMAs and a full weight step written per
    //  9 MFMAs of a wave: 743 -> 756 ADMM it/s with 3 / 3 / K over 4.  Measured and not kept on the way: one barrier per chunk instead
    //  of per step (six A buffers, whole chunks requested two ahead): 18.3 vs 18.2 us per launch, the barriers are not the bound.)
    static const int mid_cfg = getenv("QMRI_CONV_MIDCFG") ? atoi(getenv("QMRI_CONV_MIDCFG")) : 3;
    static const int deep_cfg_g = getenv("QMRI_CONV_DEEPCFG") ? atoi(getenv("QMRI_CONV_DEEPCFG")) : 3;   // (28 x 28 level, see below)
    if (splitk_on && L.nchunk6 >= 16) {
        // candidate: the 256-pixel tile (28 x 28 level) or the 128-pixel tile (56 x 56 level), K split so that about one
        // workgroup per CU results and every workgroup still walks >= 4 chunks
        // 28 x 28 level: tile config and largest K split.  Measured with blocked tensors on one box (ADMM it/s, two runs each):
        // 256-pixel tiles x 8 slices (the round-1 choice) 701.6 | x 4: 695 | 128-pixel x 4: 715 | x 2: 695 | 64-pixel x 2: 717.6 | x 1: 679;
        // then, on another box: 64-pixel x 2 (and 64-pixel unsplit at 56 x 56) 743 | configuration 3 at both levels, K over 2: 751.6 | over 4: 756
        const int deep_cfg = deep_cfg_g;
        static const int deep_ks = getenv("QMRI_CONV_DEEPKS") ? atoi(getenv("QMRI_CONV_DEEPKS")) : 4;
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
            if (cfg == 0) QMRI_TRY(launch6<0>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks));
            else if (cfg == 1) QMRI_TRY(launch6<1>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks));
            else if (cfg == 2) QMRI_TRY(launch6<2>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks));
            else QMRI_TRY(launch6<3>(ctx, L, B, in, out, nullptr, nullptr, 0, ksplit, net.d_c6part, out_ks));
            const long total = (long)B * L.Cout * in.H * in.W;
            const bool vec = in.H % 4 == 0 && out.h0 % 4 == 0 && out.hp % 4 == 0 && (!add1 || (add1->h0 == out.h0 && add1->hp == out.hp)) &&
                             (!add2 || (add2->h0 == out.h0 && add2->hp == out.hp));
            if ((add1 && add1->blk != out.blk) || (add2 && add2->blk != out.blk) || (out.blk && L.Cout % 8 != 0)) {
                qmri_set_error(ctx, "conv layer %d: residual operands and output must share one tensor format", L.index);
                return QMRI_ERR_STATE;
            }
#define REDUCE_ARGS net.d_c6part + (out.h0 - 1), ksplit, out_ks, out.fbase(), add1 ? add1->fbase() : nullptr, add2 ? add2->fbase() : nullptr,        \
                add1 ? (long)add1->Cal * add1->plane() : 0, add2 ? (long)add2->Cal * add2->plane() : 0, (long)out.Cal * out.plane(),                \
                L.Cout, in.H, in.W, out.hp, (int)out.plane(), relu_out, total, (L.sp6 == 2) ? ctx->net.d_range_flag : nullptr,                      \
                conv6_act_slot(ctx, L.sp6 == 2, L)
            if (out.blk) {
                const long total_items = total / 4;             // half-items
                k_conv6_reduce_blk<<<dim3((unsigned)((total_items + 255) / 256)), dim3(256), 0, ctx->stream>>>(
                    net.d_c6part + (out.h0 - 1), ksplit, out_ks, out.fbase(), add1 ? add1->fbase() : nullptr, add2 ? add2->fbase() : nullptr,
                    add1 ? (long)add1->Cal * add1->plane() : 0, add2 ? (long)add2->Cal * add2->plane() : 0, (long)out.Cal * out.plane(), L.Cout, in.H,
                    in.W, out.hp, (int)out.plane(), relu_out, total_items, (L.sp6 == 2) ? ctx->net.d_range_flag : nullptr,
                    conv6_act_slot(ctx, L.sp6 == 2, L), out.pcs ? 1 : 0);
            } else if (vec) k_conv6_reduce<true><<<dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, ctx->stream>>>(REDUCE_ARGS);
            else k_conv6_reduce<false><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream>>>(REDUCE_ARGS);
#undef REDUCE_ARGS
            QMRI_HIP(ctx, hipGetLastError());
            return QMRI_OK;
        }
    }
    if (L.nchunk6 >= 16 && L.Cout % 64 == 0 && ((in.H <= 32) ? deep_cfg_g : mid_cfg) == 3) return launch6<3>(ctx, L, B, in, out, add1, add2, relu_out);
    return launch6<2>(ctx, L, B, in, out, add1, add2, relu_out);
}

// A run of the full-resolution level's 3x3 layers as ONE launch with resident tiles (k_conv6r, Conv6rRun in qmri_internal.h).  *done = false: not
// eligible, nothing launched.
int conv6r_try(qmri_ctx* ctx, const Conv6rRun& run, int B, bool* done) {
    *done = false;
    NetPlan& net = ctx->net;
    static const int resident = getenv("QMRI_CONV_RESIDENT") ? atoi(getenv("QMRI_CONV_RESIDENT")) : 1;
    static const bool graph_replay = getenv("QMRI_GRAPH") && atoi(getenv("QMRI_GRAPH")) > 0;     // (a captured launch would replay stale tags)
    const int nres = run.nres, nl = nres + (run.head ? 1 : 0) + (run.tail ? 1 : 0) + (run.down ? 1 : 0);
    if (!resident || graph_replay || net.res_off || !net.d_res_xbuf || B != 1 || nres < 2 || nl > R_MAXL || (nres & 1) || net.d_stamps) return QMRI_OK;
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
    A.nlayers = nl; A.hp = src.hp; A.plane = (int)src.plane(); A.tiles_h = tiles_h; A.tiles_w = tiles_w;
    A.xbuf = net.d_res_xbuf; A.xbuf_half = (size_t)tiles * R_NTRI * 64;
    static const int xcd_order = getenv("QMRI_CONV_XCD") ? atoi(getenv("QMRI_CONV_XCD")) : 1;
    A.xcd = xcd_order;
    static const int delay = getenv("QMRI_RES_DELAY") ? std::max(0, std::min(4096, atoi(getenv("QMRI_RES_DELAY")))) : 24;
    A.epoch = net.res_epoch; A.drop = net.res_drop; A.delay = delay;
    A.range_flag = net.d_range_flag; A.am_slots = net.d_act_slots; A.am_count = net.d_act_count;
    if (!ctx->conv6r_attr) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6r<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6r_lds()));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_conv6r<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)conv6r_lds()));
        ctx->conv6r_attr = true;
    }
    g_launch_counter.fetch_add(nl, std::memory_order_relaxed);
    static const int stamp_which = getenv("QMRI_RES_STAMPS") ? atoi(getenv("QMRI_RES_STAMPS")) : 0;      // 1: the up path's launch, 2: the down path's (the one with the head)
    A.stamps = (stamp_which == 2) == (run.head != nullptr) ? (unsigned long long*)net.d_res_stamps : nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // (profile level 2: one pair for the launch, counted as its 64 -> 64 layers; the head and the tail -- not timed when they are launched alone -- take the
    //  share of the duration that their matrix work has: a quarter / half of a layer's)
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, nres, (float)nres / ((float)nres + (run.head ? 0.25f : 0.f) + (run.tail ? 0.5f : 0.f) + (run.down ? 0.25f : 0.f))));
    if (A.stamps) k_conv6r<true><<<dim3(tiles), dim3(NT6), conv6r_lds(), ctx->stream>>>(A);
    else if (e0) hipExtLaunchKernelGGL((k_conv6r<false>), dim3(tiles), dim3(NT6), (std::uint32_t)conv6r_lds(), ctx->stream, e0, e1, 0, A);
    else k_conv6r<false><<<dim3(tiles), dim3(NT6), conv6r_lds(), ctx->stream>>>(A);
    QMRI_HIP(ctx, hipGetLastError());
    net.res_epoch += (unsigned)(nl - 1);                            // one tag per layer but the last; never reset, so a stale granule never carries a current tag
    *done = true;
    return QMRI_OK;
}

size_t conv6r_xbuf_bytes(int tiles) { return (size_t)2 * tiles * R_NTRI * 64; }

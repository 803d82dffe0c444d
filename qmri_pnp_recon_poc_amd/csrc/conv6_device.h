// conv6_device.h -- what the convolution kernels of the 16-bit matrix-core path share (conv6_kernels.hip: one launch per layer;
// conv6p_kernels.hip: persistent launches for slice batches; conv6r_kernels.hip: resident tiles; conv6s_kernels.hip: the 2x2 / stride-2
// layers): operand splitting, loader instructions and their counted waits, tile geometry, kernel arguments, and the host-side helpers of
// the launchers (LDS sizes, the |output| report slot, the weight splitting of the packers).  The scheme itself is described at the top of
// conv6_kernels.hip.  Everything here is internal to its translation unit (anonymous namespace).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "qmri_internal.h"
#include "conv6_act.h"
#include <hip/hip_ext.h>

extern std::atomic<int> g_conv6_launch_counter;   // diagnostic: running number of conv6 launches (all configurations, all contexts; defined in conv6_kernels.hip)

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
    int nchunk_all, ksplit;       // split-K: this launch covers nchunk of the layer's nchunk_all chunks per workgroup, ksplit workgroups per tile
    long out_ks;                  // split-K: elements between the partial outputs of consecutive K slices
    int ntiles;                   // k_conv6p: tiles of the launch (n_ct * tiles_h * tiles_w * B)
    int launch_idx, detail;       // diagnostic: running launch number; record the per-step stamps of this launch
    unsigned* range_flag;         // f16 scheme: set to 1 when an output leaves the range the next layer's f16 split can carry
    ActMax am;                    // f16 scheme: where this launch reports the largest |output| (see ACT_LOW)
    float descale_hi, descale_lo; // f16 scheme: the layer's weights are packed times 2^k (largest |w| in [1, 2)): 2^-k and 2^-k / 2^11
    unsigned long long* stamps;   // diagnostic (knob conv_stamps): [16 workgroups][2 roles][128] barrier-arrival times, 100 MHz clock
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


// fp32-equivalent algorithmic work of one layer on a batch: 2 Cout Cin taps Hout Wout B, what the profile's flop fields count
static double conv_layer_flop(const ConvLayer& L, int B, int Hin, int Win) {
    const bool k3 = (L.kind == CONV_3X3 || L.kind == CONV_3X3N);
    const double hw = k3 ? (double)Hin * Win : (L.kind == CONV_DOWN ? (double)(Hin / 2) * (Win / 2) : (double)(2 * Hin) * (2 * Win));
    return 2.0 * L.Cout * L.Cin * (k3 ? 9.0 : (L.kind == CONV_DOWN ? 4.0 : 1.0)) * hw * B;     // (transposed 2x2: every output pixel has ONE tap)
}

// split-K layers: the pair's start event rides on the convolution launch, its stop event on the reduce launch (one layer = one profile unit)
struct ProfSpan { hipEvent_t start = nullptr, stop = nullptr; bool on = false; };

// ---- host side of the weight packers (conv6_plan_pack, conv6s_plan_pack)
inline uint16_t host_bf16(float x) {                               // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u; std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float host_bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

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
// the same pieces on the device (the weight packers of qmri_set_denoiser, round 6): integer arithmetic for bf16 exactly as host_bf16; the f16
// conversions are the hardware's round-to-nearest-even, as the host compiler's (_Float16) casts; every fp32 step is a single rounded operation
__device__ __forceinline__ unsigned short dev_bf16(float x) {
    unsigned u = __float_as_uint(x);
    if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void dev_split(int sp, float v, unsigned short (&h)[3], float scale) {
    v = __fmul_rn(v, scale);
    if (sp == 3) {
        h[0] = dev_bf16(v);
        const float r1 = __fsub_rn(v, __uint_as_float((unsigned)h[0] << 16));
        h[1] = dev_bf16(r1);
        const float r2 = __fsub_rn(r1, __uint_as_float((unsigned)h[1] << 16));
        h[2] = dev_bf16(r2);
    } else {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)__fmul_rn(__fsub_rn(v, (float)hi), LO_SCALE);
        h[0] = __builtin_bit_cast(unsigned short, hi); h[1] = __builtin_bit_cast(unsigned short, lo); h[2] = 0;
    }
}
// the layer's power-of-two weight scale from its largest |w| (conv6_weight_scale's arithmetic; the maximum itself comes from the device)
static float conv6_scale_from_max(ConvLayer& L, float mx) {
    L.w6_descale = 1.f;
    if (L.sp6 != 2) return 1.f;
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    int e = 0;
    (void)std::frexp(mx, &e);
    const int k = std::min(60, std::max(-60, 1 - e));
    L.w6_descale = std::ldexp(1.f, -k);
    return std::ldexp(1.f, k);
}
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

}  // namespace

// the persistent form for slice batches (conv6p_kernels.hip): *done = false when the layer / launch does not qualify (k_conv6 runs then)
int conv6p_try(qmri_ctx* ctx, int cfg, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1, const PTensor* add2,
               int relu_out, bool* done);

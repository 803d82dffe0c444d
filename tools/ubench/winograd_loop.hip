// Micro-benchmark (round 5): what would a Winograd F(2x2, 3x3) form of the batched 3x3 convolution cost per pixel
// on this chip, with the 16 positions SERIALISED (one M accumulator pair + 2 x 2 output accumulators per tile, as the verdict asked to cost)?
//
// Numerics are settled on the CPU (tools/winograd_check.py: 1.5e-6 against fp32 direct through all 64 layers, f16 x 3 products included).
// This is the loop of such a kernel at the 224 x 224 x 64 level -- timing only, no epilogue, random operands -- built the way k_conv6p is:
//   workgroup = 64 output channels x 4 x 16 Winograd tiles (8 x 32 output pixels), 4 matrix waves + 4 loader waves, one workgroup per CU;
//   per tile: the 10 x 34 x 64 fp32 input patch goes global -> LDS (87 KB; it cannot be double-buffered beside the operand buffers), laid out
//     [8-channel block][half][column parity][row][column / 2] in 16-byte entries: the 16 lanes of a ds_read_b128 group are the 16 tiles of one tile
//     row, and whatever position is being formed they read 16 CONSECUTIVE entries -- conflict-free (the first version, 8 x 8 tiles with the pixels
//     of a block 32 bytes apart, read with 4-way conflicts: 136 cycles per pixel for the loader side alone, profiles/r05_e_*_8x8_tiles.txt);
//   per position xi = (a, b) of the 16:
//     loader waves   V_xi = (B^T d B)_xi for 64 tiles x 64 channels ON THE FLY from the resident patch (4 patch pixels -> 3 adds per value; the
//                    two-pass form needs all 16 positions of a tile at once = 256 KB of LDS), f16 split (hi, 2^11 (x - hi)) as split_pair_h,
//                    16 KB into the B buffer of the next position; U_xi (pre-transformed, pre-split weights, 16 KB) global -> LDS A buffer
//     matrix waves   M_xi = U_xi V_xi: 32 x 32 per wave, K = 64 = 4 steps x 3 products (hi hi; lo hi + hi lo), then the output transform in
//                    registers: t = M_hi + 2^-11 M_cross, Y[i][j] += A^T[i][a] A^T[j][b] t  (1 + 1 ... 4 vector instructions per value)
//     one LDS barrier per position (double-buffered A and B)
// Reported: cycles per tile and per output pixel (one workgroup per CU, 256 CUs, 12 tiles each = a 15-slice layer's 2940 tiles), the clock held,
// and the time such a layer's LOOP would take -- to set against k_conv6p's 164 us per 15-slice layer (111 cycles per pixel and CU at 2.0 GHz,
// everything included).  Variants isolate the sides: matrix waves alone, loader waves alone.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/ubench/winograd_loop.hip -o tools/ubench/winograd_loop && tools/ubench/winograd_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NLD = 256;
constexpr int PR = 10, PC = 34, PCH = PC / 2;            // patch rows, columns, columns of one parity
constexpr int PLANE = PR * PCH;                          // entries of one (block, half, column parity) plane
constexpr int PATCH_F4 = 8 * 2 * 2 * PLANE;              // float4 of the patch: [cb 8][half 2][column parity 2][row 10][column / 2 17] = 5440 (87 KB)
constexpr int ABUF = 1024, BBUF = 1024;                  // uint4 per buffer (16 KB each)
constexpr size_t LDS_BYTES = (size_t)PATCH_F4 * 16 + 2 * (ABUF + BBUF) * 16;   // 87 040 + 65 536 = 152 576

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// hi = f16(x), lo' = f16((x - hi) 2^11): conv6_device.h split_pair_h
__device__ __forceinline__ void split_pair_h(float xa, float xb, unsigned& p0, unsigned& p1) {
    const f16x2 hi = __builtin_convertvector((f32x2){xa, xb}, f16x2);
    p0 = __builtin_bit_cast(unsigned, hi);
    const f32x2 xs = (f32x2){xa, xb} * (f32x2){2048.f, 2048.f};
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p0), "s"(-2048.f), "v"(xs[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p0), "s"(-2048.f), "v"(xs[1]));
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, f16x2));
}
// B^T rows of F(2x2, 3x3): position a takes input rows (i0, i1) with signs (s0, s1):  a=0: d0 - d2; 1: d1 + d2; 2: d2 - d1; 3: d1 - d3
template <int A> struct BT;
template <> struct BT<0> { static constexpr int i0 = 0, i1 = 2; static constexpr float s0 = 1.f, s1 = -1.f; };
template <> struct BT<1> { static constexpr int i0 = 1, i1 = 2; static constexpr float s0 = 1.f, s1 = 1.f; };
template <> struct BT<2> { static constexpr int i0 = 2, i1 = 1; static constexpr float s0 = 1.f, s1 = -1.f; };
template <> struct BT<3> { static constexpr int i0 = 1, i1 = 3; static constexpr float s0 = 1.f, s1 = -1.f; };
// A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
__host__ __device__ constexpr int at(int i, int a) { return i == 0 ? (a < 3 ? 1 : 0) : (a == 0 ? 0 : (a == 1 ? 1 : -1)); }

// one unit of the loaders' work for position (A, B): V of 8 channels (block cb) of one Winograd tile -> hi and lo' entries of the B buffer
template <int A, int B>
__device__ __forceinline__ void make_v(const f32x4* patch, uint4* bb, int cb, int tile) {
    const int ty = tile >> 4, tx = tile & 15;
    const int r0 = 2 * ty + BT<A>::i0, r1 = 2 * ty + BT<A>::i1;
    // column 2 tx + j lies in parity plane j & 1 at entry tx + (j >> 1): compile-time plane and offset, lane-consecutive entries
    constexpr int j0 = BT<B>::i0, j1 = BT<B>::i1;
    f32x4 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4* p0 = patch + (size_t)((cb * 2 + h) * 2 + (j0 & 1)) * PLANE + (j0 >> 1) + tx;
        const f32x4* p1 = patch + (size_t)((cb * 2 + h) * 2 + (j1 & 1)) * PLANE + (j1 >> 1) + tx;
        const f32x4 d00 = p0[r0 * PCH], d01 = p1[r0 * PCH], d10 = p0[r1 * PCH], d11 = p1[r1 * PCH];
        // (s0 d00 + s1 d01) s0' + (s0 d10 + s1 d11) s1': three adds / subtracts per value, signs folded at compile time
        const f32x4 u0 = (BT<B>::s1 > 0) ? d00 + d01 : d00 - d01, u1 = (BT<B>::s1 > 0) ? d10 + d11 : d10 - d11;
        v[h] = (BT<A>::s1 > 0) ? u0 + u1 : u0 - u1;
    }
    uint4 hi, lo;
    split_pair_h(v[0][0], v[0][1], hi.x, lo.x); split_pair_h(v[0][2], v[0][3], hi.y, lo.y);
    split_pair_h(v[1][0], v[1][1], hi.z, lo.z); split_pair_h(v[1][2], v[1][3], hi.w, lo.w);
    const int ks = cb >> 1, kh = cb & 1;
    bb[((0 * 4 + ks) * 2 + kh) * 64 + tile] = hi;
    bb[((1 * 4 + ks) * 2 + kh) * 64 + tile] = lo;
}

template <int A, int B, int MODE>
__device__ __forceinline__ void position(const f32x4* patch, uint4* abuf, uint4* bbuf, const uint4* __restrict__ U, int xi_next_off, int wave, int lane, int lt,
                                         f32x16 (&Y)[2][2]) {
    // buffers: this position's operands are in half (A * 4 + B) & 1, the next position's go into the other half
    constexpr int XI = A * 4 + B, CUR = XI & 1, NXT = CUR ^ 1;
    constexpr int NA = (XI + 1) & 15, NAa = NA >> 2, NAb = NA & 3;  // the next position (wraps into the next tile's first)
    asm volatile("" ::: "memory");                                   // (nothing of the next position is hoisted over this one's barrier)
    if (wave >= 4) {
        if (MODE != 1) {                                            // ---- loader waves: V and U of the NEXT position
            uint4 u[4];
            int uoff = xi_next_off;
            asm volatile("" : "+s"(uoff));                           // (opaque: or hipcc keeps 16 positions x 4 64-bit addresses in registers across the tile loop)
            const uint4* Up = U + uoff;
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = Up[lt + NLD * q];
            uint4* bb = bbuf + NXT * BBUF;
            make_v<NAa, NAb>(patch, bb, lt >> 6, lt & 63);
            asm volatile("" ::: "memory");                           // (one unit's eight patch reads in flight at a time: registers)
            make_v<NAa, NAb>(patch, bb, 4 + (lt >> 6), lt & 63);
#pragma unroll
            for (int q = 0; q < 4; ++q) abuf[NXT * ABUF + lt + NLD * q] = u[q];
        }
    } else if (MODE != 2) {                                         // ---- matrix waves: M = U V on this position's buffers, then the output transform
        const int m = wave & 1, nb = wave >> 1;
        const uint4* ab = abuf + CUR * ABUF;
        const uint4* bb = bbuf + CUR * BBUF;
        f32x16 mh, mc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { mh[r] = 0.f; mc[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f16x8 ah = __builtin_bit_cast(f16x8, ab[((ks * 2 + m) * 2 + 0) * 64 + lane]), al = __builtin_bit_cast(f16x8, ab[((ks * 2 + m) * 2 + 1) * 64 + lane]);
            const f16x8 bh = __builtin_bit_cast(f16x8, bb[((0 * 4 + ks) * 2 + (lane >> 5)) * 64 + nb * 32 + (lane & 31)]);
            const f16x8 bl = __builtin_bit_cast(f16x8, bb[((1 * 4 + ks) * 2 + (lane >> 5)) * 64 + nb * 32 + (lane & 31)]);
            mh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, mh, 0, 0, 0);
            mc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, mc, 0, 0, 0);
            mc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, mc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = __builtin_fmaf(mc[r], 1.f / 2048.f, mh[r]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    constexpr int dummy = 0; (void)dummy;
                    const int c = at(i, A) * at(j, B);
                    if (c > 0) Y[i][j][r] += t; else if (c < 0) Y[i][j][r] -= t;
                }
        }
    }
    lds_barrier();
}

template <int MODE>      // 0: both sides; 1: matrix waves alone (operands of the first positions stay in LDS); 2: loader waves alone
__global__ __launch_bounds__(NT) void k_wino(const float* __restrict__ act, const uint4* __restrict__ U, float* out, unsigned long long* cyc, int ntiles, int act_tiles) {
    extern __shared__ __align__(16) unsigned char smem[];
    f32x4* patch = (f32x4*)smem;
    uint4* abuf = (uint4*)(smem + (size_t)PATCH_F4 * 16);
    uint4* bbuf = abuf + 2 * ABUF;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lt = tid & 255;    // (wave: a scalar, so the roles are scalar branches)
    for (int i = tid; i < 2 * (ABUF + BBUF); i += NT) abuf[i] = make_uint4(0x3c003c00u + i, 0x34003a00u ^ (i * 2654435761u & 0x03ff03ffu), 0x3c003800u, 0x38003c00u + 7 * i);
    f32x16 Y[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[i][j][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < ntiles; ++t) {
        // ---- the tile's input patch: 83 KB global -> LDS, all eight waves (it cannot be prefetched: no room for a second patch)
        const f32x4* src = (const f32x4*)act + (size_t)((blockIdx.x * ntiles + t) % act_tiles) * PATCH_F4;
        if (MODE != 1) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {                   // (6 + 5 requests per thread in flight: registers; 5440 float4 in all)
                f32x4 r[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) { const int i = tid + NT * (6 * half + q); r[q] = src[i < PATCH_F4 ? i : 0]; }
#pragma unroll
                for (int q = 0; q < 6; ++q) { const int i = tid + NT * (6 * half + q); if (i < PATCH_F4) patch[i] = r[q]; }
                asm volatile("" ::: "memory");
            }
        }
        lds_barrier();
        // ---- the first position's operands (the loop below makes position xi + 1 while xi is multiplied)
        if (wave >= 4 && MODE != 1) {
            uint4* bb = bbuf + 0 * BBUF;
            make_v<0, 0>(patch, bb, lt >> 6, lt & 63);
            asm volatile("" ::: "memory");
            make_v<0, 0>(patch, bb, 4 + (lt >> 6), lt & 63);
#pragma unroll
            for (int q = 0; q < 4; ++q) abuf[0 * ABUF + lt + NLD * q] = U[lt + NLD * q];
        }
        lds_barrier();
#define POS(A_, B_) position<A_, B_, MODE>(patch, abuf, bbuf, U, (((A_ * 4 + B_) + 1) & 15) * ABUF, wave, lane, lt, Y);
        POS(0, 0) POS(0, 1) POS(0, 2) POS(0, 3) POS(1, 0) POS(1, 1) POS(1, 2) POS(1, 3)
        POS(2, 0) POS(2, 1) POS(2, 2) POS(2, 3) POS(3, 0) POS(3, 1) POS(3, 2) POS(3, 3)
#undef POS
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += Y[i][j][r];
    if (MODE == 2) s += __builtin_bit_cast(float, bbuf[tid].x) + patch[tid][0];
    out[blockIdx.x * NT + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, const float* act, const uint4* U, int act_tiles) {
    const int nwg = 256, ntiles = 12;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nwg * NT * 4); hipMalloc(&cyc, nwg * 8);
    hipFuncSetAttribute((const void*)k_wino<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) k_wino<MODE><<<nwg, NT, LDS_BYTES>>>(act, U, out, cyc, ntiles, act_tiles);       // warm: the clock settles under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) k_wino<MODE><<<nwg, NT, LDS_BYTES>>>(act, U, out, cyc, ntiles, act_tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> h(nwg); hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < nwg; ++i) c += (double)h[i];
    c /= nwg;
    const double per_tile = c / ntiles, clk = c / (ms * 1e6);                                // cycles per 256-pixel tile; GHz
    printf("%-46s %8.0f cycles per tile = %6.1f per output pixel (matrix cores alone: 24.0)  clock %.2f GHz  launch %.1f us for 12 tiles per CU  "
           "-> a 15-slice 224 x 224 x 64 layer's loop (2940 tiles on 256 CUs): %.1f us\n", name, per_tile, per_tile / 256.0, clk, ms * 1e3, ms * 1e3 * (2940.0 / 256.0) / ntiles);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int act_tiles = 512;                                        // 512 x 83 KB = 42 MB of fp32 patches: more than the L2s, inside the Infinity Cache
    std::vector<float> ha((size_t)act_tiles * PATCH_F4 * 4);
    unsigned st = 12345u;
    for (float& v : ha) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) * (1.0f / 16777216.0f); }
    std::vector<unsigned> hu((size_t)16 * ABUF * 4);
    for (unsigned& v : hu) {                                          // random f16 pairs with ordinary exponents (toggling operands: the clock the chip really holds)
        st = st * 1664525u + 1013904223u; const unsigned a = ((st >> 9) & 0x3FFu) | ((13u + ((st >> 20) & 3u)) << 10) | ((st >> 31) << 15);
        st = st * 1664525u + 1013904223u; const unsigned b = ((st >> 9) & 0x3FFu) | ((13u + ((st >> 20) & 3u)) << 10) | ((st >> 31) << 15);
        v = a | (b << 16);
    }
    float* act; uint4* U;
    hipMalloc(&act, ha.size() * 4); hipMalloc(&U, hu.size() * 4);
    hipMemcpy(act, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(U, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
    printf("Winograd F(2x2, 3x3), 16 positions serialised, f16 x 3 products, workgroup = 64 cout x 64 tiles (8 x 32 pixels), 4 matrix + 4 loader waves, %zu KB of LDS, conflict-free patch reads\n", LDS_BYTES / 1024);
    run<0>("matrix + loader waves", act, U, act_tiles);
    run<1>("matrix waves alone (operands stay in LDS)", act, U, act_tiles);
    run<2>("loader waves alone (patch, V, U)", act, U, act_tiles);
    run<0>("matrix + loader waves (again)", act, U, act_tiles);
    printf("reference: k_conv6p, the direct form, takes 164 us per 15-slice layer with everything in it (epilogue, residual operands, tile switches) = 111 cycles per output\n"
           "pixel and CU at 2.0 GHz; its matrix waves alone 120 us (profiles/r03_hh_*).  The direct form's matrix-core cycles are 54 per pixel, this form's 24.\n");
    return 0;
}

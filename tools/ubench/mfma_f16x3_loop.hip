// Micro-benchmark: the chunk loop of k_conv6<0, 2>'s MFMA waves (f16 x 3 products: 2 cout tiles x 2 pixel blocks per wave,
// 9 taps, 8 fragment reads and 12 MFMAs per tap, two accumulator sets) alone in a workgroup of 4 waves -- no loaders.
// READS: 2 = A and B fragments from LDS as in the kernel, 1 = only B from LDS (A stays in registers), 0 = no LDS reads.
// Tells how much of the step time is the LDS read path of the four MFMA waves themselves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SP = 2, IHP = 24, NPX = 24 * 17 + 18, AST = 3 * 2 * SP * 64, MW = 2, NCT = 2;

__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// DATA: 0 = nearly constant operands, 1 = pseudo-random f16 values in [-2, 2) (every bit of the mantissa toggles)
__device__ __forceinline__ unsigned rnd16(unsigned& st) {
    st = st * 1664525u + 1013904223u;
    const unsigned m = (st >> 9) & 0x3FFu, e = 13u + ((st >> 20) & 3u), sg = (st >> 31) << 15;      // exponents 13..16 -> 0.25 .. 4
    return sg | (e << 10) | m;
}
template <int SYNC, int READS, int DATA>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int nchunk) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;
    uint4* Bbuf = Abuf + 3 * AST;
    for (int i = threadIdx.x; i < 3 * AST + 2 * SP * 2 * NPX; i += 256) {
        if (DATA == 0) Abuf[i] = make_uint4(0x3c003c00u + (i & 7), 0x3c013c00u, 0x3c003c02u, 0x3c033c00u);
        else {
            unsigned st = i * 2654435761u + blockIdx.x;
            uint4 v;
            v.x = rnd16(st) | (rnd16(st) << 16); v.y = rnd16(st) | (rnd16(st) << 16); v.z = rnd16(st) | (rnd16(st) << 16); v.w = rnd16(st) | (rnd16(st) << 16);
            Abuf[i] = v;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h2 = lane >> 5;
    const int pbw = 4 * wave, pbh = 0, m0 = 0;
    const int pxl = (pbw + (li >> 3)) * IHP + pbh + (li & 7);
    f32x16 acc[MW][NCT], accl[MW][NCT];
    for (int m = 0; m < MW; ++m) for (int n = 0; n < NCT; ++n) for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
    u32x4 bf[2][NCT][SP], af[2][MW][SP];
    for (int s = 0; s < 2; ++s) for (int m = 0; m < 2; ++m) for (int sp = 0; sp < SP; ++sp) {
        af[s][m][sp] = __builtin_bit_cast(u32x4, Abuf[lane + 64 * (m * 2 + sp)]);
        bf[s][m][sp] = __builtin_bit_cast(u32x4, Bbuf[lane + 64 * (m * 2 + sp)]);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < nchunk; ++c) {
        const uint4* ab = Abuf + lane;
        const uint4* bb = Bbuf + (c & 1) * (SP * 2 * NPX) + h2 * NPX + pxl;
        auto frag_a = [&](int T, int set, int m, int sp) __attribute__((always_inline)) {
            const int kh = T / 3, kw = T - 3 * kh;
            if (READS >= 2) af[set][m][sp] = __builtin_bit_cast(u32x4, ab[kh * AST + ((kw * 2 + (m0 + m)) * SP + sp) * 64]);
        };
        auto frag_b = [&](int T, int set, int n, int sp) __attribute__((always_inline)) {
            const int kh = T / 3, kw = T - 3 * kh;
            if (READS >= 1) bf[set][n][sp] = __builtin_bit_cast(u32x4, bb[sp * 2 * NPX + kw * IHP + kh + 8 * n]);
        };
        auto frags = [&](int T, int set) __attribute__((always_inline)) {
            frag_a(T, set, 0, 0); frag_b(T, set, 0, 0); frag_a(T, set, 0, 1); frag_b(T, set, 0, 1);
            frag_b(T, set, 1, 0); frag_b(T, set, 1, 1);
            frag_a(T, set, 1, 0); frag_a(T, set, 1, 1);
        };
        frags(0, 0);
#pragma unroll
        for (int T = 0; T < 9; ++T) {
            const int cur = T & 1;
            if (T < 8) frags(T + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NCT; ++n) {
                    acc[m][n] = mfma_h(af[cur][m][0], bf[cur][n][0], acc[m][n]);
                    f32x16 l_ = accl[m][n];
                    l_ = mfma_h(af[cur][m][1], bf[cur][n][0], l_);
                    l_ = mfma_h(af[cur][m][0], bf[cur][n][1], l_);
                    accl[m][n] = l_;
                }
            if (SYNC && T % 3 == 2) __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < MW; ++m) for (int n = 0; n < NCT; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r] + accl[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SYNC, int READS, int DATA> void run(const char* name) {
    const int nchunk = 400, nwg = 256;
    const size_t lds = (size_t)(3 * AST + 2 * SP * 2 * NPX) * 16;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nwg * 256 * 4); hipMalloc(&cyc, nwg * 8);
    hipFuncSetAttribute((const void*)k<SYNC, READS, DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SYNC, READS, DATA><<<nwg, 256, lds>>>(out, cyc, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SYNC, READS, DATA><<<nwg, 256, lds>>>(out, cyc, nchunk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < nwg; ++i) c += h[i];
    c /= nwg;
    const double nm = 108.0 * nchunk;
    printf("%-44s cycles/MFMA %.1f  (per 36-MFMA step %.0f)  ns/MFMA %.2f  clock %.2f GHz\n", name, c / nm, 36 * c / nm, ms * 1e6 / nm, c / (ms * 1e6));
}
int main() {
    run<0, 2, 0>("A+B fragments from LDS, no sync");
    run<1, 2, 0>("A+B fragments from LDS, barrier per step");
    run<0, 1, 0>("B fragments from LDS only, no sync");
    run<0, 0, 0>("no LDS reads, no sync");
    run<0, 2, 1>("random operands: A+B from LDS, no sync");
    run<1, 2, 1>("random operands: A+B from LDS, barrier/step");
    run<0, 0, 1>("random operands: no LDS reads, no sync");
    return 0;
}

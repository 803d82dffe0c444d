// Micro-benchmark (round 3): does the MFMA SHAPE change what the chip sustains on toggling operands?  MI355X_MICROARCH.md ("DVFS give-back", item 7)
// reports 1.12-1.15 x the FLOP/s for v_mfma_f32_16x16x32 against v_mfma_f32_32x32x16 at equal cycles per FLOP.  Here: the f16 x 3 product
// scheme of k_conv6 on a 64 x 64 wave tile, every fragment read from LDS by ds_read_b128, one wave per SIMD, 256 workgroups, pseudo-random f16 data.
//   shape 0: per K = 16: 4 A + 4 B fragment reads, 12 x v_mfma_f32_32x32x16_f16
//   shape 1: per K = 32: 8 A + 8 B fragment reads, 48 x v_mfma_f32_16x16x32_f16        (the same FLOP, LDS bytes and matrix-core cycles per K)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NENT = 4096;   // 64 KB of LDS operands

__device__ __forceinline__ unsigned rnd16(unsigned& st) {
    st = st * 1664525u + 1013904223u;
    const unsigned m = (st >> 9) & 0x3FFu, e = 13u + ((st >> 20) & 3u), sg = (st >> 31) << 15;
    return sg | (e << 10) | m;
}
template <int SHAPE, int SYNC, int DATA>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int nk32) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* buf = (uint4*)smem;
    for (int i = threadIdx.x; i < NENT; i += 256) {
        unsigned st = i * 2654435761u + blockIdx.x;
        uint4 v;
        if (DATA) { v.x = rnd16(st) | (rnd16(st) << 16); v.y = rnd16(st) | (rnd16(st) << 16); v.z = rnd16(st) | (rnd16(st) << 16); v.w = rnd16(st) | (rnd16(st) << 16); }
        else v = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        buf[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* base = buf + lane + 64 * wave;
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][2], accl[2][2];
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
        for (int it = 0; it < 2 * nk32; ++it) {                     // K = 16 per iteration
            const uint4* p = base + ((it * 8 * 64) & (NENT - 1024));
            u32x4 a[2][2], b[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) { a[m][sp] = __builtin_bit_cast(u32x4, p[(m * 2 + sp) * 64]); b[m][sp] = __builtin_bit_cast(u32x4, p[256 + (m * 2 + sp) * 64]); }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][0]), acc[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][1]), __builtin_bit_cast(f16x8, b[n][0]), accl[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][1]), accl[m][n], 0, 0, 0);
                }
            if (SYNC && (it % 3) == 2) __syncthreads();
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r] + accl[m][n][r];
    } else {
        f32x4 acc[4][4], accl[4][4];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) { acc[m][n][r] = 0.f; accl[m][n][r] = 0.f; }
        for (int it = 0; it < nk32; ++it) {                         // K = 32 per iteration
            const uint4* p = base + ((it * 16 * 64) & (NENT - 1024 - 64));
            u32x4 a[4][2], b[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) { a[m][sp] = __builtin_bit_cast(u32x4, p[(m * 2 + sp) * 64]); b[m][sp] = __builtin_bit_cast(u32x4, p[512 + (m * 2 + sp) * 64]); }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][0]), acc[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m][1]), __builtin_bit_cast(f16x8, b[n][0]), accl[m][n], 0, 0, 0);
                    accl[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m][0]), __builtin_bit_cast(f16x8, b[n][1]), accl[m][n], 0, 0, 0);
                }
            if (SYNC && (it % 3) == 2) __syncthreads();
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r] + accl[m][n][r];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int SYNC, int DATA> void run(const char* name) {
    const int nk32 = 6000, nwg = 256;
    const size_t lds = (size_t)NENT * 16;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nwg * 256 * 4); hipMalloc(&cyc, nwg * 8);
    hipFuncSetAttribute((const void*)k<SHAPE, SYNC, DATA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<SHAPE, SYNC, DATA><<<nwg, 256, lds>>>(out, cyc, nk32);       // warm: the clock settles under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) k<SHAPE, SYNC, DATA><<<nwg, 256, lds>>>(out, cyc, nk32);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h[256]; hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < nwg; ++i) c += h[i];
    c /= nwg;
    const double flop = 2.0 * 64 * 64 * 32 * 3 * (double)nk32 * 4 * nwg;       // executed MFMA FLOP of the launch
    printf("%-58s cycles per K=32 step %.0f (ideal 768)  %.1f TFLOP/s executed  clock %.2f GHz\n", name, c / nk32, flop / (ms * 1e-3) / 1e12, c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 0, 1>("32x32x16, random f16 operands, no barrier");
    run<1, 0, 1>("16x16x32, random f16 operands, no barrier");
    run<0, 1, 1>("32x32x16, random f16 operands, barrier per 3 K-steps");
    run<1, 1, 1>("16x16x32, random f16 operands, barrier per 3 K-steps");
    run<0, 0, 0>("32x32x16, constant operands, no barrier");
    run<1, 0, 0>("16x16x32, constant operands, no barrier");
    run<0, 0, 1>("32x32x16, random f16 operands, no barrier (again)");
    run<1, 0, 1>("16x16x32, random f16 operands, no barrier (again)");
    return 0;
}

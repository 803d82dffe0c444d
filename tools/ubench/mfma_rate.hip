// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4 on gfx950, alone and beside ds_read_b32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float av, float bv) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = bv + i;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x16){0};
    float a0 = av + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float b = LDS ? sh[(threadIdx.x + 64 * u + it) & 4095] : bv;
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[a], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float av, float bv) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0};
    float a0 = av + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    for (int wg : {256, 512, 1024}) {
        double fl32 = (double)wg * 4 * iters * 16 * 4096.0;
        float t;
        t = timeit([&] { k32<1, false><<<wg, 256>>>(out, iters, 1.f, 2.f); });
        printf("32x32x2 acc1 noLDS  wg=%4d: %.3f ms  %.1f TF\n", wg, t, fl32 * 1 / t / 1e9);
        t = timeit([&] { k32<2, false><<<wg, 256>>>(out, iters, 1.f, 2.f); });
        printf("32x32x2 acc2 noLDS  wg=%4d: %.3f ms  %.1f TF\n", wg, t, fl32 * 2 / t / 1e9);
        t = timeit([&] { k32<1, true><<<wg, 256>>>(out, iters, 1.f, 2.f); });
        printf("32x32x2 acc1 LDS    wg=%4d: %.3f ms  %.1f TF\n", wg, t, fl32 * 1 / t / 1e9);
        t = timeit([&] { k32<2, true><<<wg, 256>>>(out, iters, 1.f, 2.f); });
        printf("32x32x2 acc2 LDS    wg=%4d: %.3f ms  %.1f TF\n", wg, t, fl32 * 2 / t / 1e9);
        double fl16 = (double)wg * 4 * iters * 16 * 2048.0;
        t = timeit([&] { k16<4><<<wg, 256>>>(out, iters, 1.f, 2.f); });
        printf("16x16x4 acc4        wg=%4d: %.3f ms  %.1f TF\n", wg, t, fl16 * 4 / t / 1e9);
    }
    return 0;
}

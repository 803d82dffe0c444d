// Probe (round 3): what a dependent launch costs as a function of the launch shape -- grid, block size, dynamic LDS, register footprint --
// for kernels that do next to nothing.  A chain of 200 launches on one stream, HIP events around it; prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char smem[];
template <int REGS>
__global__ void k(float* p, int n) {
    float v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = p[(threadIdx.x + i) & 63];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += v[i] * (float)(i + n);
    if (s == 123.456f) p[blockIdx.x] = s + smem[threadIdx.x & 15];     // (never true: keeps the registers and the LDS allocation alive)
}
template <int REGS>
static float run(int grid, int block, size_t lds, float* d) {
    hipFuncSetAttribute((const void*)k<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) k<REGS><<<grid, block, lds>>>(d, i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) k<REGS><<<grid, block, lds>>>(d, i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 200.f;
}
int main() {
    float* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    const int grids[] = {196, 256, 1024};
    for (int g : grids) {
        printf("grid %4d  block 256 LDS 0      regs  8: %.2f us   regs 200: %.2f us\n", g, run<8>(g, 256, 0, d), run<200>(g, 256, 0, d));
        printf("grid %4d  block 512 LDS 0      regs  8: %.2f us   regs 200: %.2f us\n", g, run<8>(g, 512, 0, d), run<200>(g, 512, 0, d));
        printf("grid %4d  block 512 LDS 64 KB  regs  8: %.2f us   regs 200: %.2f us\n", g, run<8>(g, 512, 64 * 1024, d), run<200>(g, 512, 64 * 1024, d));
        printf("grid %4d  block 512 LDS 157 KB regs  8: %.2f us   regs 200: %.2f us\n", g, run<8>(g, 512, 157 * 1024, d), run<200>(g, 512, 157 * 1024, d));
    }
    return 0;
}

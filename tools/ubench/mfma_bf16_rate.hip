// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 in the pattern of k_conv6's MFMA waves.
//   variant 0: 72 MFMAs per step on 4 accumulators round-robin, operands in registers
//   variant 1: the same plus the 36 ds_read_b128 fragment reads of a step (12 per 24 MFMAs, requested one tap ahead)
// One wave per SIMD (256 threads per workgroup, one workgroup per CU).  Prints shader cycles (s_memtime) and ns per MFMA.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int steps) {
    __shared__ uint4 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 fa[2][6], fb[2][6];
    for (int s = 0; s < 2; ++s) for (int q = 0; q < 6; ++q) { fa[s][q] = __builtin_bit_cast(bf16x8, lds[lane + 64 * q]); fb[s][q] = __builtin_bit_cast(bf16x8, lds[lane + 64 * (q + 6)]); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < steps; ++g) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int cur = t & 1;
            if (VAR == 1) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    fa[cur ^ 1][q] = __builtin_bit_cast(bf16x8, lds[lane + 64 * (q + 12 * t) + (g & 1) * 2048]);
                    fb[cur ^ 1][q] = __builtin_bit_cast(bf16x8, lds[lane + 64 * (q + 6 + 12 * t) + (g & 1) * 2048]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x16 a_ = acc[m * 2 + n];
#pragma unroll
                    for (int p = 0; p < 6; ++p) a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][m * 3 + p % 3], fb[cur][n * 3 + p / 2], a_, 0, 0, 0);
                    acc[m * 2 + n] = a_;
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR> void run(const char* name) {
    const int steps = 2000, nwg = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nwg * 256 * 4); hipMalloc(&cyc, nwg * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<VAR><<<nwg, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<VAR><<<nwg, 256>>>(out, cyc, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < nwg; ++i) c += h[i];
    c /= nwg;
    const double nm = 72.0 * steps;
    printf("%-28s cycles/MFMA %.1f   ns/MFMA %.2f   => clock %.2f GHz   chip rate %.0f TFLOP/s\n", name, c / nm, ms * 1e6 / nm, c / (ms * 1e6),
           nm * 32768.0 * nwg * 4 / (ms * 1e-3) / 1e12);
}
int main() { run<0>("registers only"); run<1>("with fragment reads (LDS)"); return 0; }

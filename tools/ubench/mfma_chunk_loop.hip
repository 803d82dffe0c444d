// Micro-benchmark: the chunk loop of k_conv6's MFMA waves (CFG 0: 2 cout tiles x 2 pixel blocks per wave, 9 taps, fragments
// of tap T+1 requested before the MFMAs of tap T) alone in a workgroup of 4 waves -- no loaders, no synchronisation.
// Separates the cost of the instruction pattern from the cost of sharing the CU with the loader waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int IHP = 24, NPX = 24 * 17 + 18, AST = 3 * 2 * 3 * 64, MW = 2, NCT = 2;

template <int SYNC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int nchunk) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* Abuf = (uint4*)smem;
    uint4* Bbuf = Abuf + 3 * AST;
    for (int i = threadIdx.x; i < 3 * AST + 2 * 6 * NPX; i += 256) Abuf[i] = make_uint4(0x3f803f80u + i, 0x3f813f80u, 0x3f803f82u, 0x3f833f80u);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h2 = lane >> 5;
    const int pbw = 4 * wave, pbh = 0, m0 = 0;
    const int pxl = (pbw + (li >> 3)) * IHP + pbh + (li & 7);
    f32x16 acc[MW][NCT];
    for (int m = 0; m < MW; ++m) for (int n = 0; n < NCT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < nchunk; ++c) {
        const uint4* ab = Abuf + lane;
        const uint4* bb = Bbuf + (c & 1) * (3 * 2 * NPX) + h2 * NPX + pxl;
        bf16x8 bf[2][NCT][3], af[2][MW][3];
        auto frags = [&](int T, int set) __attribute__((always_inline)) {
            const int kh = T / 3, kw = T - 3 * kh;
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bf[set][n][sp] = __builtin_bit_cast(bf16x8, bb[sp * 2 * NPX + kw * IHP + kh + 8 * n]);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) af[set][m][sp] = __builtin_bit_cast(bf16x8, ab[kh * AST + ((kw * 2 + (m0 + m)) * 3 + sp) * 64]);
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
                    f32x16 a_ = acc[m][n];
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][2], bf[cur][n][0], a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][0], bf[cur][n][2], a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][1], bf[cur][n][1], a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][1], bf[cur][n][0], a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][0], bf[cur][n][1], a_, 0, 0, 0);
                    a_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][m][0], bf[cur][n][0], a_, 0, 0, 0);
                    acc[m][n] = a_;
                }
            if (SYNC && T % 3 == 2) __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < MW; ++m) for (int n = 0; n < NCT; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SYNC> void run(const char* name) {
    const int nchunk = 400, nwg = 256;
    const size_t lds = (size_t)(3 * AST + 2 * 6 * NPX) * 16;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nwg * 256 * 4); hipMalloc(&cyc, nwg * 8);
    hipFuncSetAttribute((const void*)k<SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SYNC><<<nwg, 256, lds>>>(out, cyc, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SYNC><<<nwg, 256, lds>>>(out, cyc, nchunk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < nwg; ++i) c += h[i];
    c /= nwg;
    const double nm = 216.0 * nchunk;
    printf("%-34s cycles/MFMA %.1f  (per 72-MFMA step %.0f)  ns/MFMA %.2f  clock %.2f GHz\n", name, c / nm, 72 * c / nm, ms * 1e6 / nm, c / (ms * 1e6));
}
int main() { run<0>("chunk loop, no sync"); run<1>("chunk loop, __syncthreads per step"); return 0; }

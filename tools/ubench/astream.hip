// Micro-benchmark of the conv inner loop: A (weights) streamed from global/L2 with a prefetch ring, B from LDS,
// v_mfma_f32_32x32x2_f32.  Sweeps px tiles per wave (A reuse), lookahead and the weight footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float f4g(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int NP, int R, int BAR, int EXTRA>
__global__ __launch_bounds__(256 + 64 * EXTRA) void k(const float4* __restrict__ w, size_t wmask, float* out, int nunits) {
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += 256 + 64 * EXTRA) sh[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {            // idle partner waves: barriers only
        if (BAR) for (int u0 = 0; u0 < nunits; u0 += R) if ((u0 / R) % BAR == BAR - 1) __syncthreads();
        return;
    }
    f32x16 acc[NP];
    for (int p = 0; p < NP; ++p) acc[p] = (f32x16){0};
    // every workgroup walks the weight buffer from a different start (like different cout tiles / K slices)
    size_t pos = ((size_t)blockIdx.x * 4 + wave) * 977 * 64;
    float4 a[R];
#pragma unroll
    for (int u = 0; u < R - 1; ++u) a[u] = w[((pos + (size_t)u * 64) & wmask) + lane];
    for (int u0 = 0; u0 < nunits; u0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            a[(u + R - 1) % R] = w[((pos + (size_t)(u0 + u + R - 1) * 64) & wmask) + lane];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float b = sh[(lane + 64 * (4 * u + jj) + 1024 * p + u0) & 8191];
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4g(a[u], jj), b, acc[p], 0, 0, 0);
                }
            }
        }
        if (BAR && (u0 / R) % BAR == BAR - 1) __syncthreads();
    }
    float s = 0;
    for (int p = 0; p < NP; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NP, int R, int BAR, int EXTRA> int run(const float4* w, size_t wfloat4, float* out, int wg, const char* tag) {
    const int nunits = 4608 / NP;             // same MFMA count per wave for every NP
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<NP, R, BAR, EXTRA><<<wg, 256 + 64 * EXTRA>>>(w, wfloat4 - 1, out, nunits); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<NP, R, BAR, EXTRA><<<wg, 256 + 64 * EXTRA>>>(w, wfloat4 - 1, out, nunits);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = (double)wg * 4 * nunits * 4 * NP * 4096.0;
    printf("%-8s NP=%d R=%d BAR=%d EXTRA=%d wg=%4d weights=%6.1f MB : %.3f ms  %6.1f TF\n", tag, NP, R, BAR, EXTRA, wg, wfloat4 * 16 / 1e6, ms, fl / ms / 1e9);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 2048 * 256 * 4));
    for (size_t mb : {16}) {
        size_t n4 = mb * (1 << 20) / 16;
        float4* w; CK(hipMalloc(&w, n4 * 16));
        CK(hipMemset(w, 0, n4 * 16));
        for (int wg : {256, 512}) {
            run<1, 6, 0, 0>(w, n4, out, wg, "nobar");
            run<1, 6, 3, 0>(w, n4, out, wg, "bar72");      // barrier every 3*6 units = 72 MFMAs
            run<1, 6, 3, 4>(w, n4, out, wg, "bar72+4");
            run<1, 6, 1, 4>(w, n4, out, wg, "bar24+4");
            run<2, 3, 0, 0>(w, n4, out, wg, "nobar");
            run<2, 3, 6, 0>(w, n4, out, wg, "bar144");
            run<2, 3, 6, 4>(w, n4, out, wg, "bar144+4");
        }
        CK(hipFree(w));
    }
    return 0;
}

// Probe (round 3): can a kernel start before its predecessor in the SAME stream has finished (hipExtAnyOrderLaunch), and what does
// the second-stream alternative look like?  K1 holds 64 workgroups for ~200 us; K2 records when its first instruction ran.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k1(unsigned long long* t, int us) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void k2(unsigned long long* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64();
}
int main() {
    unsigned long long* d; unsigned long long h[3];
    hipMalloc(&d, 64);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, 64); hipDeviceSynchronize();
            hipLaunchKernelGGL(k1, dim3(64), dim3(256), 0, s1, d, 200);
            if (mode == 0) hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, s1, d);
            else if (mode == 1) hipExtLaunchKernelGGL(k2, dim3(64), dim3(256), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            else hipLaunchKernelGGL(k2, dim3(64), dim3(256), 0, s2, d);
            hipDeviceSynchronize();
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("%-44s K1 ran %.1f us; K2 started %.1f us after K1's start (%.1f us %s K1's end)\n",
                   mode == 0 ? "same stream, ordinary launch:" : mode == 1 ? "same stream, hipExtAnyOrderLaunch:" : "second stream:",
                   (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0 >= 0 ? ((double)h[2] - (double)h[1]) / 100.0 : ((double)h[1] - (double)h[2]) / 100.0,
                   h[2] >= h[1] ? "after" : "BEFORE");
        }
    }
    return 0;
}

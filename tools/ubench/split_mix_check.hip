// Check (round 3): the f32 -> f16 hi / lo split written with v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 (5 instructions per pair) gives the bits of the reference
// formulation (cvt, subtract, scale, cvt) on 1 M random pairs over 40 binades, zeros included.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(float xa, float xb, unsigned& p0, unsigned& p1) {
    const f16x2 hi = __builtin_convertvector((f32x2){xa, xb}, f16x2);
    const float ra = __builtin_fmaf((float)hi[0], -1.0f, xa), rb = __builtin_fmaf((float)hi[1], -1.0f, xb);
    const f16x2 lo = __builtin_convertvector((f32x2){ra * 2048.f, rb * 2048.f}, f16x2);
    p0 = __builtin_bit_cast(unsigned, hi); p1 = __builtin_bit_cast(unsigned, lo);
}
__device__ __forceinline__ void split_mix(float xa, float xb, unsigned& p0, unsigned& p1) {
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){xa, xb}, f16x2));
    float ra, rb; unsigned lo;
    const float sc = 2048.f;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(xa));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(xb));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(lo) : "v"(ra), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(rb), "s"(sc));
    p0 = hi; p1 = lo;
}
__global__ void k(const float* x, unsigned* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned a0, a1, b0, b1;
    split_ref(x[2 * i], x[2 * i + 1], a0, a1);
    split_mix(x[2 * i], x[2 * i + 1], b0, b1);
    o[4 * i] = a0; o[4 * i + 1] = a1; o[4 * i + 2] = b0; o[4 * i + 3] = b1;
}
int main() {
    const int n = 1 << 20;
    float* hx = new float[2 * n]; unsigned* ho = new unsigned[4 * n];
    unsigned st = 12345;
    for (int i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; unsigned e = 100 + (st >> 8) % 40; unsigned bits = ((st & 1) << 31) | (e << 23) | ((st >> 9) & 0x7fffff); if (i % 97 == 0) bits = 0; memcpy(&hx[i], &bits, 4); }
    float* dx; unsigned* d;
    hipMalloc(&dx, 8 * n); hipMalloc(&d, 16 * n);
    hipMemcpy(dx, hx, 8 * n, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d, n);
    hipMemcpy(ho, d, 16 * n, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) if (ho[4 * i] != ho[4 * i + 2] || ho[4 * i + 1] != ho[4 * i + 3]) { if (bad < 5) printf("mismatch %d: x %g %g ref %08x %08x mix %08x %08x\n", i, hx[2*i], hx[2*i+1], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]); ++bad; }
    printf("pairs %d mismatches %ld\n", n, bad);
    return bad != 0;
}

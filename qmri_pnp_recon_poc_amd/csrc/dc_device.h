// dc_device.h -- device-side helpers shared by the data-consistency kernels (dc_kernels.hip, lsqr_kernels.hip).
#pragma once
#include "qmri_internal.h"
#include "fft_codelets.h"

namespace dcdev {
using namespace qfft;

constexpr int NT = 256;          // threads per block for every kernel in this file
constexpr int DC_MAXS = 10;      // channel lines the w-pass kernels hold in LDS (the reference always has s = 10)
constexpr int DC_VCAP = 2048;    // doubles of V kept in LDS by the w-pass kernels (T*s <= 2048, e.g. T = 200, s = 10)
constexpr int DC_CH = 512;       // samples of a k-row staged in LDS at a time by the adjoint scatter

// Write-through (sc1) store of one complex double.  A plain store leaves its line dirty in the XCD's L2 and the kernel boundary then
// writes all of them back before the next (dependent) kernel starts (+ B / 6 TB/s, MI355X_MICROARCH.md price list, row "boundary");
// a write-through store sends the bytes while other workgroups still compute.  Every kernel here hands its output to another launch.
__device__ __forceinline__ void st_wt(double2* p, double2 v) {
    typedef double d2v_ __attribute__((ext_vector_type(2)));
    const d2v_ t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it stalls until every global
// store the wave has issued is acknowledged; the kernels here never communicate through global memory inside a
// block, so their stores are left in flight.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// lane exchange inside groups of 8 lanes on the VALU (DPP), no LDS traffic
template <int CTRL> __device__ __forceinline__ double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_HALF_MIRROR = 0x141, DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;   // lane i <-> 7-i ; quad_perm [1,0,3,2] ; [2,3,0,1]
// a[0..7] of an aligned group of 8 lanes -> ((a0+a7)+(a1+a6)) + ((a2+a5)+(a3+a4)) in lane 0 of the group (fixed tree)
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_get<DPP_HALF_MIRROR>(v);
    v += dpp_get<DPP_XOR1>(v);
    v += dpp_get<DPP_XOR2>(v);
    return v;
}

// Sum over the 64 lanes of a wave in one fixed order, valid in EVERY lane: four DPP steps inside each row of 16 lanes (pairs, quads, halves, row),
// then the four row sums through v_readlane, ((r0 + r1) + (r2 + r3)).  Round 3: the __shfl_down trees used before compile to ds_bpermute_b32
// (12 dependent LDS-crossbar round trips per double, ~0.3 us); the one-launch LSQR iteration does four such reductions per iteration.
constexpr int DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ double wave_sum_all(double v) {
    v += dpp_get<DPP_XOR1>(v);
    v += dpp_get<DPP_XOR2>(v);
    v += dpp_get<DPP_HALF_MIRROR>(v);
    v += dpp_get<DPP_ROW_MIRROR>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum_all(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_barrier();
    if (lane == 0) sh[wid] = v;
    lds_barrier();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
    return r;
}

// Two-step FFT of `nlines` lines held in LDS (natural order, pitch LINE).  On return thread (line2,k1) holds
// X[k1 + R1*k2] in out[k2].  LINE_FAST selects the step-2 thread layout: line fastest or k1 fastest.
template <int R1, int R2, bool LINE_FAST>
__device__ __forceinline__ bool fft_lds(cd* lds, int nlines, const double2* __restrict__ tw, cd* out, int& line2, int& k1) {
    typedef Plan<R1, R2> P;
    const int tid = threadIdx.x;
    const int line1 = tid / R2, n2 = tid - line1 * R2;
    const bool act1 = tid < nlines * R2;
    cd a[R1];
    lds_barrier();
    if (act1) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) a[n1] = lds[line1 * P::LINE + R2 * n1 + n2];
        Dft<R1>::run(a);
#pragma unroll
        for (int q = 1; q < R1; ++q) a[q] = mul(a[q], tw[n2 * q]);
    }
    lds_barrier();
    if (act1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) lds[line1 * P::LINE + P::SP * n2 + q] = a[q];
    }
    lds_barrier();
    if (LINE_FAST) { k1 = tid / nlines; line2 = tid - k1 * nlines; }
    else { line2 = tid / R1; k1 = tid - line2 * R1; }
    const bool act2 = tid < nlines * R1;
    if (act2) {
#pragma unroll
        for (int q = 0; q < R2; ++q) out[q] = lds[line2 * P::LINE + P::SP * q + k1];
        Dft<R2>::run(out);
    }
    return act2;
}

template <int R1, int R2> struct Cfg {
    typedef Plan<R1, R2> P;
    static constexpr int L = (16 * P::LINE * 16 <= 65536 && 16 * (R1 > R2 ? R1 : R2) <= NT) ? 16 : 8;   // lines per h-pass block
};


// two sums at once (one pair of barriers)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
    a = wave_sum_all(a); b = wave_sum_all(b);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_barrier();
    if (lane == 0) { sh[wid] = a; sh[NT / 64 + wid] = b; }
    lds_barrier();
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { ra += sh[i]; rb += sh[NT / 64 + i]; }
    a = ra; b = rb;
}

// Canonical sum of n partials, computed by one wave: 64 strided columns, then a shuffle tree.  Every block of both
// kernels calls this with the same arrays, so alpha and beta carry the same bits everywhere.  Result valid in lane 0.
__device__ __forceinline__ double wave_sum(const double* __restrict__ p, int n) {
    const int lane = threadIdx.x & 63;
    double r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int i = lane + 64 * q; r[q] = p[(i < n) ? i : 0]; }      // all in flight together
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) a += (lane + 64 * q < n) ? r[q] : 0.0;
    for (int i = lane + 512; i < n; i += 64) a += p[i];
    return wave_sum_all(a);
}

// u(m+1:end) update, shared by both kernels so that they produce the same bits:  sqrt(r) v - alpha (u / beta_prev)
__device__ __forceinline__ double2 ub_update(double2 v, double2 ub, double sr, double alpha, double inv_bprev) {
    return make_double2(__fma_rn(v.x, sr, -(alpha * (ub.x * inv_bprev))), __fma_rn(v.y, sr, -(alpha * (ub.y * inv_bprev))));
}

}  // namespace dcdev

// conv6_act.h -- the low-magnitude guard of the f16 operand split (conv6_kernels.hip), device pieces shared with the kernel that runs right after a
// forward pass inside the PnP-ADMM loop (dc_kernels.hip, k_dual_fwd_h): the per-layer reduction of the |output| reports rides in that launch
// instead of one of its own (k_act_check, which the other callers of the network keep).
#pragma once
#include <hip/hip_runtime.h>
#include "qmri_internal.h"

// The low side of the f16 split: an activation below 2^-14 has a subnormal hi piece, i.e. an ABSOLUTE error of ~2^-36 instead of a
// relative one of 2^-22.  That is harmless while the tensor it belongs to has ordinary magnitudes (the error is relative to the
// tensor's largest entries, as in any fp32 dot product) and harmful when a WHOLE layer output is tiny -- the next layer then
// amplifies the absolute error.  Every kernel of the f16 scheme therefore reports the largest |output| of its tensor: one value per
// wave into a slot array (plain stores, no atomics, no fences), reduced per layer at the end of the forward pass.
// What "tiny" means is calibrated per layer: the set-up probe of qmri_set_denoiser (f16 kernels against f32-MFMA kernels, end to end)
// has shown the network accurate with each layer at the magnitude the probe produced, and records those magnitudes; a forward pass
// raises bit 1 of the range flag when a layer comes out below ACT_LOW AND more than ACT_DROP below its calibrated magnitude -- this
// image makes the layer collapse where the probe did not.  (Layers that are tiny for every input -- the deep levels of the
// synthetic bench network sit at 1e-8 -- were already tiny under the probe and are covered by its end-to-end comparison.)
// The callers answer the bit like the overflow bit: the network is re-packed for the bf16 scheme and the call is repeated.
constexpr float ACT_LOW = 0x1p-12f;   // 2.4e-4: below it the f16 pieces carry less than fp32's relative accuracy
constexpr float ACT_DROP = 0x1p-10f;  // ... and this far below the calibrated magnitude of the layer
constexpr int ACT_MAXSLOT = 1 << 17;  // slots per layer (one per wave of the reporting launch)

// record != 0 (the set-up probe): store the maximum as the layer's calibrated magnitude; else raise bit 1 of the range flag if the layer collapsed.
// host_words (pinned, or null): [0] = the overflow bit the conv kernels of this pass raised, [1 + layer] = 2 if that layer collapsed -- the
// host reads them after its next synchronisation instead of copying the device flag back after every forward pass.  nlayers = 0: nothing to do.
// (struct ActCheckArgs: qmri_internal.h)

// largest of a wave's non-negative values, as its bit pattern in an SGPR: four DPP steps on the VALU inside each row of 16 lanes
// (no LDS round trips at the very end of a kernel), then the four rows through v_readlane and scalar max (non-negative floats
// order like their bit patterns)
__device__ __forceinline__ unsigned wave_max_bits(float v) {
    int x = __builtin_bit_cast(int, v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));   // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));   // row_mirror
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return (unsigned)max(max(a, b), max(c, d));
}

// One 256-thread workgroup finishes one layer: the largest |output| over the slots its kernels reported; the count is cleared for the next pass.
// red: 4 floats of LDS.  Every thread of the workgroup must call it (barrier inside).
__device__ __forceinline__ void act_check_layer(const ActCheckArgs& a, int layer, float* red) {
    const int n = a.count[layer];
    const float* row = a.slots + (size_t)layer * ACT_MAXSLOT;
    float m = 0.f;
    for (int i0 = 0; i0 < n; i0 += 256 * 8) {                       // eight independent loads in flight per thread: one memory latency
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = i0 + threadIdx.x + 256 * q; v[q] = row[(i < n) ? i : 0]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) m = fmaxf(m, (i0 + (int)threadIdx.x + 256 * q < n) ? v[q] : 0.f);
    }
    m = __builtin_bit_cast(float, wave_max_bits(m));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        bool low = false;
        if (n > 0) {
            m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (a.record) a.ref[layer] = m;
            else if (m > 0.f && m < ACT_LOW && m < a.ref[layer] * ACT_DROP) { atomicOr(a.range_flag, 2u); low = true; }
            a.count[layer] = 0;
        }
        if (a.host_words) {
            a.host_words[1 + layer] = low ? 2u : 0u;
            if (layer == 0) a.host_words[0] = __hip_atomic_load(a.range_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 5u;   // (bits 0 and 2 -- overflow, k_conv6r hand-off time-out: set by kernels that are over)
        }
    }
}

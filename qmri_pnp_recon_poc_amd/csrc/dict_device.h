// dict_device.h -- device pieces shared by the dictionary-match kernels (dict_kernels.hip: s <= 16 channels, atoms against register-resident pixels;
// dictw_kernels.hip: s <= 1024 channels, a channel-blocked GEMM), the Xfit kernel and the TSMI synthesis (synth_kernels.hip).
//
// Reference semantics: main_files/dictionary_matching/mrf_dtm_cpu.m
//   :92      [mt,dm] = max(abs(ip),[],1)               single-precision MAGNITUDES compared, first index wins ties
//   :94-96   pd = ip(dm); X = pd .* D(dm,:); pd = pd ./ normD(dm)
//   :136-160 qmap = lut(dm,:) (NaN -> 0), pd, mt, dm (1-based)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The floats whose correctly rounded square root is s: lo .. hi (two or three consecutive floats share one root).
// sqrtf() is the correctly rounded square root under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt (the __fsqrt_rn
// intrinsic is NOT: without OCML_BASIC_ROUNDED_OPERATIONS it is the 1-ulp native instruction) -- bit-identical to glibc's sqrtf.
// x rounds to s iff (prev(s) + s)/2 < sqrt(x) < (s + next(s))/2 (a tie is impossible: a midpoint has 25 significant bits, its square
// an odd 50th one, x only 24), i.e. iff mid_lo^2 < x < mid_hi^2 with both squares exact in double precision.
__device__ __forceinline__ void sqrt_preimage(float s, float m2, float& lo, float& hi) {
    if (!(s >= 1e-30f && s <= 1e30f)) {                                 // zero, tiny, infinite or NaN: walk (never in practice)
        lo = hi = m2;
        for (int it = 0; it < 4; ++it) { const float n = __uint_as_float(__float_as_uint(hi) + 1u); if (sqrtf(n) == s) hi = n; else break; }
        for (int it = 0; it < 4 && lo > 0.f; ++it) { const float n = __uint_as_float(__float_as_uint(lo) - 1u); if (sqrtf(n) == s) lo = n; else break; }
        return;
    }
    const double sd = (double)s;
    const double mid_hi = 0.5 * (sd + (double)__uint_as_float(__float_as_uint(s) + 1u));
    const double mid_lo = 0.5 * (sd + (double)__uint_as_float(__float_as_uint(s) - 1u));
    const double bh = mid_hi * mid_hi, bl = mid_lo * mid_lo;            // exact
    hi = (float)bh;                                                     // nearest float; step down if it did not land below bh
    if ((double)hi >= bh) hi = __uint_as_float(__float_as_uint(hi) - 1u);
    lo = (float)bl;                                                     // ... step up if it did not land above bl
    if ((double)lo <= bl) lo = __uint_as_float(__float_as_uint(lo) + 1u);
}

// The incumbent of a lane (one pixel column of a 32 x 32 MFMA tile, the rows of its lane half): best = abs(ip), thr = just below the
// pre-image of best under sqrtf (see inc_update), (cre, cim) = ip, bidx = atom (0-based).
struct Inc { float best, thr, cre, cim; int bidx; };

// One finished 32-atom x 32-pixel tile of products (C/D layout of v_mfma_f32_32x32x2_f32: register r of lane (j, h) is row
// (r & 3) + 8 (r >> 2) + 4 h, column j) against the incumbent.  t = the tile's index (atoms 32 t ..), h = lane >> 5.
// max(abs(ip)) compares single-precision MAGNITUDES: two atoms whose |ip|^2 differ in the last bits but whose sqrtf rounds to the
// same single tie, and the first index wins (mrf_dtm_cpu.m:92).  That semantics is kept without a square root per candidate: `thr` is
// the largest float just BELOW the incumbent's pre-image, a tile is looked at only when its largest |ip|^2 exceeds it, and an atom of
// EQUAL magnitude wins only with the lower index -- so the tiles may be visited in any order.
// |ip|^2 = fma(im, im, re * re), the bits the oracle computes -- one v_mul_f32 and one v_fma_f32 per row, NOT the packed forms (the files
// are compiled with -fno-slp-vectorize): beside MFMAs a v_pk_fma_f32 costs the wave ~22 cycles more than the two plain instructions it
// replaces (MI355X_MICROARCH.md, constants table).
__device__ __forceinline__ void inc_update(int t, int h, const f32x16& are, const f32x16& aim, Inc& I) {
    float &best = I.best, &thr = I.thr, &cre = I.cre, &cim = I.cim;
    int& bidx = I.bidx;
    float m2[16], tmax;
#pragma unroll
    for (int r = 0; r < 16; ++r) m2[r] = __builtin_fmaf(aim[r], aim[r], are[r] * are[r]);
    tmax = fmaxf(fmaxf(fmaxf(fmaxf(m2[0], m2[1]), fmaxf(m2[2], m2[3])), fmaxf(fmaxf(m2[4], m2[5]), fmaxf(m2[6], m2[7]))),
                 fmaxf(fmaxf(fmaxf(m2[8], m2[9]), fmaxf(m2[10], m2[11])), fmaxf(fmaxf(m2[12], m2[13]), fmaxf(m2[14], m2[15]))));
    if (tmax > thr) {          // some atom of this tile reaches the incumbent's magnitude
        // the tile's magnitude is sqrtf of its largest |ip|^2; MATLAB's max keeps the FIRST atom with the largest magnitude, i.e. the
        // lowest row of the tile whose |ip|^2 lies in the root's pre-image [lo, hi]: one square root per update, not per candidate.
        float lo, hi;
        const float mag = sqrtf(tmax);
        sqrt_preimage(mag, tmax, lo, hi);
        int rsel = 15;                                                  // (the tile's maximum itself is >= lo: some row qualifies)
        float nre = are[15], nim = aim[15];
#pragma unroll
        for (int r = 14; r >= 0; --r) if (m2[r] >= lo) { rsel = r; nre = are[r]; nim = aim[r]; }     // (ascending rows = ascending atoms)
        const int nidx = t * 32 + (rsel & 3) + 8 * (rsel >> 2) + 4 * h;    // C/D row of the 32x32 MFMA tile
        if (mag > best || nidx < bidx) {                                // (mag >= best here: tmax > thr means tmax >= the incumbent's lo)
            best = mag; bidx = nidx; cre = nre; cim = nim;
            // (magnitude 0 -- an all-zero pixel -- has nothing below it: there the incumbent is row 0 of the wave's FIRST tile,
            //  which must be visited first and be the wave's lowest index, so only a non-zero product may come here again)
            thr = (lo > 0.f) ? __uint_as_float(__float_as_uint(lo) - 1u) : 0.f;
        }
    }
}

// A candidate o against the incumbent b, both (|ip|, atom index bits, re, im): larger magnitude, then lower index -- max(abs(ip)) with the
// first index winning ties, whatever the split of the atoms over lanes, waves, workgroups (mrf_dtm_cpu.m:92)
__device__ __forceinline__ bool cand_better(float ob, int oi, float best, int bidx) { return ob > best || (ob == best && oi < bidx); }

// The outputs of one pixel from its winner (mrf_dtm_cpu.m:94-96,136-160).  win (nullable): (re ip, im ip, atom index bits, |ip|) kept for k_dict_xfit.
__device__ __forceinline__ void finish_pixel(int p, int Npix, int K, float best, int bidx, float cre, float cim, const float* __restrict__ normD,
                                             const float* __restrict__ lut, int Q, float* __restrict__ qmap, float* __restrict__ pd,
                                             float* __restrict__ mt, int32_t* __restrict__ dm, float4* __restrict__ win) {
    if (bidx >= K || bidx < 0) bidx = 0;                             // cannot happen: padded atoms are all-zero and never beat a real one
    const float nd = normD[bidx];
    if (dm) dm[p] = bidx + 1;                                        // 1-based  :92,:156-160
    if (mt) mt[p] = best;                                            // :150-154
    if (pd) { pd[2 * (size_t)p] = cre / nd; pd[2 * (size_t)p + 1] = cim / nd; }     // :96,:144-148
    if (qmap)
        for (int q = 0; q < Q; ++q) {
            const float v = lut[(size_t)bidx + (size_t)K * q];
            qmap[(size_t)p + (size_t)Npix * q] = (v != v) ? 0.f : v;                  // NaN -> 0  :138
        }
    if (win) win[p] = make_float4(cre, cim, __int_as_float(bidx), best);
}

// D(a, c) from the device copy of the dictionary, whichever fragment order it is packed in (qmri_set_dictionary):
//   narrow (s <= 16):  pack[tile32][lane][npl],       lane = (a & 31) + 32 (c & 1), entry c >> 1            (dict_kernels.hip)
//   wide   (s  > 16):  pack[tile32][G8][lane][4],     group c >> 3, lane = (a & 31) + 32 (c & 1), entry (c & 7) >> 1  (dictw_kernels.hip)
struct DictView { const float* pack; int wide, npl, G8; };
__device__ __forceinline__ float dict_atom(const DictView& v, int a, int c) {
    const int lane = (a & 31) + 32 * (c & 1);
    if (v.wide) return v.pack[(((size_t)(a >> 5) * v.G8 + (c >> 3)) * 64 + lane) * 4 + ((c & 7) >> 1)];
    return v.pack[((size_t)(a >> 5) * 64 + lane) * v.npl + (c >> 1)];
}

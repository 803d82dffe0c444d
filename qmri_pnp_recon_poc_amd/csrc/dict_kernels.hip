// dict_kernels.hip -- MRF dictionary template match on the matrix cores (gfx950): exact single-precision products (f32 MFMA) behind an f16
// filter that decides which 32-atom tiles can hold the winner (k_dict_match_f, the default); k_dict_match is the exact products alone
// (dictionaries with non-finite entries, qmri_debug_dict_filter(ctx, 0, ..)).  Both give the oracle's bits.
//
// Reference semantics: main_files/dictionary_matching/mrf_dtm_cpu.m
//   :54      x = single(x)
//   :91      ip = dict.D * ctranspose(x(cind,:))      (K x s) * (s x B): ip(j,p) = sum_c D(j,c) conj(x(p,c))
//   :92      [mt,dm] = max(abs(ip),[],1)               first index wins ties
//   :94-96   pd = ip(dm) / normD(dm)
//   :136-160 qmap = lut(dm,:) (NaN -> 0), pd, mt, dm (1-based)
// The reference materialises ip in blocks of <= 1e9 elements (:74) and then walks pixels in an interpreted
// loop; here the K x Npix product never exists: each wave keeps one 32-pixel column tile's X fragment in
// registers, streams 32-atom tiles of D through v_mfma_f32_32x32x2_f32 (real and imaginary chains) and
// reduces |ip|^2 to a running (max, argmax) in the epilogue.  A f32 MFMA accumulates k in order with one
// fma per product, so ip is bit-identical to a sequential fmaf chain over c = 0..s-1 -- the arithmetic the
// oracle spells out -- and the argmax can be checked bit-exactly.
// max(abs(ip)) compares single-precision MAGNITUDES: two atoms whose |ip|^2 differ in the last bits but whose
// sqrtf rounds to the same single tie, and the first index wins (:92).  The loop keeps that semantics without a
// square root per candidate: beside the best magnitude it holds `thr`, the largest |ip|^2 whose correctly rounded
// square root is still that magnitude; a candidate can replace the incumbent iff its |ip|^2 reaches the pre-image (checked once per
// 32-atom tile on the tile's maximum; thr sits just below the pre-image and an atom of EQUAL magnitude wins only with the lower
// index, so the tiles may be visited in any order).  The square root and the bounds of its pre-image are evaluated only when a tile
// gets that far.
//
// Work split of k_dict_match: a workgroup (4 waves) takes one 32-pixel tile and one of P contiguous parts of the atom tiles; wave w takes the part's
// tiles w, w+4, ...; the four (max, argmax) candidates per pixel are merged through LDS preferring the lower index on ties.  P > 1
// when the pixel tiles alone do not fill the device evenly: a workgroup walks ALL its atoms (1.1 ms at K = 98 304), so 1568 pixel
// tiles on 1280 resident workgroups took two rounds, the second at 22 % occupancy -- the matrix pipe idled 40 % of the launch.  The
// parts' candidates go to a scratch array and k_dict_merge picks per pixel: larger magnitude, then lower index -- the same rule,
// so max(abs(ip)) with the first index winning ties (mrf_dtm_cpu.m:92) whatever the split.
#include <algorithm>
#include "qmri_internal.h"
#include "dict_device.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NT = 256;
constexpr int MAXPAIR = 8;      // s <= 16 here; more channels: dictw_kernels.hip

// One 32-atom tile through the exact products and the incumbent rule (dict_device.h).  The file is compiled with -amdgpu-mfma-vgpr-form
// (Makefile): the products land in VGPRs and the epilogue reads them in place -- with AGPR accumulators 32 of its 80 vector instructions
// per tile were v_accvgpr_read, and the epilogue's issue slots, not the matrix pipe, set the pace (10 MFMAs per 32 x 32 outputs).
template <int NPAIR, int NV>
__device__ __forceinline__ void exact_tile(int t, int h, const f32x4 (&av)[NV], const float (&bre)[NPAIR], const float (&bim)[NPAIR], Inc& I) {
    f32x16 are = {0}, aim = {0};
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
        are = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], bre[q], are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q >> 2][q & 3], bim[q], aim, 0, 0, 0);
    }
    inc_update(t, h, are, aim, I);
}

// D packed as MFMA A-fragments, all pairs of a lane together: pack[tile][lane][q] = D[tile*32 + (lane&31)][2q + (lane>>5)]  (0 beyond K or s;
// NPL = 4 or 8 floats per lane, so a tile is one or two 16-byte requests per lane instead of one 4-byte request per pair)
template <int NPAIR>
__global__ __launch_bounds__(NT) void k_dict_match(const double2* __restrict__ X, int Npix, int s, const float* __restrict__ pack,
                                                    int ntiles_all, int K, const float* __restrict__ normD,
                                                    const float* __restrict__ lut, int Q, float* __restrict__ qmap,
                                                    float* __restrict__ pd, float* __restrict__ mt, int32_t* __restrict__ dm,
                                                    float4* __restrict__ part, float4* __restrict__ win) {
    __shared__ float s_best[4][32];
    __shared__ int s_idx[4][32];
    __shared__ float s_re[4][32];
    __shared__ float s_im[4][32];
    // (the wave number as a SCALAR: the tile counter, its bound checks and the fragment pointers of the loop below then live on the scalar
    //  unit -- round 3: a PMC pass showed the f32 MFMA and the vector ALU not to overlap (MFMA busy 0.67 + VALU 0.35 of the launch), so every
    //  vector instruction of the loop is paid in full; 8 of its ~43 were tile-address arithmetic)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int p = blockIdx.x * 32 + j;
    // B fragments: B[k = 2q + h][j] = x(p, c = 2q + h)  (real chain) and -imag (conjugate) for the imaginary chain
    float bre[NPAIR], bim[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
        const int c = 2 * q + h;
        double2 v = make_double2(0.0, 0.0);
        if (p < Npix && c < s) v = X[(size_t)p + (size_t)Npix * c];
        bre[q] = (float)v.x;                   // single(x)  mrf_dtm_cpu.m:54
        bim[q] = -(float)v.y;                  // conj
    }
    Inc I = {-1.0f, -1.0f, 0.f, 0.f, 0};
    float &best = I.best, &cre = I.cre, &cim = I.cim;
    int& bidx = I.bidx;
    // this workgroup's part of the atom tiles: [tbeg, ntiles)
    const int tper = (ntiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int tbeg = (int)blockIdx.y * tper, ntiles = min(ntiles_all, tbeg + tper);
    // The atom fragments of tile t + 4 are requested before the products of tile t (register double buffer): the loop used to
    // request a tile's fragments and wait for them at once, one L2 latency per tile hidden only by occupancy.
    constexpr int NPL = (NPAIR <= 4) ? 4 : 8, NV = NPL / 4;
    f32x4 av[NV], avn[NV];
    auto tile_body = [&](int t, const f32x4 (&av)[NV]) __attribute__((always_inline)) { exact_tile<NPAIR, NV>(t, h, av, bre, bim, I); };
    // Visiting order.  Atoms lie on a (T1, T2) grid and neighbouring pixels have neighbouring matches: in ascending order the scan climbs
    // towards the match row by row and 28 % of the tiles brought a new incumbent for some lane of the wave (measured on the bench slice;
    // the update path costs three times the plain tile).  A wave therefore walks its tiles i = 0 .. n-1 (tile tbeg + wave + 4 i) in
    // BIT-REVERSED order of i -- coarse to fine over its whole part: the incumbent is close to the final one after a few tiles and 4.5 %
    // of the tiles reach the update path.  The order is scalar arithmetic (s_brev_b32); the result does not depend on it (see above).
    const int n = (ntiles - tbeg - wave + 3) / 4;                      // this wave's tiles (<= 0: none)
    int nb = 0;
    while ((1 << nb) < n) ++nb;
    const unsigned kend = 1u << nb;
    auto next_i = [&](unsigned& k) __attribute__((always_inline)) -> int {          // next valid i at or after counter k (n when exhausted); k moves past it
        while (k < kend) {
            const int i = nb ? (int)(__builtin_bitreverse32(k) >> (32 - nb)) : 0;
            ++k;
            if (i < n) return i;
        }
        return n;
    };
    auto request = [&](int i, f32x4 (&dst)[NV]) __attribute__((always_inline)) {
        const f32x4* ap = (const f32x4*)(pack + ((size_t)(tbeg + wave + 4 * i) * 64 + lane) * NPL);
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = ap[v];
    };
    {
        unsigned k = 0;
        int i0 = (n > 0) ? next_i(k) : n;
        if (i0 < n) request(i0, av);
        while (i0 < n) {
            const int i1 = next_i(k);
            if (i1 < n) request(i1, avn);
            tile_body(tbeg + wave + 4 * i0, av);
            if (i1 >= n) break;
            i0 = next_i(k);
            if (i0 < n) request(i0, av);
            tile_body(tbeg + wave + 4 * i1, avn);
        }
    }
    // merge the two lane halves (same pixel, interleaved atom rows): larger value, then lower index
    {
        const float ob = __shfl(best, lane ^ 32, 64), ore = __shfl(cre, lane ^ 32, 64), oim = __shfl(cim, lane ^ 32, 64);
        const int oi = __shfl(bidx, lane ^ 32, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; cre = ore; cim = oim; }
    }
    if (h == 0) { s_best[wave][j] = best; s_idx[wave][j] = bidx; s_re[wave][j] = cre; s_im[wave][j] = cim; }
    __syncthreads();
    if (tid < 32 && p < Npix) {
        best = s_best[0][tid]; bidx = s_idx[0][tid]; cre = s_re[0][tid]; cim = s_im[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = s_best[w][tid];
            const int oi = s_idx[w][tid];
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; cre = s_re[w][tid]; cim = s_im[w][tid]; }
        }
        if (bidx >= K) bidx = 0;       // cannot happen: padded atoms are all-zero and never beat a real one
        if (part) {                    // atoms split over workgroups: k_dict_merge finishes the pixel
            part[(size_t)blockIdx.y * Npix + p] = make_float4(best, __int_as_float(bidx), cre, cim);
            return;
        }
        finish_pixel(p, Npix, K, best, bidx, cre, cim, normD, lut, Q, qmap, pd, mt, dm, win);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The match with the f16 filter in front of the exact products (the default; k_dict_match above when D cannot be cut into f16 pieces).
//
// Per pixel the channels are scaled by a power of two to max |component| in [0.5, 1) and cut into f16 pieces hi + lo (22 bits); D likewise
// on the host with one power of two g for the whole dictionary (pack16).  Three f16 products (lo hi, hi lo, hi hi; K = 16 >= s in ONE
// v_mfma_f32_32x32x16_f16 each) give ip to |error| <= 2^-19 B, B = g R |x| >= |ip| (R the largest row norm of D) -- the pieces' rounding
// 5 * 2^-22 B, the accumulation <= 2^-20 B -- and the exact chain itself is within 2^-20 B of the real product.
// So |ip|^2 of the atom MATLAB's max picks (and of every atom of the same magnitude) is, as the filter sees it, within 2^-15.8 B^2 of
// the largest filtered |ip|^2 seen so far, and a tile goes to the exact products iff some lane's filtered maximum comes within
// marg = 2^-14 B^2 of its running maximum.  Nothing else depends on the filter: the listed tiles go through exact_tile(), whose rule
// does not depend on the order.  192 matrix-core cycles per tile instead of 640, and ~5 % of the tiles listed (bench slice).
//
// Work split: a workgroup takes FOUR 32-pixel tiles (one per wave) and one of P parts of the atom tiles; every wave walks all tiles of
// the part.  The f16 fragments pass through LDS in steps of FSTEP tiles, loaded once per workgroup (each wave fetching its own copy, as
// the exact kernel does, asked the L2 for 94 GB/s per CU at the matrix cores' pace -- more than a CU gets).  Steps are visited in
// bit-reversed order (see k_dict_match).
constexpr int LCAP = 128;       // tiles a wave collects for the exact products before it works them off
constexpr int FSTEP = 8;        // tiles per LDS step: 16 KB of f16 pieces, [tile][hi | lo][lane] of 16 bytes
template <int NPAIR, bool SEED>
__global__ __launch_bounds__(NT) void k_dict_match_f(const double2* __restrict__ X, int Npix, int s, const float* __restrict__ pack,
                                                      int ntiles_all, int tper, int K, const float* __restrict__ normD,
                                                      const float* __restrict__ lut, int Q, float* __restrict__ qmap,
                                                      float* __restrict__ pd, float* __restrict__ mt, int32_t* __restrict__ dm,
                                                      float4* __restrict__ part, const uint4* __restrict__ pack16, float marg_coef,
                                                      int* __restrict__ gmax, int seed_stride, float4* __restrict__ win) {
    __shared__ uint4 s_a[2][FSTEP * 128];
    __shared__ int s_list[4][LCAP];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 31, h = lane >> 5;
    const int p = ((int)blockIdx.x * 4 + wave) * 32 + j;
    constexpr int NPL = (NPAIR <= 4) ? 4 : 8, NV = NPL / 4;
    // exact products: B[k = 2q + h][j] = x(p, c = 2q + h) (real chain) and -imag (conjugate) for the imaginary chain
    float bre[NPAIR], bim[NPAIR];
#pragma unroll
    for (int q = 0; q < NPAIR; ++q) {
        const int c = 2 * q + h;
        double2 v = make_double2(0.0, 0.0);
        if (p < Npix && c < s) v = X[(size_t)p + (size_t)Npix * c];
        bre[q] = (float)v.x;                   // single(x)  mrf_dtm_cpu.m:54
        bim[q] = -(float)v.y;                  // conj
    }
    // filter: B[k = 8 h + jj][j], pieces of the scaled pixel
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    f16x8 brh, brl, bih, bil;
    float marg;
    {
        float xr[8], xi[8], mx = 0.f, n2 = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int c = 8 * h + jj;
            double2 v = make_double2(0.0, 0.0);
            if (p < Npix && c < s) v = X[(size_t)p + (size_t)Npix * c];
            xr[jj] = (float)v.x; xi[jj] = -(float)v.y;
            mx = fmaxf(mx, fmaxf(fabsf(xr[jj]), fabsf(xi[jj])));
            n2 = fmaf(xr[jj], xr[jj], fmaf(xi[jj], xi[jj], n2));
        }
        mx = fmaxf(mx, __shfl(mx, lane ^ 32, 64));
        n2 += __shfl(n2, lane ^ 32, 64);
        const bool zero = (mx == 0.f) && (n2 == 0.f);
        const bool ok = zero || (mx > 1e-30f && mx < 1e30f && n2 == n2);
        int e = 0;
        if (ok && !zero) (void)frexpf(mx, &e);
        const float sc = ldexpf(1.f, -e);                               // max |component| * sc in [0.5, 1)
        float n2s = 0.f;                                                // |x|^2 of the scaled pixel (this lane's channels)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float a = ok ? xr[jj] * sc : 0.f, b = ok ? xi[jj] * sc : 0.f;
            const _Float16 ah = (_Float16)a, bh = (_Float16)b;
            brh[jj] = ah; brl[jj] = (_Float16)(a - (float)ah);
            bih[jj] = bh; bil[jj] = (_Float16)(b - (float)bh);
            n2s = fmaf(a, a, fmaf(b, b, n2s));
        }
        n2s += __shfl(n2s, lane ^ 32, 64);
        // (a pixel with a non-finite, tiny or huge channel is not filtered: marg = +inf sends every tile to the exact products)
        marg = ok ? marg_coef * n2s * 1.001f : __builtin_inff();
    }
    // The running maximum starts from a SEED: gmax[p] holds the largest filtered |ip|^2 of pixel p over a coarse sample of the whole
    // dictionary (every seed_stride-th step), left there by a first launch of this kernel with SEED = true (3 % of the filter's work; -1
    // when the dictionary is small and that launch is skipped).  Any value found there belongs to an atom of the dictionary, so the
    // argument above holds with it.  What it buys: without it every wave climbs from -1 in each of the P atom parts and 7 % of the tiles
    // went to the exact products (21 parts of 152 tiles on the bench slice); within the margin of the final maximum are ~0.5 %.
    const bool pub = (marg < __builtin_inff()) && p < Npix;
    Inc I = {-1.0f, -1.0f, 0.f, 0.f, 0};
    float runa = (!SEED && pub) ? __int_as_float(gmax[p]) : -1.f, cut = runa - marg;
    int cnt = 0;                                                        // listed tiles (scalar)
    // this workgroup's part of the atom tiles: [tbeg, tbeg + n), local index i
    // (SEED: the whole dictionary, steps (blockIdx.y + gridDim.y k) seed_stride)
    const int tbeg = SEED ? 0 : (int)blockIdx.y * tper, n = SEED ? ntiles_all : min(ntiles_all - tbeg, tper);
    const int nsteps = (n + FSTEP - 1) / FSTEP;
    int nb = 0;
    while ((1 << nb) < nsteps) ++nb;
    const unsigned kend = 1u << nb;
    auto next_step = [&](unsigned& k) __attribute__((always_inline)) -> int {       // next valid step at or after counter k (nsteps when exhausted)
        if constexpr (SEED) {
            const long i = ((long)blockIdx.y + (long)gridDim.y * k) * seed_stride;
            ++k;
            return i < nsteps ? (int)i : nsteps;
        }
        while (k < kend) {
            const int i = nb ? (int)(__builtin_bitreverse32(k) >> (32 - nb)) : 0;
            ++k;
            if (i < nsteps) return i;
        }
        return nsteps;
    };
    auto stage_load = [&](int st, uint4 (&r)[4]) __attribute__((always_inline)) {
        const int t0 = tbeg + st * FSTEP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            r[q] = (t0 + idx / 128 < ntiles_all) ? pack16[(size_t)t0 * 128 + idx] : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage_store = [&](int b, const uint4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s_a[b][tid + 256 * q] = r[q];
    };
    auto filter_tile = [&](int i, uint4 a_hi, uint4 a_lo) __attribute__((always_inline)) {
        const f16x8 dh = __builtin_bit_cast(f16x8, a_hi), dl = __builtin_bit_cast(f16x8, a_lo);
        f32x16 fr = {0}, fi = {0};
        fr = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, brh, fr, 0, 0, 0);     // (smallest terms first)
        fi = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, bih, fi, 0, 0, 0);
        fr = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, brl, fr, 0, 0, 0);
        fi = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, bil, fi, 0, 0, 0);
        fr = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, brh, fr, 0, 0, 0);
        fi = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, bih, fi, 0, 0, 0);
        float m2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) m2[r] = __builtin_fmaf(fi[r], fi[r], fr[r] * fr[r]);
        auto max3 = [](float a, float b, float c) __attribute__((always_inline)) { return fmaxf(fmaxf(a, b), c); };     // (v_max3_f32: 8 instead of 10 instructions)
        const float tm = fmaxf(max3(max3(m2[0], m2[1], m2[2]), max3(m2[3], m2[4], m2[5]), m2[15]),
                               max3(max3(m2[6], m2[7], m2[8]), max3(m2[9], m2[10], m2[11]), max3(m2[12], m2[13], m2[14])));
        const bool tr = !(tm <= cut);
        if constexpr (SEED) { runa = fmaxf(runa, tm); return; }
        if (__builtin_amdgcn_ballot_w64(tr)) {                          // (uniform)
            if (tr) { runa = fmaxf(runa, tm); cut = runa - marg; }
            if (lane == 0) s_list[wave][cnt] = i;
            ++cnt;
        }
    };
    f32x4 av[NV], avn[NV];
    auto request = [&](int i, f32x4 (&dst)[NV]) __attribute__((always_inline)) {
        const f32x4* ap = (const f32x4*)(pack + ((size_t)(tbeg + i) * 64 + lane) * NPL);
#pragma unroll
        for (int v = 0; v < NV; ++v) dst[v] = ap[v];
    };
    auto exact_pass = [&]() __attribute__((always_inline)) {            // the listed tiles through the exact products
        if (cnt == 0) return;
        int i0 = __builtin_amdgcn_readfirstlane(s_list[wave][0]);
        request(i0, av);
        for (int l = 0; l < cnt; l += 2) {
            int i1 = 0;
            if (l + 1 < cnt) { i1 = __builtin_amdgcn_readfirstlane(s_list[wave][l + 1]); request(i1, avn); }
            exact_tile<NPAIR, NV>(tbeg + i0, h, av, bre, bim, I);
            if (l + 1 >= cnt) break;
            if (l + 2 < cnt) { i0 = __builtin_amdgcn_readfirstlane(s_list[wave][l + 2]); request(i0, av); }
            exact_tile<NPAIR, NV>(tbeg + i1, h, avn, bre, bim, I);
        }
        cnt = 0;
    };
    {
        // Tile 0 of the dictionary is listed whatever the filter says: for a pixel whose products are all zero MATLAB's max keeps atom 1,
        // and exact_tile() gives that answer when the dictionary's first tile is the first one it sees (its note on magnitude 0).
        if (!SEED && blockIdx.y == 0 && n > 0) { if (lane == 0) s_list[wave][0] = 0; cnt = 1; }
        unsigned k = 0;
        uint4 r[4];
        int st = (nsteps > 0) ? next_step(k) : nsteps, cur = 0;
        if (st < nsteps) { stage_load(st, r); stage_store(0, r); }
        __syncthreads();
        while (st < nsteps) {
            while (st < nsteps && cnt <= LCAP - FSTEP) {                // (room for one more step; every wave passes one barrier per step)
                const int stn = next_step(k);
                if (stn < nsteps) stage_load(stn, r);
                const int ntl = min(FSTEP, n - st * FSTEP);
                const uint4* ab = s_a[cur] + lane;
                uint4 fh = ab[0], fl = ab[64];                          // (the next tile's fragments are read before this tile's products)
                for (int tt = 0; tt < ntl; ++tt) {
                    const int tn = (tt + 1 < ntl) ? tt + 1 : tt;
                    const uint4 nh = ab[tn * 128], nl = ab[tn * 128 + 64];
                    filter_tile(st * FSTEP + tt, fh, fl);
                    fh = nh; fl = nl;
                }
                if (stn < nsteps) stage_store(cur ^ 1, r);
                __syncthreads();
                cur ^= 1; st = stn;
            }
            if constexpr (!SEED) exact_pass();
        }
    }
    if constexpr (SEED) {
        if (pub && runa >= 0.f) (void)__hip_atomic_fetch_max(gmax + p, __float_as_int(runa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // merge the two lane halves (same pixel, interleaved atom rows): larger value, then lower index
    {
        const float ob = __shfl(I.best, lane ^ 32, 64), ore = __shfl(I.cre, lane ^ 32, 64), oim = __shfl(I.cim, lane ^ 32, 64);
        const int oi = __shfl(I.bidx, lane ^ 32, 64);
        if (ob > I.best || (ob == I.best && oi < I.bidx)) { I.best = ob; I.bidx = oi; I.cre = ore; I.cim = oim; }
    }
    if (h == 0 && p < Npix) {
        int bidx = I.bidx;
        if (bidx >= K) bidx = 0;       // cannot happen: padded atoms are all-zero and never beat a real one
        if (part) {                    // atoms split over workgroups: k_dict_merge finishes the pixel
            part[(size_t)blockIdx.y * Npix + p] = make_float4(I.best, __int_as_float(bidx), I.cre, I.cim);
            return;
        }
        finish_pixel(p, Npix, K, I.best, bidx, I.cre, I.cim, normD, lut, Q, qmap, pd, mt, dm, win);
    }
}

// per pixel: the best candidate of the P atom parts (larger magnitude, then lower index), then the outputs as in k_dict_match
__global__ __launch_bounds__(256) void k_dict_merge(const float4* __restrict__ part, int P, int Npix, int K, const float* __restrict__ normD,
                                                     const float* __restrict__ lut, int Q, float* __restrict__ qmap, float* __restrict__ pd,
                                                     float* __restrict__ mt, int32_t* __restrict__ dm, float4* __restrict__ win) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Npix) return;
    float4 b = part[p];
    for (int k = 1; k < P; ++k) {
        const float4 o = part[(size_t)k * Npix + p];
        if (cand_better(o.x, __float_as_int(o.y), b.x, __float_as_int(b.y))) b = o;
    }
    finish_pixel(p, Npix, K, b.x, __float_as_int(b.y), b.z, b.w, normD, lut, Q, qmap, pd, mt, dm, win);
}

// Xfit(p, :) = ip(dm(p)) .* D(dm(p), :)  (mrf_dtm_cpu.m:95,129-134; single complex, interleaved, Npix x s column-major): the product of two
// singles rounded once, which is what MATLAB's double product of the two (exact) followed by single() gives.
__global__ __launch_bounds__(256) void k_dict_xfit(const float4* __restrict__ win, int Npix, int s, DictView dv, float2* __restrict__ xfit) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Npix) return;
    const float4 w = win[p];
    const int a = __float_as_int(w.z);
    for (int c = blockIdx.y; c < s; c += gridDim.y) {
        const float d = dict_atom(dv, a, c);
        xfit[(size_t)p + (size_t)Npix * c] = make_float2(w.x * d, w.y * d);
    }
}

}  // namespace

// grow-only device scratch of the dictionary match (the stream is drained before a buffer in use is replaced)
int dict_scratch(qmri_ctx* ctx, void** buf, size_t* cap, size_t need_bytes) {
    if (*cap >= need_bytes) return QMRI_OK;
    if (*buf) { QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream)); QMRI_HIP(ctx, hipFree(*buf)); *buf = nullptr; *cap = 0; }
    QMRI_HIP(ctx, hipMalloc(buf, need_bytes));
    *cap = need_bytes;
    return QMRI_OK;
}

int dict_launch_merge(qmri_ctx* ctx, const float4* part, int P, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win) {
    const DictHost& D = ctx->dict;
    k_dict_merge<<<dim3((Npix + 255) / 256), dim3(256), 0, ctx->stream>>>(part, P, Npix, D.K, D.d_normD, D.d_lut, D.Q, d_qmap, d_pd, d_mt, d_dm, win);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

static int dict_launch_narrow(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win);

// d_xfit (nullable): Npix x s complex single (mrf_dtm_cpu.m:95,129-134, par.f.Xout)
int dict_launch(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float2* d_xfit) {
    DictHost& D = ctx->dict;
    float4* win = nullptr;
    if (d_xfit) {
        QMRI_TRY(dict_scratch(ctx, (void**)&D.d_win, &D.win_cap, (size_t)Npix * sizeof(float4)));
        win = D.d_win;
    }
    if (D.wide) QMRI_TRY(dictw_launch(ctx, d_X, Npix, d_qmap, d_pd, d_mt, d_dm, win));
    else QMRI_TRY(dict_launch_narrow(ctx, d_X, Npix, d_qmap, d_pd, d_mt, d_dm, win));
    if (d_xfit) {
        const int npair = (D.s + 1) / 2;
        const DictView dv = {D.d_pack, D.wide, (npair <= 4) ? 4 : 8, D.G8};
        k_dict_xfit<<<dim3((Npix + 255) / 256, std::min(D.s, 64)), dim3(256), 0, ctx->stream>>>(win, Npix, D.s, dv, d_xfit);
        QMRI_HIP(ctx, hipGetLastError());
    }
    return QMRI_OK;
}

static int dict_launch_narrow(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win) {
    DictHost& D = ctx->dict;
    const int npair = (D.s + 1) / 2;
    if (npair < 1 || npair > MAXPAIR) {
        qmri_set_error(ctx, "dictionary match supports s <= %d channels (got %d)", 2 * MAXPAIR, D.s);
        return QMRI_ERR_UNSUPPORTED;
    }
    const bool filt = D.d_pack16 && D.filter_on;
    if (!D.slots) {
        int per_cu = 0, per_cu_f = 0;
        hipDeviceProp_t prop;
        QMRI_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        QMRI_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dict_match<5>, NT, 0));
        QMRI_HIP(ctx, (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, k_dict_match_f<5, false>, NT, 0)));
        D.slots = std::max(1, per_cu) * prop.multiProcessorCount;
        D.slots_f = std::max(1, per_cu_f) * prop.multiProcessorCount;
    }
    // atom parts: none if the pixel tiles fill the device's resident workgroups at least four times over (a ragged last round then
    // costs little), else as many as give ~8 rounds, each part keeping >= 64 atom tiles per wave
    const int ptiles = filt ? (Npix + 127) / 128 : (Npix + 31) / 32;
    const int slots = filt ? D.slots_f : D.slots;
    int P = 1, tper = D.ntiles;
    if (ptiles < 4 * slots) {
        P = (8 * slots + ptiles - 1) / ptiles;
        P = std::max(1, std::min(P, D.ntiles / (filt ? 64 : 4 * 64)));
    }
    if (filt) {                                                         // whole LDS steps per part, no empty part
        tper = ((D.ntiles + P - 1) / P + FSTEP - 1) / FSTEP * FSTEP;
        P = (D.ntiles + tper - 1) / tper;
    }
    float4* part = nullptr;
    if (P > 1) {
        QMRI_TRY(dict_scratch(ctx, (void**)&D.d_part, &D.part_cap, (size_t)P * Npix * sizeof(float4)));
        part = D.d_part;
    }
    dim3 grid(ptiles, P), blk(NT);
    const float mc = D.marg_coef * D.margin_scale;
    if (filt) {
        QMRI_TRY(dict_scratch(ctx, (void**)&D.d_gmax, &D.gmax_cap, (size_t)Npix * sizeof(int)));
        QMRI_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)D.d_gmax, (int)0xBF800000u /* -1.0f */, (size_t)Npix, ctx->stream));
    }
    // seed launch: ~12 steps of the whole dictionary per pixel, spread over as many workgroups as give one round of the device
    const int nsteps_all = (D.ntiles + FSTEP - 1) / FSTEP;
    const int seed_stride = std::max(1, nsteps_all / 12), nseed = (nsteps_all + seed_stride - 1) / seed_stride;
    const bool seed = filt && P > 1 && nsteps_all >= 48;
    const int Ps = seed ? std::max(1, std::min(nseed, (slots + ptiles - 1) / ptiles)) : 0;
#define LAUNCH(NP)                                                                                                              \
    do {                                                                                                                        \
        if (seed) k_dict_match_f<NP, true><<<dim3(ptiles, Ps), blk, 0, ctx->stream>>>(d_X, Npix, D.s, D.d_pack, D.ntiles, tper, D.K, D.d_normD, D.d_lut, \
                                                                    D.Q, nullptr, nullptr, nullptr, nullptr, nullptr, D.d_pack16, mc, D.d_gmax, seed_stride, nullptr); \
        if (filt) k_dict_match_f<NP, false><<<grid, blk, 0, ctx->stream>>>(d_X, Npix, D.s, D.d_pack, D.ntiles, tper, D.K, D.d_normD, D.d_lut, D.Q, \
                                                                    d_qmap, d_pd, d_mt, d_dm, part, D.d_pack16, mc, D.d_gmax, 0, win); \
        else k_dict_match<NP><<<grid, blk, 0, ctx->stream>>>(d_X, Npix, D.s, D.d_pack, D.ntiles, D.K, D.d_normD, D.d_lut, D.Q,  \
                                                             d_qmap, d_pd, d_mt, d_dm, part, win);                              \
    } while (0)
    switch (npair) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        case 4: LAUNCH(4); break;
        case 5: LAUNCH(5); break;
        case 6: LAUNCH(6); break;
        case 7: LAUNCH(7); break;
        default: LAUNCH(8); break;
    }
#undef LAUNCH
    QMRI_HIP(ctx, hipGetLastError());
    if (P > 1) QMRI_TRY(dict_launch_merge(ctx, part, P, Npix, d_qmap, d_pd, d_mt, d_dm, win));
    return QMRI_OK;
}

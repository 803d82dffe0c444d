// lsqr_kernels.hip -- the LSQR x-update of PnP-ADMM as two fused kernels per iteration (gfx950).
//
// Reference semantics (file:line relative to the reference root):
//   lsqr call  PnP_ADMM.m:102      x = lsqr(@afun, [y; sqrt(r) z], cg_tol, cg_iter, [], [], x0)
//   afun       PnP_ADMM.m:153-171  B = [A; sqrt(r) I],  A = F.forward, A' = F.adjoint (main_recon_tsmis_FFT.m:228-229)
//   (MathWorks lsqr restated from its documentation, as in oracle/orc_lsqr.c)
//
// One Golub-Kahan step needs A*v and A'*u, i.e. two 2-D transforms, and each 2-D transform needs one device-wide
// exchange (the transpose between the h pass and the w pass).  Two exchanges per iteration are therefore the minimum,
// and this file spends exactly two launches on them:
//
//   k_lsqr_w  (one block per k-row)   w-pass of the forward transform, gather-combine with V for the row's samples,
//                                     u(1:m) update, scatter-combine of the new u back onto the row, inverse w-pass.
//                                     The block also owns an s*M slice of the image-domain vectors: v = v/alpha and
//                                     u(m+1:end) update there.
//   k_lsqr_h  (one block per L lines) scalar recurrences + stopping tests, inverse h-pass, d / x / v updates,
//                                     forward h-pass of the new (not yet normalised) v.
//
// Every global operand of a block (its lines / row, its slice of the vectors, the row's samples, V, the twiddles, the
// partial sums) is requested up front, so a block pays roughly one memory latency, not one per phase.
//
// The transforms are linear, so the 1/alpha and 1/beta normalisations that LSQR applies before A and A' are applied
// after them instead (alpha = ||v|| and beta = ||u|| are device-wide sums that only the *next* kernel can know).
// Norms are per-block partial sums reduced in a fixed order by every consumer block: results are run-to-run
// reproducible and every block takes the same convergence decision without a host round trip.
#include <algorithm>
#include <cstdlib>
#include "dc_device.h"

using namespace dcdev;

namespace {

constexpr int NTW = 512;         // threads of k_lsqr_w
constexpr int NE_PRE = 5;        // samples per thread of k_lsqr_w requested up front (5 * 512 = 2560 >= the busiest spiral row)
constexpr int SL = 8;            // lanes sharing one scatter group (<= DC_GCAP samples of one k location, all channels)
constexpr int LH = 8;            // lines per k_lsqr_h block

// phase stamps of the last launch (slice 0), 100 MHz constant clock; costs one scalar branch when disabled
#define LSQR_STAMP(KID, k)                                                                       \
    do {                                                                                         \
        if (ls.stamps && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 512)                \
            ls.stamps[((KID)*512 + blockIdx.x) * 16 + (k)] = wall_clock64();                      \
    } while (0)

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
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    return a;
}

// block totals of two per-thread values (contains its own barriers)
template <int NTHR>
__device__ __forceinline__ void block_sum2_t(double& a, double& b, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds_barrier();
    if (lane == 0) { sh[wid] = a; sh[NTHR / 64 + wid] = b; }
    lds_barrier();
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int i = 0; i < NTHR / 64; ++i) { ra += sh[i]; rb += sh[NTHR / 64 + i]; }
    a = ra; b = rb;
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

// u(m+1:end) update, shared by both kernels so that they produce the same bits:  sqrt(r) v - alpha (u / beta_prev)
__device__ __forceinline__ double2 ub_update(double2 v, double2 ub, double sr, double alpha, double inv_bprev) {
    return make_double2(__fma_rn(v.x, sr, -(alpha * (ub.x * inv_bprev))), __fma_rn(v.y, sr, -(alpha * (ub.y * inv_bprev))));
}

// dynamic LDS of k_lsqr_w, in bytes, for a staging capacity of ucap samples
template <int R1, int R2> constexpr size_t lsqr_w_lds(int ucap, int vcap, int gcap) {
    return (size_t)DC_MAXS * Plan<R1, R2>::LINE * 16 + (size_t)ucap * 16 + (size_t)Plan<R1, R2>::N * 16 +
           (size_t)gcap * DC_MAXS * 16 + (size_t)vcap * 8 + 2 * (NTW / 64) * 8 + ((Plan<R1, R2>::N + 4) & ~3) * 4 + (size_t)ucap * 2;
}
constexpr int NG_PRE = 2;        // scatter groups per 8 lanes requested up front (2 * 512 / 8 = 128 groups)
constexpr int NV_PRE = 4;        // V values per thread requested up front (4 * 512 = 2048 >= T*s of the reference: 200 * 10)

// -----------------------------------------------------------------------------------------------------------------
// k_lsqr_w
//   INIT : tmp holds the h-pass of x0.   u(1:m) = y - A x0 ; u(m+1:end) = sqrt(r) z - sqrt(r) x0   (b - B*x0)
//   ITER : tmp holds the h-pass of the un-normalised v.   v = v/alpha ; u = B v - alpha (u / beta_prev)
//   both : tmp(row kh) <- conj-domain inverse w-pass of  sum_t V(t,c) u(t,k)   (A'u before the 1/beta scale)
// partial sums: pu[ii&1][kh] = |u(m+1:end) slice|^2 ,  pu[ii&1][N + kh] = |u(1:m) row|^2
// -----------------------------------------------------------------------------------------------------------------
template <int R1, int R2, bool INIT>
__global__ __launch_bounds__(NTW) void k_lsqr_w(OpDev op, LsqrDev ls, const double2* __restrict__ x0,
                                                 const double2* __restrict__ z, double2* tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N;
    extern __shared__ __align__(16) unsigned char smem[];
    const int ucap = ls.ucap, vcap = ls.vcap;
    cd* lds = (cd*)smem;                                   // s channel lines of the row
    double2* ulds = (double2*)(lds + DC_MAXS * P::LINE);   // the row's new u(1:m), k-sorted
    cd* twl = (cd*)(ulds + ucap);                          // twiddles
    cd* part = twl + N;                                    // per scatter group: sum over its samples, all channels
    double* vlds = (double*)(part + ls.gcap * DC_MAXS);    // V(t, c)
    double* red = vlds + vcap;
    int* gl = (int*)(red + 2 * (NTW / 64));                // first scatter group of each kw of the row
    unsigned short* tlds = (unsigned short*)(gl + ((N + 4) & ~3));   // frame of each staged sample
    constexpr int M = N;                                   // square grids only (checked by qmri_set_operator)
    const int tid = threadIdx.x, b = blockIdx.y, kh = blockIdx.x, s = op.s, sM = s * M;
    const size_t n = (size_t)s * N * M;
    const size_t mb = (size_t)b * op.m;
    const size_t g0 = (size_t)b * n + (size_t)kh * sM;     // image-domain slice owned by this block
    LSQR_STAMP(0, 0);

    // ---- request everything
    const int done = INIT ? 0 : ls.st[b].done;
    const int r0 = op.kptr[kh * M], r1 = op.kptr[(kh + 1) * M];
    constexpr int NQR = (N * DC_MAXS + NTW - 1) / NTW;
    double2 rrow[NQR], ra[NQR], rb[NQR];
#pragma unroll
    for (int q = 0; q < NQR; ++q) {                        // (clamped, not predicated: the loads stay branch-free and in flight)
        const int i = (tid + NTW * q < sM) ? tid + NTW * q : sM - 1;
        const int c = i / M, w = i - c * M;
        rrow[q] = tmp[(size_t)b * n + ((size_t)c * N + kh) * M + w];
        ra[q] = INIT ? x0[g0 + i] : ls.v[g0 + i];
        rb[q] = INIT ? z[g0 + i] : ls.ub[g0 + i];
    }
    double rv[NV_PRE];                                     // V(t, c) is read once per sample and channel: keep it on chip
#pragma unroll
    for (int q = 0; q < NV_PRE; ++q) { const int i = tid + NTW * q; rv[q] = op.Vt[(i < op.T * s) ? i : 0]; }
    const cd rtw = op.tw[(tid < N) ? tid : 0];
    const int rk = op.gkw[kh * (M + 1) + ((tid <= M) ? tid : 0)];
    const int gbase = op.gptr[kh], ng = op.gptr[kh + 1] - gbase;
    KGroup rg[NG_PRE];
#pragma unroll
    for (int q = 0; q < NG_PRE; ++q) { const int g = (tid + NTW * q) / SL; rg[q] = op.grp[(g < ng) ? gbase + g : 0]; }
    const int hi0 = (r0 + ucap < r1) ? r0 + ucap : r1;     // end of the first (normally the only) chunk of samples
    KEntry ren[NE_PRE];
    double2 rut[NE_PRE];
#pragma unroll
    for (int q = 0; q < NE_PRE; ++q) {
        const int e = (r0 + tid + NTW * q < hi0) ? r0 + tid + NTW * q : ((hi0 > r0) ? hi0 - 1 : 0);
        ren[q] = op.ent[e];
        rut[q] = INIT ? ls.yk[mb + e] : ls.ut[mb + e];
    }
    if (done) return;
    if (!INIT && tid < 64) {
        const double pa = wave_sum(ls.pv[(ls.ii - 1) & 1] + (size_t)b * ls.nblk_h, ls.nblk_h);
        const double pb = wave_sum(ls.pu[(ls.ii - 1) & 1] + (size_t)b * ls.npu, ls.npu);
        if (tid == 0) { red[0] = pa; red[1] = pb; }
    }

    // ---- operands to LDS, scalars
#pragma unroll
    for (int q = 0; q < NQR; ++q) {
        const int i = tid + NTW * q;
        if (i < sM) { const int c = i / M, w = i - c * M; lds[c * P::LINE + w] = rrow[q]; }
    }
#pragma unroll
    for (int q = 0; q < NV_PRE; ++q) { const int i = tid + NTW * q; if (i < vcap) vlds[i] = rv[q]; }
    for (int i = tid + NTW * NV_PRE; i < op.T * s; i += NTW) vlds[i] = op.Vt[i];
    if (tid < N) twl[tid] = rtw;
    if (tid <= M) gl[tid] = rk;
    double alpha = 0.0, inv_alpha = 1.0, inv_bprev = 1.0;
    const double sr = ls.sr;
    lds_barrier();
    if (!INIT) {
        alpha = sqrt(red[0]);
        inv_alpha = 1.0 / alpha;
        inv_bprev = 1.0 / sqrt(red[1]);
    }
    LSQR_STAMP(0, 1);
    // ---- image-domain slice
    double acc_b = 0.0;
#pragma unroll
    for (int q = 0; q < NQR; ++q) {
        const int i = tid + NTW * q;
        if (i < sM) {
            double2 ub;
            if (INIT) {
                ub = make_double2(rb[q].x * sr - ra[q].x * sr, rb[q].y * sr - ra[q].y * sr);
                ls.ub[g0 + i] = ub;
            } else {                                       // only its norm is needed here: k_lsqr_h recomputes and stores it
                ub = ub_update(make_double2(ra[q].x * inv_alpha, ra[q].y * inv_alpha), rb[q], sr, alpha, inv_bprev);
            }
            acc_b += ub.x * ub.x + ub.y * ub.y;
        }
    }
    LSQR_STAMP(0, 2);
    // ---- forward w-pass of the row, back to natural order in LDS
    cd out[R2];
    int line2, k1;
    const bool act = fft_lds<R1, R2, false>(lds, s, twl, out, line2, k1);
    lds_barrier();
    if (act) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) lds[line2 * P::LINE + k1 + R1 * k2] = out[k2];
    }
    lds_barrier();
    LSQR_STAMP(0, 3);
    const double sc = 1.0 / sqrt((double)N * (double)M);

    // ---- gather-combine + u(1:m) update, one thread per sample
    const int nch = (r1 - r0 + ucap - 1) / ucap;           // chunks of ucap samples: 1 unless the row outgrows the staging area
    double acc_t = 0.0;
    auto sample = [&](int e, const KEntry en, const double2 uold) {
        double re = 0.0, im = 0.0;
        for (int c = 0; c < s; ++c) {
            const double v = vlds[en.t * s + c];
            const cd X = lds[c * P::LINE + en.kw];
            re += v * X.x;
            im += v * X.y;
        }
        re *= sc; im *= sc;
        double2 u;
        if (INIT) {
            u = make_double2(uold.x - re, uold.y - im);                   // y - A x0
        } else {
            u.x = re * inv_alpha - alpha * (uold.x * inv_bprev);           // A v - alpha (u / beta_prev)
            u.y = im * inv_alpha - alpha * (uold.y * inv_bprev);
        }
        ls.ut[mb + e] = u;
        acc_t += u.x * u.x + u.y * u.y;
        if (nch == 1) { ulds[e - r0] = u; tlds[e - r0] = en.t; }
    };
#pragma unroll
    for (int q = 0; q < NE_PRE; ++q) {
        const int e = r0 + tid + NTW * q;
        if (e < hi0) sample(e, ren[q], rut[q]);
    }
    for (int e = r0 + tid + NTW * NE_PRE; e < r1; e += NTW) sample(e, op.ent[e], INIT ? ls.yk[mb + e] : ls.ut[mb + e]);
    if (nch > 1) {                                         // prefetched samples only cover [r0, hi0)
        for (int e = hi0 + tid; e < r0 + NTW * NE_PRE && e < r1; e += NTW) sample(e, op.ent[e], INIT ? ls.yk[mb + e] : ls.ut[mb + e]);
    }
    LSQR_STAMP(0, 4);

    // ---- scatter-combine.  SL lanes per group (<= DC_GCAP samples of one k location) add their samples for all channels
    // from LDS and combine in a fixed tree; a k location's groups are then added in order.  Atomics-free, fixed order.
    auto group_sum = [&](const KGroup g, int sub, int lo_rel, int hi_rel, cd* dst, bool first) {
        double xr[DC_MAXS], xi[DC_MAXS];
#pragma unroll
        for (int c = 0; c < DC_MAXS; ++c) { xr[c] = 0.0; xi[c] = 0.0; }
        double2 u[DC_GCAP / SL];
        int t[DC_GCAP / SL];
#pragma unroll
        for (int j = 0; j < DC_GCAP / SL; ++j) {           // all of the lane's samples in flight at once
            const int e = g.b + sub + SL * j;
            const bool ok = e < g.e && e >= lo_rel && e < hi_rel;
            const int ec = ok ? e - lo_rel : 0;
            u[j] = ulds[ec]; t[j] = tlds[ec];
            if (!ok) u[j] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < DC_GCAP / SL; ++j) {
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) {
                const double v = vlds[t[j] * s + c];       // (c >= s reads a neighbour: finite garbage into unused sums)
                xr[c] += v * u[j].x; xi[c] += v * u[j].y;
            }
        }
#pragma unroll
        for (int c = 0; c < DC_MAXS; ++c) { xr[c] = group8_sum(xr[c]); xi[c] = group8_sum(xi[c]); }
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) {
                if (first) dst[c] = mk(xr[c], xi[c]);
                else { const cd o = dst[c]; dst[c] = mk(o.x + xr[c], o.y + xi[c]); }
            }
        }
    };
    for (int ch = 0; ch < nch; ++ch) {
        const int lo = r0 + ch * ucap, hi = (lo + ucap < r1) ? lo + ucap : r1;
        if (nch > 1) {
            __syncthreads();                               // (full barrier: u(1:m) is re-read from global memory here)
            for (int i = tid; i < hi - lo; i += NTW) { ulds[i] = ls.ut[mb + lo + i]; tlds[i] = op.ent[lo + i].t; }
        }
        lds_barrier();                                     // staged samples visible
        LSQR_STAMP(0, 8);
#pragma unroll
        for (int q = 0; q < NG_PRE; ++q) {
            const int g = (tid + NTW * q) / SL;
            if (g < ng) group_sum(rg[q], tid & (SL - 1), lo - r0, hi - r0, part + g * DC_MAXS, ch == 0);
            LSQR_STAMP(0, 9 + q);
        }
        for (int g = (tid + NTW * NG_PRE) / SL; g < ng; g += NTW / SL)
            group_sum(op.grp[gbase + g], tid & (SL - 1), lo - r0, hi - r0, part + g * DC_MAXS, ch == 0);
    }
    lds_barrier();                                         // group sums complete; every gather read of the lines is done
    for (int i = tid; i < sM; i += NTW) {
        const int kw = i / s, c = i - kw * s;
        double xr = 0.0, xi = 0.0;
        if (nch > 0) for (int g = gl[kw]; g < gl[kw + 1]; ++g) { const cd p = part[g * DC_MAXS + c]; xr += p.x; xi += p.y; }
        lds[c * P::LINE + kw] = mk(xr, -xi);               // conjugate: inverse transform by conj-FFT-conj
    }
    LSQR_STAMP(0, 5);
    block_sum2_t<NTW>(acc_b, acc_t, red);
    if (tid == 0) {
        double* pu = ls.pu[ls.ii & 1] + (size_t)b * ls.npu;
        pu[kh] = acc_b;
        pu[N + kh] = acc_t;
    }
    if (fft_lds<R1, R2, false>(lds, s, twl, out, line2, k1)) {
        double2* dst = tmp + (size_t)b * n + ((size_t)line2 * N + kh) * M;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) dst[k1 + R1 * k2] = out[k2];
    }
    LSQR_STAMP(0, 6);
}

// -----------------------------------------------------------------------------------------------------------------
// k_lsqr_h : lines (c, w0..w0+LH) of tmp, all kh.
//   scalars (every block, identical bits) -> inverse h-pass -> A'u = ifft2(.)*sqrt(NM) / beta
//   INIT : v = B'u ; d = 0                                   ITER : d = (v - thet d)/rho ; x += phi d ; v = B'u - beta v
//   then the forward h-pass of the new v back into the same lines of tmp, and pv[ii&1][block] = |v lines|^2
// -----------------------------------------------------------------------------------------------------------------
template <int R1, int R2, bool INIT>
__global__ __launch_bounds__(NT, 2) void k_lsqr_h(OpDev op, LsqrDev ls, double2* tmp, double2* __restrict__ xio) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = LH;
    __shared__ cd lds[L * P::LINE];
    __shared__ cd twl[N];
    __shared__ double red[2 * NT / 64];
    constexpr int M = N;
    const int tid = threadIdx.x, b = blockIdx.y;
    const size_t n = (size_t)op.s * N * M;
    LsqrState* st = ls.st + b;
    LSQR_STAMP(1, 0);
    const int l0 = blockIdx.x * L;
    const int c = l0 / M, w0 = l0 - c * M;
    double2* lines = tmp + (size_t)b * n + (size_t)c * N * M + w0;
    const size_t gb = (size_t)b * n + (size_t)l0 * N;      // the same lines in the image domain: L*N contiguous elements

    // ---- request everything
    const int done = INIT ? 0 : st->done;
    constexpr int NQ = (L * N + NT - 1) / NT;
    double2 rl[NQ], rub[NQ], rvv[NQ], rd[NQ], rx[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {                         // (clamped, not predicated: the loads stay branch-free and in flight)
        const int i = (tid + NT * q < L * N) ? tid + NT * q : L * N - 1;
        const int kh = i / L, line = i - kh * L;
        rl[q] = lines[(size_t)kh * M + line];
        rub[q] = ls.ub[gb + i];
        if (!INIT) { rvv[q] = ls.v[gb + i]; rd[q] = ls.d[gb + i]; rx[q] = xio[gb + i]; }
    }
    const cd rtw = op.tw[(tid < N) ? tid : 0];
    if (done) return;
    if (tid < 64) {
        double pa, pb;
        if (INIT) {
            pa = wave_sum(ls.pu[0] + (size_t)b * ls.npu, ls.npu);
            pb = wave_sum(ls.pz + (size_t)b * ls.nblk_z, ls.nblk_z);
        } else {
            pa = wave_sum(ls.pv[(ls.ii - 1) & 1] + (size_t)b * ls.nblk_h, ls.nblk_h);
            pb = wave_sum(ls.pu[ls.ii & 1] + (size_t)b * ls.npu, ls.npu);
        }
        if (tid == 0) { red[0] = pa; red[1] = pb; }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + NT * q;
        if (i < L * N) { const int kh = i / L, line = i - kh * L; lds[line * P::LINE + kh] = rl[q]; }
    }
    if (tid < N) twl[tid] = rtw;
    lds_barrier();
    const double pa = red[0], pb = red[1];

    double inv_beta = 1.0, beta = 0.0, thet = 0.0, inv_rho = 1.0, phi = 0.0, alpha = 0.0, inv_alpha = 1.0, inv_bprev = 1.0;
    const double sr = ls.sr;
    const bool writer = blockIdx.x == 0 && tid == 0;
    if (INIT) {
        const double beta0 = sqrt(pa);
        const double n2b = sqrt(st->ny2 + sr * sr * pb);
        const bool fin = (beta0 == 0.0 || n2b == 0.0);       // x0 already exact, or b = 0
        if (writer) {
            LsqrScalars S;
            S.c = 1.0; S.s = 0.0; S.phibar = beta0; S.normr = beta0; S.norma = 0.0; S.factor = beta0;
            S.thet = 0.0; S.rho = 1.0; S.phi = 0.0; S.beta = beta0; S.alpha = 0.0;
            st->sc[0] = S;
            st->n2b = n2b;
            st->tolb = ls.tol * n2b;
            st->iter = fin ? 0 : ls.maxit;
            st->flag = fin ? 0 : 1;
            st->done = fin ? 1 : 0;
        }
        if (fin) return;
        inv_beta = 1.0 / beta0;
    } else {
        alpha = sqrt(pa);
        beta = sqrt(pb);
        const LsqrScalars O = st->sc[(ls.ii - 1) & 1];
        inv_alpha = 1.0 / alpha;
        inv_bprev = 1.0 / O.beta;                            // same bits as k_lsqr_w's: both come from wave_sum of pu[(ii-1)&1]
        LsqrScalars S;
        const double normar = alpha * O.factor;
        S.norma = sqrt(O.norma * O.norma + alpha * alpha + beta * beta);
        S.thet = -O.s * alpha;
        const double rhot = O.c * alpha;
        S.rho = sqrt(rhot * rhot + beta * beta);
        S.c = rhot / S.rho;
        S.s = -beta / S.rho;
        S.phi = S.c * O.phibar;
        S.phibar = S.s * O.phibar;
        S.beta = beta; S.alpha = alpha;
        bool conv = false;
        if (normar == 0.0) conv = true;                                   // all-zero correction
        if (normar / (S.norma * O.normr) <= ls.tol) conv = true;         // min ||b - Bx|| test
        if (O.normr <= st->tolb) conv = true;                             // Bx = b test
        S.normr = fabs(S.s) * O.normr;
        S.factor = fabs(S.s * S.phi);
        if (writer) {
            st->sc[ls.ii & 1] = S;
            if (conv) { st->done = 1; st->flag = 0; st->iter = ls.ii - 1; }
        }
        if (conv) return;
        inv_beta = 1.0 / beta; thet = S.thet; inv_rho = 1.0 / S.rho; phi = S.phi;
    }
    LSQR_STAMP(1, 1);
    cd out[R2];
    int line2, k1;
    const bool act = fft_lds<R1, R2, false>(lds, L, twl, out, line2, k1);
    const double sc = 1.0 / sqrt((double)N * (double)M);
    lds_barrier();                                        // every step-2 read of the lines is done
    if (act) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) lds[line2 * P::LINE + k1 + R1 * k2] = mk(out[k2].x * sc, -out[k2].y * sc);
    }
    lds_barrier();
    LSQR_STAMP(1, 2);
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + NT * q;
        if (i < L * N) {
            const int line = i / N, h = i - line * N;
            const cd a = lds[line * P::LINE + h];
            const double vx = a.x * inv_beta, vy = a.y * inv_beta;
            double2 ub = rub[q];
            double2 vr;
            if (INIT) {
                // v = B'*u = A'*u(1:m) + sqrt(r) u(m+1:end)       PnP_ADMM.m:164-167
                vr = make_double2(vx + (ub.x * inv_beta) * sr, vy + (ub.y * inv_beta) * sr);
                ls.d[gb + i] = make_double2(0.0, 0.0);
            } else {
                const double2 vh = make_double2(rvv[q].x * inv_alpha, rvv[q].y * inv_alpha);   // v = v/alpha
                ub = ub_update(vh, ub, sr, alpha, inv_bprev);                                     // u(m+1:end), as normed in k_lsqr_w
                ls.ub[gb + i] = ub;
                double2 dd = rd[q];
                dd.x = (vh.x - thet * dd.x) * inv_rho;            // d = (v - thet d)/rho
                dd.y = (vh.y - thet * dd.y) * inv_rho;
                ls.d[gb + i] = dd;
                double2 xv = rx[q];
                xv.x += phi * dd.x; xv.y += phi * dd.y;           // x = x + phi d
                xio[gb + i] = xv;
                vr = make_double2((vx + (ub.x * inv_beta) * sr) - beta * vh.x,
                                  (vy + (ub.y * inv_beta) * sr) - beta * vh.y);   // v = B'u - beta v
            }
            ls.v[gb + i] = vr;
            acc += vr.x * vr.x + vr.y * vr.y;
            lds[line * P::LINE + h] = vr;
        }
    }
    LSQR_STAMP(1, 3);
    const double tot = block_sum(acc, red);
    if (tid == 0) ls.pv[ls.ii & 1][(size_t)b * ls.nblk_h + blockIdx.x] = tot;
    if (fft_lds<R1, R2, true>(lds, L, twl, out, line2, k1)) {
        double2* dst = lines + line2;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) dst[(size_t)(k1 + R1 * k2) * M] = out[k2];
    }
    LSQR_STAMP(1, 4);
}

template <int R1, int R2>
int launch_lsqr_t(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, bool init, int B, const double2* x0,
                  const double2* z, double2* tmp, double2* xio) {
    dim3 gh(op.s * op.M / LH, B), gw(op.N, B);
    hipStream_t st = ctx->stream;
    const size_t wlds = lsqr_w_lds<R1, R2>(ls.ucap, ls.vcap, ls.gcap);
    bool& attr_set = ctx->lsqr_lds_attr[R1 == 16 ? (R2 == 14 ? 0 : 1) : (R2 == 8 ? 2 : 3)];   // > 64 KB of dynamic LDS must be allowed once
    if (!attr_set) {
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_lsqr_w<R1, R2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        QMRI_HIP(ctx, hipFuncSetAttribute((const void*)k_lsqr_w<R1, R2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    if (init) {
        k_lsqr_w<R1, R2, true><<<gw, dim3(NTW), wlds, st>>>(op, ls, x0, z, tmp);
        k_lsqr_h<R1, R2, true><<<gh, dim3(NT), 0, st>>>(op, ls, tmp, xio);
    } else {
        k_lsqr_w<R1, R2, false><<<gw, dim3(NTW), wlds, st>>>(op, ls, nullptr, nullptr, tmp);
        k_lsqr_h<R1, R2, false><<<gh, dim3(NT), 0, st>>>(op, ls, tmp, xio);
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

}  // namespace

int dc_lsqr_nblk_h(int M, int s) { return s * M / LH; }

// LDS plan of k_lsqr_w: V (T*s doubles, padded) always lives in LDS; the staging area takes the busiest k-row if it
// fits in what is left of the 160 KB, else rows are processed in chunks.  Returns false if V itself cannot fit.
bool dc_lsqr_plan(int N, int T, int s, int maxrow, int maxgroups, int* ucap_out, int* vcap_out) {
    const int vcap = ((T * s + DC_MAXS + 15) / 16) * 16;
    int ucap = ((std::max(maxrow, 1) + NTW - 1) / NTW) * NTW;
    if (const char* e = getenv("QMRI_LSQR_UCAP")) { const int v = atoi(e); if (v >= NTW && v % NTW == 0) ucap = std::min(ucap, v); }   // test hook
    const size_t fixed = (size_t)DC_MAXS * (N + 32) * 16 + (size_t)N * 16 + (size_t)maxgroups * DC_MAXS * 16 + (size_t)vcap * 8 + 128 + (size_t)(N + 4) * 4;
    const size_t budget = 160 * 1024 - 1024;
    if (fixed + (size_t)NTW * 18 > budget) return false;
    while ((size_t)ucap * 18 + fixed > budget) ucap -= NTW;
    *ucap_out = ucap; *vcap_out = vcap;
    return true;
}

// tmp must hold the forward h-pass of x0 (dc_launch_fwd with DC_FWD_H_ONLY) when init is true.
int dc_launch_lsqr(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, bool init, int B, const double2* x0,
                   const double2* z, double2* tmp, double2* xio) {
    switch (op.N) {
        case 224: return launch_lsqr_t<16, 14>(ctx, op, ls, init, B, x0, z, tmp, xio);
        case 128: return launch_lsqr_t<16, 8>(ctx, op, ls, init, B, x0, z, tmp, xio);
        case 64: return launch_lsqr_t<8, 8>(ctx, op, ls, init, B, x0, z, tmp, xio);
        case 32: return launch_lsqr_t<8, 4>(ctx, op, ls, init, B, x0, z, tmp, xio);
        default:
            qmri_set_error(ctx, "unsupported grid size N=%d (supported: 32, 64, 128, 224)", op.N);
            return QMRI_ERR_UNSUPPORTED;
    }
}

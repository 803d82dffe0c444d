// kslsqr_kernels.hip -- the LSQR x-update of PnP-ADMM, iterated in k-space on the sampled locations only (gfx950).
//
// Reference semantics (file:line relative to the reference root):
//   lsqr call  PnP_ADMM.m:102      x = lsqr(@afun, [y; sqrt(r) z], cg_tol, cg_iter, [], [], x0)
//   afun       PnP_ADMM.m:153-171  B = [A; sqrt(r) I],  A = F.forward, A' = F.adjoint (main_recon_tsmis_FFT.m:228-229)
//   (MathWorks lsqr restated from its documentation, as in oracle/orc_lsqr.c)
//
// A = P*U with U the unitary 2-D DFT (fft2/sqrt(NM)) and P the sample-and-combine-with-V operator, which only
// couples the s channels of ONE k location.  Substituting xhat = U x turns B into [P; sqrt(r) I]: the same Krylov
// iteration (same alpha, beta, rho, phi, same stopping tests, iterates related by the unitary U, i.e. equal up to
// rounding) but without a single transform inside the loop.  Two further facts shrink it:
//   * every k location is independent apart from the two norms per iteration (alpha = ||v||, beta = ||u||);
//   * on a never-sampled k the operator is the scalar sqrt(r), so v, u(m+1:end), d and x - x0 stay multiples of
//     zhat - xhat0 there: four scalars (ua, ub, uc, ue) and one number R = sum |zhat - xhat0|^2 carry 78 % of
//     k-space (spiral mask) exactly.
// The vectors of the iteration therefore live on the sampled locations ("slots"): ns*s complex numbers (1.8 MB at
// 224 x 224 x 10 with the spiral mask), which stay in L2 for the whole solve.
//
//   k_ks_init_a  (block per k-row)  residual b - B*x0 in k-space: u(1:m) = y - P xhat0, u(m+1:end) = sqrt(r)(zhat - xhat0)
//   k_ks_b<INIT> (block per work unit) beta0, v = B'u / beta0, d = 0
//   k_ks_a       u = B v - alpha u          (needs alpha -> after k_ks_b)
//   k_ks_b       scalars + stopping tests, d, x, v = B'u - beta v   (needs beta -> after k_ks_a)
//   k_ks_final_w (block per k-row)  xhat = compact x on sampled k, xhat0 + ue (zhat - xhat0) elsewhere; ||y - P xhat||^2;
//                                   inverse w-pass (the h-pass is k_adj_h of dc_kernels.hip)
// Two launches per LSQR iteration; the two transforms per x-update (zhat in, x out) are outside the loop.
// Norms are per-block partial sums combined by wave_sum (one fixed order for every consumer): run-to-run reproducible,
// and every block takes the same convergence decision without a host round trip.
#include <algorithm>
#include <cstdlib>
#include "dc_device.h"
#include <hip/hip_ext.h>

using namespace dcdev;

namespace {

constexpr int KT = 256;          // threads of every kernel in this file
// Capacities of a work unit (round 5): the iteration kernels are instantiated for three shapes of unit, chosen when the operator is planned
// (api_core.cpp, KS_CAPS in qmri_internal.h) so that a single slice's units fit the chip at once and the one-launch iteration applies:
//   0  64 slots x 1024 samples   the spiral masks of cut1 ... cut3 (11 051 sampled k, ~11 samples each at T = 200): ~250 units
//   1  256 slots x  768 samples  masks that sample EVERY k a few times (EPI: 50 176 k x 2.7 samples): 196 - 250 units instead of 784
//   2  64 slots x 2560 samples   few k, many samples each (spiral cut0, T = 1000: 56 per k): ~248 units instead of 604
template <int ID_, int SCAP_, int ECAP_, int GCAP_, int SL_> struct KsCaps {
    static constexpr int ID = ID_, SCAP = SCAP_, ECAP = ECAP_, GCAP = GCAP_, SL = SL_, GCAPB = ECAP_ / GCAP_ + SCAP_;   // GCAP samples per scatter group, SL lanes share one
    static constexpr int NEQ = (SCAP_ * DC_MAXS + KT - 1) / KT;    // (slot, channel) elements per thread
    static constexpr int NSQ = (ECAP_ + KT - 1) / KT;              // samples per thread
    static constexpr int NGQ = (GCAPB * SL_ + KT - 1) / KT;        // scatter groups per SL lanes
    static constexpr int MINW = (ID_ == 0 || ID_ == 3) ? 2 : 1;    // workgroups per CU the register budget is planned for
};
typedef KsCaps<0, 64, 1024, 32, 8> Caps0;
typedef KsCaps<1, 256, 768, 4, 1> Caps1;
typedef KsCaps<2, 64, 2560, 32, 8> Caps2;
typedef KsCaps<3, 64, 256, 4, 1> Caps3;
template <class CP> constexpr bool caps_ok() {
    return CP::SCAP == KS_CAPS[CP::ID].scap && CP::ECAP == KS_CAPS[CP::ID].ecap && CP::GCAP == KS_CAPS[CP::ID].gcap && CP::SL == KS_CAPS[CP::ID].sl && CP::GCAPB == KS_CAPS[CP::ID].gcapb;
}
static_assert(caps_ok<Caps0>() && caps_ok<Caps1>() && caps_ok<Caps2>() && caps_ok<Caps3>(), "unit shapes: host table (qmri_internal.h KS_CAPS)");
// sum over the SL lanes that share a scatter group (SL = 8: fixed DPP tree; SL = 1: the lane's own sum)
template <int SL> __device__ __forceinline__ double group_sum(double v) { if constexpr (SL == 8) return group8_sum(v); else return v; }
constexpr int NVQ = 8;                                     // V values per thread requested up front (8 * 256 = 2048)
constexpr int KS_GRAN_MAXG = 320;                          // k_ks_persist: most work units per slice (its all-reduce holds 2G granules in 64 x 10 registers)

#define KS_STAMP(KID, k)                                                                         \
    do {                                                                                         \
        if (ks.stamps && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 512)                \
            ks.stamps[((KID)*512 + blockIdx.x) * 16 + (k)] = wall_clock64();                     \
    } while (0)

__device__ __forceinline__ void load_v(const OpDev& op, double* rv) {
#pragma unroll
    for (int q = 0; q < NVQ; ++q) { const int i = threadIdx.x + KT * q; rv[q] = op.Vt[(i < op.T * op.s) ? i : 0]; }
}
__device__ __forceinline__ void store_v(const OpDev& op, const double* rv, double* vlds) {
#pragma unroll
    for (int q = 0; q < NVQ; ++q) { const int i = threadIdx.x + KT * q; if (i < op.T * op.s) vlds[i] = rv[q]; }
    for (int i = threadIdx.x + KT * NVQ; i < op.T * op.s; i += KT) vlds[i] = op.Vt[i];
}

// ---------------------------------------------------------------------------------------------------------------
// k_ks_init_a : one block per k-row kh.  xhat0 / zhat rows -> compact x, u(m+1:end) on the sampled k; R on the rest;
// u(1:m) = y - P xhat0 for the row's samples.
// FWDW: the row of zhat is not read but MADE here -- the w-pass of z's transform (k_fwd_w<DC_SPECTRUM>: FFT along w of the row's s channel
// lines of tmp, the h-pass output) runs in the same launch, the row goes to registers and to ks.zhat (k_ks_final_w reads it): one launch and
// one round trip of zhat less per x-update.  The FFT buffer and the xhat0 lines share one piece of LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int R1, int R2, bool FWDW>
__global__ __launch_bounds__(KT) void k_ks_init_a(OpDev op, KsDev ks, const double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int M = N;                                   // square grids only (checked by qmri_set_operator)
    const int tid = threadIdx.x, kh = blockIdx.x, b = blockIdx.y, s = op.s, sM = s * M;
    cd* lines = (cd*)smem;                                 // [c][kw] xhat0 of the row  (FWDW: first the FFT buffer, [c][P::LINE])
    double* vlds = (double*)(lines + (FWDW ? DC_MAXS * P::LINE : sM));
    __shared__ double red[3 * KT / 64];
    const size_t n = (size_t)s * N * M;
    const double sr = ks.sr;
    // ---- request everything (clamped, not predicated): one memory latency for the whole row
    constexpr int NQ = (DC_MAXS * N + KT - 1) / KT, NE = 10;   // (NE: samples per thread requested up front -- the busiest k-rows of the spiral hold 2 400 samples against a mean of 550, and every further round of the loop below is another dependent memory latency for the whole launch: 4 -> 10, 15.9 -> 12.2 us per launch)
    double2 xv[NQ], zv[NQ];
    int slot[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = (tid + KT * q < sM) ? tid + KT * q : sM - 1;
        const int c = i / M, kw = i - c * M;
        const size_t g = (size_t)b * n + ((size_t)c * N + kh) * M + kw;
        xv[q] = ks.xhat[g];
        zv[q] = FWDW ? tmp[g] : ks.zhat[g];                // (FWDW: the h-pass output, same [c][kh][w] layout)
        slot[q] = op.kslot[kh * M + kw];
    }
    const int r0 = op.kptr[kh * M], r1 = op.kptr[(kh + 1) * M];
    const size_t mb = (size_t)b * op.m;
    KEntry en[NE];
    double2 yv[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = (r0 + tid + KT * q < r1) ? r0 + tid + KT * q : ((r1 > r0) ? r1 - 1 : 0);
        en[q] = op.ent[e]; yv[q] = ks.yk[mb + e];
    }
    double rv[NVQ];
    load_v(op, rv);
    if constexpr (FWDW) {
        // w-pass of the row (k_fwd_w<DC_SPECTRUM>'s arithmetic: same codelets, same scaling), then this thread's elements back into zv
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + KT * q;
            if (i < sM) { const int c = i / M; lines[c * P::LINE + (i - c * M)] = zv[q]; }
        }
        cd out[R2];
        int line2, k1;
        const bool act = fft_lds<R1, R2, false>(lines, s, op.tw, out, line2, k1);
        lds_barrier();
        if (act) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) lines[line2 * P::LINE + k1 + R1 * k2] = out[k2];
        }
        lds_barrier();
        const double sc = 1.0 / sqrt((double)N * (double)M);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + KT * q;
            if (i < sM) {
                const int c = i / M, kw = i - c * M;
                const cd z = lines[c * P::LINE + kw];
                zv[q] = make_double2(z.x * sc, z.y * sc);
                st_wt(ks.zhat + (size_t)b * n + ((size_t)c * N + kh) * M + kw, zv[q]);
            }
        }
        lds_barrier();                                     // (the buffer becomes the xhat0 lines below)
    }
    double accS = 0.0, accR = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + KT * q;
        if (i < sM) {
            const int c = i / M;
            lines[i] = xv[q];
            if (slot[q] >= 0) {
                const size_t ci = ((size_t)b * ks.ns + slot[q]) * s + c;
                const double2 ub = make_double2(zv[q].x * sr - xv[q].x * sr, zv[q].y * sr - xv[q].y * sr);   // sqrt(r) z - sqrt(r) x0
                ks.cx[ci] = xv[q];
                ks.cub[ci] = ub;
                accS += ub.x * ub.x + ub.y * ub.y;
            } else {
                const double dx = zv[q].x - xv[q].x, dy = zv[q].y - xv[q].y;
                accR += dx * dx + dy * dy;
            }
        }
    }
    store_v(op, rv, vlds);
    lds_barrier();
    double accT = 0.0;
    auto sample = [&](int e, const KEntry k, const double2 y) {
        double re = 0.0, im = 0.0;
        for (int c = 0; c < s; ++c) {
            const double v = vlds[k.t * s + c];
            const cd X = lines[c * M + k.kw];
            re += v * X.x; im += v * X.y;
        }
        const double2 u = make_double2(y.x - re, y.y - im);     // u(1:m) = y - A x0
        ks.ut[mb + e] = u;
        accT += u.x * u.x + u.y * u.y;
    };
#pragma unroll
    for (int q = 0; q < NE; ++q) { const int e = r0 + tid + KT * q; if (e < r1) sample(e, en[q], yv[q]); }
    for (int e = r0 + tid + KT * NE; e < r1; e += KT) sample(e, op.ent[e], ks.yk[mb + e]);
    block_sum2(accS, accT, red);
    const double r = block_sum(accR, red);
    if (tid == 0) {
        ks.pinit[(size_t)b * 2 * N + kh] = accS;
        ks.pinit[(size_t)b * 2 * N + N + kh] = accT;
        ks.pR[(size_t)b * N + kh] = r;
        if (kh == 0) ks.st[b].pad = 0;                      // (77: a workgroup of k_ks_persist gave up on this solve -- a new solve starts clean,
    }                                                       //  whichever kernel takes the first Golub-Kahan step)
}

// ---------------------------------------------------------------------------------------------------------------
// k_ks_a : u = B v - alpha (u / beta_prev) on one work unit (a run of slots and their samples).
// v is stored un-normalised; 1/alpha is applied on the fly.  u(m+1:end) is only normed here (k_ks_b stores it).
// partial sums: pu[ii&1][g] = |u(m+1:end)|^2 ,  pu[ii&1][G + g] = |u(1:m)|^2
// ---------------------------------------------------------------------------------------------------------------
template <class CP>
__global__ __launch_bounds__(KT, CP::MINW) void k_ks_a(OpDev op, KsDev ks) {
    constexpr int NEQ = CP::NEQ, NSQ = CP::NSQ;
    __shared__ cd vl[CP::SCAP * DC_MAXS];                  // v of the unit's slots, [slot][c]
    __shared__ double red[2 * KT / 64];
    extern __shared__ __align__(16) unsigned char smem[];
    double* vlds = (double*)smem;
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y, s = op.s;
    LsqrState* st = ks.st + b;
    KS_STAMP(0, 0);
    const int done = st->done;
    const KsUnit un = ks.unit[g];                                   // (one scalar load instead of a chain of dependent ones)
    const int s0 = un.s0, s1 = un.s1, ne = (s1 - s0) * s;
    const int e0 = un.e0, nsamp = un.e1 - e0;
    const size_t cb = ((size_t)b * ks.ns + s0) * s, mb = (size_t)b * op.m + e0;
    double2 rcv[NEQ], rub[NEQ];
#pragma unroll
    for (int q = 0; q < NEQ; ++q) {                        // (clamped, not predicated: the loads stay branch-free and in flight)
        const int i = (tid + KT * q < ne) ? tid + KT * q : 0;
        rcv[q] = ks.cv[cb + i]; rub[q] = ks.cub[cb + i];
    }
    KSample res[NSQ];
    double2 rut[NSQ];
#pragma unroll
    for (int q = 0; q < NSQ; ++q) {
        const int j = (tid + KT * q < nsamp) ? tid + KT * q : 0;
        res[q] = ks.es[e0 + j]; rut[q] = ks.ut[mb + j];
    }
    double rv[NVQ];
    load_v(op, rv);
    const LsqrScalars* O = &st->sc[(ks.ii - 1) & 1];
    const double oua = O->ua, obeta = O->beta, R = st->R;
    if (done) return;
    if (tid < 64) {
        const double pa = wave_sum(ks.pv[(ks.ii - 1) & 1] + (size_t)b * ks.G, ks.G);
        if (tid == 0) red[0] = pa;
    }
#pragma unroll
    for (int q = 0; q < NEQ; ++q) { const int i = tid + KT * q; if (i < ne) vl[i] = rcv[q]; }
    store_v(op, rv, vlds);
    lds_barrier();
    const double alpha = sqrt(red[0] + (oua * oua) * R);
    const double inv_alpha = 1.0 / alpha, inv_bprev = 1.0 / obeta, sr = ks.sr;
    KS_STAMP(0, 1);
    double acc_b = 0.0, acc_t = 0.0;
#pragma unroll
    for (int q = 0; q < NEQ; ++q) {
        if (tid + KT * q < ne) {
            const double2 ub = ub_update(make_double2(rcv[q].x * inv_alpha, rcv[q].y * inv_alpha), rub[q], sr, alpha, inv_bprev);
            acc_b += ub.x * ub.x + ub.y * ub.y;
        }
    }
#pragma unroll
    for (int q = 0; q < NSQ; ++q) {
        const int j = tid + KT * q;
        if (j < nsamp) {
            const cd* vrow = vl + res[q].ls * s;
            const double* Vt = vlds + res[q].t * s;
            double re = 0.0, im = 0.0;
            for (int c = 0; c < s; ++c) { re += Vt[c] * vrow[c].x; im += Vt[c] * vrow[c].y; }
            double2 u;
            u.x = re * inv_alpha - alpha * (rut[q].x * inv_bprev);     // A v - alpha (u / beta_prev)
            u.y = im * inv_alpha - alpha * (rut[q].y * inv_bprev);
            ks.ut[mb + j] = u;
            acc_t += u.x * u.x + u.y * u.y;
        }
    }
    KS_STAMP(0, 2);
    block_sum2(acc_b, acc_t, red);
    if (tid == 0) {
        double* pu = ks.pu[ks.ii & 1] + (size_t)b * 2 * ks.G;
        pu[g] = acc_b;
        pu[ks.G + g] = acc_t;
    }
    KS_STAMP(0, 3);
}

// ---------------------------------------------------------------------------------------------------------------
// k_ks_b : scalars (every block, identical bits), then on one work unit
//   INIT : v = B'u / beta0 ; d = 0                     ITER : d = (v - thet d)/rho ; x += phi d ; v = B'u/beta - beta v
// B'u on a slot = sum over its samples of V(t,c) u(t,k)  +  sqrt(r) u(m+1:end).
// partial sums: pv[ii&1][g] = |v|^2 of the unit
// ---------------------------------------------------------------------------------------------------------------
template <class CP, bool INIT>
__global__ __launch_bounds__(KT, CP::MINW) void k_ks_b(OpDev op, KsDev ks) {
    constexpr int NEQ = CP::NEQ, NSQ = CP::NSQ, NGQ = CP::NGQ, SL = CP::SL, GPL = CP::GCAP / CP::SL;   // (GPL: a group's samples per lane)
    __shared__ double2 ulds[CP::ECAP];
    __shared__ cd part[CP::GCAPB * DC_MAXS];
    __shared__ unsigned short tlds[CP::ECAP];
    __shared__ int sgl[CP::SCAP + 1];
    __shared__ double red[2 * KT / 64];
    extern __shared__ __align__(16) unsigned char smem[];
    double* vlds = (double*)smem;
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y, s = op.s;
    LsqrState* st = ks.st + b;
    KS_STAMP(1, 0);
    const int done = INIT ? 0 : st->done;
    const KsUnit un = ks.unit[g];
    const int s0 = un.s0, s1 = un.s1, nsl = s1 - s0, ne = nsl * s;
    const int e0 = un.e0, nsamp = un.e1 - e0;
    const int g0 = un.g0, ng = un.g1 - g0;
    const size_t cb = ((size_t)b * ks.ns + s0) * s, mb = (size_t)b * op.m + e0;
    double2 rcv[NEQ], rub[NEQ], rd[NEQ], rx[NEQ];
#pragma unroll
    for (int q = 0; q < NEQ; ++q) {
        const int i = (tid + KT * q < ne) ? tid + KT * q : 0;
        rub[q] = ks.cub[cb + i];
        if (!INIT) { rcv[q] = ks.cv[cb + i]; rd[q] = ks.cd[cb + i]; rx[q] = ks.cx[cb + i]; }
    }
    KSample res[NSQ];
    double2 rut[NSQ];
#pragma unroll
    for (int q = 0; q < NSQ; ++q) {
        const int j = (tid + KT * q < nsamp) ? tid + KT * q : 0;
        res[q] = ks.es[e0 + j]; rut[q] = ks.ut[mb + j];
    }
    KsGroup rg[NGQ];
#pragma unroll
    for (int q = 0; q < NGQ; ++q) { const int gi = (tid + KT * q) / SL; rg[q] = ks.grp[g0 + ((gi < ng) ? gi : 0)]; }
    constexpr int NGS = (CP::SCAP + KT) / KT;              // first-group entries per thread (SCAP + 1 of them)
    int rsg[NGS];
#pragma unroll
    for (int q = 0; q < NGS; ++q) { const int i = tid + KT * q; rsg[q] = ks.sgrp[s0 + ((i <= nsl) ? i : 0)] - g0; }
    double rv[NVQ];
    load_v(op, rv);
    if (done) return;
    if (tid < 64) {
        double pa, pb, pc = 0.0;
        if (INIT) {
            pa = wave_sum(ks.pinit + (size_t)b * 2 * op.N, 2 * op.N);
            pb = wave_sum(ks.pR + (size_t)b * op.N, op.N);
            pc = wave_sum(ks.pz + (size_t)b * ks.nblk_z, ks.nblk_z);
        } else {
            pa = wave_sum(ks.pv[(ks.ii - 1) & 1] + (size_t)b * ks.G, ks.G);
            pb = wave_sum(ks.pu[ks.ii & 1] + (size_t)b * 2 * ks.G, 2 * ks.G);
        }
        if (tid == 0) { red[0] = pa; red[1] = pb; red[2] = pc; }
    }
#pragma unroll
    for (int q = 0; q < NSQ; ++q) { const int j = tid + KT * q; if (j < nsamp) { ulds[j] = rut[q]; tlds[j] = res[q].t; } }
#pragma unroll
    for (int q = 0; q < NGS; ++q) { const int i = tid + KT * q; if (i <= nsl) sgl[i] = rsg[q]; }
    store_v(op, rv, vlds);
    lds_barrier();
    const double pa = red[0], pb = red[1], pc = red[2];

    double inv_beta = 1.0, beta = 0.0, thet = 0.0, inv_rho = 1.0, phi = 0.0, alpha = 0.0, inv_alpha = 1.0, inv_bprev = 1.0;
    const double sr = ks.sr;
    const bool writer = g == 0 && tid == 0;
    if (INIT) {
        const double R = pb;
        const double beta0 = sqrt(pa + (sr * sr) * R);
        const double n2b = sqrt(st->ny2 + sr * sr * pc);
        const bool fin = (beta0 == 0.0 || n2b == 0.0);       // x0 already exact, or b = 0
        if (writer) {
            LsqrScalars S;
            S.c = 1.0; S.s = 0.0; S.phibar = beta0; S.normr = beta0; S.norma = 0.0; S.factor = beta0;
            S.thet = 0.0; S.rho = 1.0; S.phi = 0.0; S.beta = beta0; S.alpha = 0.0;
            S.ua = fin ? 0.0 : (sr * (1.0 / beta0)) * sr;     // v = sqrt(r) u(m+1:end) / beta0 on the never-sampled k
            S.ub = sr; S.uc = 0.0; S.ue = 0.0;
            st->sc[0] = S;
            st->R = R;
            st->ue_final = 0.0;
            st->n2b = n2b;
            st->tolb = ks.tol * n2b;
            st->iter = fin ? 0 : ks.maxit;
            st->flag = fin ? 0 : 1;
            st->done = fin ? 1 : 0;
            st->pad = 0;                                  // (77: a workgroup of k_ks_persist gave up on this solve)
            if (ks.hst) { LsqrState* h = ks.hst + b; h->iter = st->iter; h->flag = st->flag; h->done = st->done; }     // (visible to the host when the launch is over)
        }
        if (fin) return;
        inv_beta = 1.0 / beta0;
    } else {
        const LsqrScalars O = st->sc[(ks.ii - 1) & 1];
        const double R = st->R;
        alpha = sqrt(pa + (O.ua * O.ua) * R);
        inv_alpha = 1.0 / alpha;
        inv_bprev = 1.0 / O.beta;
        const double ua_n = O.ua * inv_alpha;                                           // v = v/alpha
        const double ub_n = __fma_rn(ua_n, sr, -(alpha * (O.ub * inv_bprev)));          // as ub_update
        beta = sqrt(pb + (ub_n * ub_n) * R);
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
        if (normar / (S.norma * O.normr) <= ks.tol) conv = true;         // min ||b - Bx|| test
        if (O.normr <= st->tolb) conv = true;                             // Bx = b test
        S.normr = fabs(S.s) * O.normr;
        S.factor = fabs(S.s * S.phi);
        inv_beta = 1.0 / beta; thet = S.thet; inv_rho = 1.0 / S.rho; phi = S.phi;
        S.uc = (ua_n - thet * O.uc) * inv_rho;                            // d = (v - thet d)/rho
        S.ue = O.ue + phi * S.uc;                                         // x = x + phi d
        S.ua = ((ub_n * inv_beta) * sr) - beta * ua_n;                    // v = B'u/beta - beta v
        S.ub = ub_n;
        if (writer) {
            st->sc[ks.ii & 1] = S;
            if (conv) {
                st->done = 1; st->flag = 0; st->iter = ks.ii - 1;
                if (ks.hst) { LsqrState* h = ks.hst + b; h->done = 1; h->flag = 0; h->iter = ks.ii - 1; }
            }
            else st->ue_final = S.ue;
        }
        if (conv) return;
    }
    KS_STAMP(1, 1);

    // ---- sum_t V(t,c) u(t,k) per group: SL lanes share a group's samples (interleaved), all channels, fixed tree
#pragma unroll
    for (int q = 0; q < NGQ; ++q) {
        const int gi = (tid + KT * q) / SL, sub = tid & (SL - 1);
        if (gi < ng) {
            const KsGroup gr = rg[q];
            double xr[DC_MAXS], xi[DC_MAXS];
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) { xr[c] = 0.0; xi[c] = 0.0; }
            double2 u[GPL];
            int t[GPL];
#pragma unroll
            for (int j = 0; j < GPL; ++j) {       // all of the lane's samples in flight at once
                const int e = gr.b + sub + SL * j;
                const bool ok = e < gr.e;
                u[j] = ulds[ok ? e : 0]; t[j] = tlds[ok ? e : 0];
                if (!ok) u[j] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < GPL; ++j) {
#pragma unroll
                for (int c = 0; c < DC_MAXS; ++c) {
                    const double v = vlds[t[j] * s + c];   // (c >= s reads a neighbour: finite garbage into unused sums)
                    xr[c] += v * u[j].x; xi[c] += v * u[j].y;
                }
            }
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) { xr[c] = group_sum<SL>(xr[c]); xi[c] = group_sum<SL>(xi[c]); }
            if (sub == 0) {
#pragma unroll
                for (int c = 0; c < DC_MAXS; ++c) part[gi * DC_MAXS + c] = mk(xr[c], xi[c]);
            }
        }
    }
    lds_barrier();
    KS_STAMP(1, 2);

    // ---- vector updates on the unit's (slot, channel) elements
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NEQ; ++q) {
        const int i = tid + KT * q;
        if (i < ne) {
            const int ls = i / s, c = i - ls * s;
            double qx = 0.0, qy = 0.0;
            for (int gi = sgl[ls]; gi < sgl[ls + 1]; ++gi) { const cd p = part[gi * DC_MAXS + c]; qx += p.x; qy += p.y; }
            const double vx = qx * inv_beta, vy = qy * inv_beta;
            double2 ub = rub[q];
            double2 vr;
            if (INIT) {
                // v = B'*u = A'*u(1:m) + sqrt(r) u(m+1:end)       PnP_ADMM.m:164-167
                vr = make_double2(vx + (ub.x * inv_beta) * sr, vy + (ub.y * inv_beta) * sr);
                ks.cd[cb + i] = make_double2(0.0, 0.0);
            } else {
                const double2 vh = make_double2(rcv[q].x * inv_alpha, rcv[q].y * inv_alpha);   // v = v/alpha
                ub = ub_update(vh, ub, sr, alpha, inv_bprev);                                     // u(m+1:end), as normed in k_ks_a
                ks.cub[cb + i] = ub;
                double2 dd = rd[q];
                dd.x = (vh.x - thet * dd.x) * inv_rho;            // d = (v - thet d)/rho
                dd.y = (vh.y - thet * dd.y) * inv_rho;
                ks.cd[cb + i] = dd;
                double2 xv = rx[q];
                xv.x += phi * dd.x; xv.y += phi * dd.y;           // x = x + phi d
                ks.cx[cb + i] = xv;
                vr = make_double2((vx + (ub.x * inv_beta) * sr) - beta * vh.x,
                                  (vy + (ub.y * inv_beta) * sr) - beta * vh.y);   // v = B'u - beta v
            }
            ks.cv[cb + i] = vr;
            acc += vr.x * vr.x + vr.y * vr.y;
        }
    }
    KS_STAMP(1, 3);
    const double tot = block_sum(acc, red);
    if (tid == 0) ks.pv[ks.ii & 1][(size_t)b * ks.G + g] = tot;
    KS_STAMP(1, 4);
}

// ---------------------------------------------------------------------------------------------------------------
// k_ks_persist (round 3): ALL iterations of one LSQR solve in ONE launch.
//
// The state of the iteration on a work unit -- v, u(m+1:end), d, x of its <= 64 slots and u(1:m) of its <= 1024 samples -- is 19 KB:
// it stays in this workgroup's registers and LDS for the whole solve, so an iteration touches no global vector at all.  What is
// left of an iteration is exactly its two grid-wide sums (|u|^2 -> beta, |v|^2 -> alpha): every workgroup publishes its partial as
// a tagged granule (two 8-byte words {lo32(value), tag}, {hi32(value), tag}, agent-scope relaxed atomic stores = write-through sc1
// stores) and one wave per workgroup polls the G (2G) granules of its slice with agent-scope relaxed atomic loads (sc1: served
// past the L1) until every tag carries this iteration's number, then adds them in the SAME canonical order as wave_sum() -- so
// alpha, beta, the stopping decision and every vector carry the same bits as k_ks_a / k_ks_b produce (tested bit for bit).  No
// fence, no counter: a granule is its own flag (8-byte agent atomics on both sides, MI355X_MICROARCH.md "Valid forms").
// The work that does not depend on a sum is placed in front of the wait for it (staging v in LDS; the scatter sums V' u per group),
// so most of a hand-off's latency is covered.
// Requirement: the G x B workgroups must be resident together (the host checks the occupancy and falls back to the two-launch
// iteration otherwise -- EPI masks, cut0, slice batches).  Every spin is bounded: on a time-out the kernel stores nothing but the
// abort flag (state 77), and the host repeats the solve with the two-launch iteration from the untouched inputs.
// ---------------------------------------------------------------------------------------------------------------
struct KsGran { unsigned long long w0, w1; };                       // {lo32(value) | tag << 32}, {hi32(value) | tag << 32}
constexpr int KS_SPIN_MAX = 1 << 17;                                 // polls before a waiting wave gives up (~0.1-0.3 s)

typedef __attribute__((address_space(1))) unsigned long long gu64;   // global address space: global_load / global_store ... sc1, never flat_
__device__ __forceinline__ void gran_store(KsGran* g, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    __hip_atomic_store((gu64*)&g->w0, (b & 0xFFFFFFFFull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu64*)&g->w1, (b >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// canonical sum of n granules (n <= 64 * NQ) by one wave, in the order of wave_sum(); returns false on time-out.  Lane 0 holds the sum.
template <int NQ>
__device__ __forceinline__ bool gran_sum(KsGran* g, int n, unsigned tag, double& out) {
    const int lane = threadIdx.x & 63;
    double r[NQ];
    bool ok = false;
    for (int spin = 0; spin < KS_SPIN_MAX; ++spin) {
        bool all = true;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = lane + 64 * q;
            KsGran* e = g + ((i < n) ? i : 0);
            const unsigned long long a = __hip_atomic_load((gu64*)&e->w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load((gu64*)&e->w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            all = all && ((unsigned)(a >> 32) == tag) && ((unsigned)(b >> 32) == tag);
            r[q] = __longlong_as_double((long long)((a & 0xFFFFFFFFull) | (b << 32)));
        }
        if (__builtin_amdgcn_ballot_w64(all) == ~0ull) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) a += (lane + 64 * q < n) ? r[q] : 0.0;
    out = wave_sum_all(a);                                          // (the same tree as wave_sum())
    return ok;
}

// the solve is over: tell the host through the pinned state (system-scope release: iter / done / flag of this thread are visible before the
// sequence word).  The host spins on that word instead of waiting for an event: an event record between this launch and the next kernel
// left the GPU idle for 5.7 us per x-update (tools/iter_times.py).
__device__ __forceinline__ void ks_tell_host(LsqrState* h, unsigned seq) {
    if (h) __hip_atomic_store(&h->pad, (int32_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum_t V(t,c) u(t,k) per scatter group of a unit (k_ks_persist; the same statements as k_ks_b's): SL lanes share a group's samples, fixed tree
template <class CP>
__device__ __forceinline__ void ks_group_sums(int tid, int ng, const KsGroup (&rg)[CP::NGQ], const double2* ulds, const unsigned short* tlds,
                                              const double* vlds, cd* part) {
    constexpr int NGQ = CP::NGQ, SL = CP::SL, GPL = CP::GCAP / CP::SL, s = DC_MAXS;
#pragma unroll
    for (int q = 0; q < NGQ; ++q) {
        const int gi = (tid + KT * q) / SL, sub = tid & (SL - 1);
        if (gi < ng) {
            const KsGroup gr = rg[q];
            double xr[DC_MAXS], xi[DC_MAXS];
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) { xr[c] = 0.0; xi[c] = 0.0; }
            double2 u[GPL];
            int t[GPL];
#pragma unroll
            for (int j = 0; j < GPL; ++j) {
                const int e = gr.b + sub + SL * j;
                const bool okk = e < gr.e;
                u[j] = ulds[okk ? e : 0]; t[j] = tlds[okk ? e : 0];
                if (!okk) u[j] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < GPL; ++j) {
#pragma unroll
                for (int c = 0; c < DC_MAXS; ++c) {
                    const double v = vlds[t[j] * s + c];
                    xr[c] += v * u[j].x; xi[c] += v * u[j].y;
                }
            }
#pragma unroll
            for (int c = 0; c < DC_MAXS; ++c) { xr[c] = group_sum<SL>(xr[c]); xi[c] = group_sum<SL>(xi[c]); }
            if (sub == 0) {
#pragma unroll
                for (int c = 0; c < DC_MAXS; ++c) part[gi * DC_MAXS + c] = mk(xr[c], xi[c]);
            }
        }
    }
}

template <class CP>
__global__ __launch_bounds__(KT, CP::MINW) void k_ks_persist(OpDev op, KsDev ks, KsGran* gu_all, KsGran* gv_all, unsigned tag0, int test_drop, int* sticky,
                                                             int init_here) {
    constexpr int NEQ = CP::NEQ, NSQ = CP::NSQ, NGQ = CP::NGQ, SL = CP::SL;
    __shared__ cd vl[CP::SCAP * DC_MAXS];                  // v of the unit's slots, [slot][c]   (k_ks_a)
    __shared__ double2 ulds[CP::ECAP];                     // u(1:m) of the unit's samples        (k_ks_b)
    __shared__ cd part[CP::GCAPB * DC_MAXS];
    __shared__ unsigned short tlds[CP::ECAP];
    __shared__ int sgl[CP::SCAP + 1];
    __shared__ double red[2 * KT / 64 + 4];
    __shared__ int abort_flag;
    extern __shared__ __align__(16) unsigned char smem[];
    double* vlds = (double*)smem;
    constexpr int s = DC_MAXS;                             // (the launcher requires op.s == DC_MAXS, the reference's 10 channels: row pitches become constants)
    const int tid = threadIdx.x, g = blockIdx.x, b = blockIdx.y + ks.b0, G = ks.G;
    LsqrState* st = ks.st + b;
    // (a spin of an EARLIER launch of this context timed out: nothing of this launch can be trusted to complete either -- requested here with
    //  the other operands, looked at below)
    const int stuck = __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const KsUnit un = ks.unit[g];
    const int s0 = un.s0, s1 = un.s1, nsl = s1 - s0, ne = nsl * s;
    const int e0 = un.e0, nsamp = un.e1 - e0;
    const int g0 = un.g0, ng = un.g1 - g0;
    const size_t cb = ((size_t)b * ks.ns + s0) * s, mb = (size_t)b * op.m + e0;
    KsGran* gu[2] = {gu_all + (size_t)b * 4 * G, gu_all + (size_t)b * 4 * G + 2 * G};     // [parity][2G]
    KsGran* gv[2] = {gv_all + (size_t)b * 2 * G, gv_all + (size_t)b * 2 * G + G};         // [parity][G]
    // ---- the unit's state, once
    double2 rcv[NEQ], rub[NEQ], rd[NEQ], rx[NEQ];
#pragma unroll
    for (int q = 0; q < NEQ; ++q) {
        const int i = (tid + KT * q < ne) ? tid + KT * q : 0;
        rub[q] = ks.cub[cb + i]; rx[q] = ks.cx[cb + i];
        if (!init_here) { rcv[q] = ks.cv[cb + i]; rd[q] = ks.cd[cb + i]; }
        else { rcv[q] = make_double2(0.0, 0.0); rd[q] = make_double2(0.0, 0.0); }      // (d = 0: k_ks_b<INIT>)
    }
    KSample res[NSQ];
    double2 rut[NSQ];
#pragma unroll
    for (int q = 0; q < NSQ; ++q) {
        const int j = (tid + KT * q < nsamp) ? tid + KT * q : 0;
        res[q] = ks.es[e0 + j]; rut[q] = ks.ut[mb + j];
    }
    KsGroup rg[NGQ];
#pragma unroll
    for (int q = 0; q < NGQ; ++q) { const int gi = (tid + KT * q) / SL; rg[q] = ks.grp[g0 + ((gi < ng) ? gi : 0)]; }
    constexpr int NGS = (CP::SCAP + KT) / KT;
    int rsg[NGS];
#pragma unroll
    for (int q = 0; q < NGS; ++q) { const int i = tid + KT * q; rsg[q] = ks.sgrp[s0 + ((i <= nsl) ? i : 0)] - g0; }
    double rv[NVQ];
    load_v(op, rv);
    LsqrScalars O;
    double R = 0.0, tolb = 0.0;
    const double sr = ks.sr;
    const double ny2 = st->ny2;
    if (!init_here) { O = st->sc[0]; R = st->R; tolb = st->tolb; }  // written by k_ks_b<INIT>
    if (!init_here && st->done) {                                   // x0 already exact, or b = 0 (uniform over the grid)
        if (g == 0 && tid == 0 && ks.hst) { LsqrState* h = ks.hst + b; h->done = 1; h->flag = st->flag; h->iter = st->iter; ks_tell_host(h, tag0); }
        return;
    }
    if (stuck) {                                                    // (uniform over the grid: the word was set before this launch began)
        if (g == 0 && tid == 0) { st->pad = 77; st->flag = 77; if (ks.hst) { LsqrState* h = ks.hst + b; h->flag = 77; h->done = 0; h->iter = 0; ks_tell_host(h, tag0); } }
        return;
    }
    if (tid < 64) {
        if (!init_here) {
            const double pa0 = wave_sum(ks.pv[0] + (size_t)b * G, G);   // |v|^2 partials of the INIT launch (plain: another kernel's output)
            if (tid == 0) red[0] = pa0;
        } else {                                                    // the sums k_ks_b<INIT> takes from k_ks_init_a's partials
            const double pa = wave_sum(ks.pinit + (size_t)b * 2 * op.N, 2 * op.N);
            const double pb = wave_sum(ks.pR + (size_t)b * op.N, op.N);
            const double pc = wave_sum(ks.pz + (size_t)b * ks.nblk_z, ks.nblk_z);
            if (tid == 0) { red[0] = pa; red[1] = pb; red[2] = pc; }
        }
    }
#pragma unroll
    for (int q = 0; q < NSQ; ++q) { const int j = tid + KT * q; if (j < nsamp) { tlds[j] = res[q].t; if (init_here) ulds[j] = rut[q]; } }
#pragma unroll
    for (int q = 0; q < NGS; ++q) { const int i = tid + KT * q; if (i <= nsl) sgl[i] = rsg[q]; }
    if (tid == 0) abort_flag = 0;
    store_v(op, rv, vlds);
    const bool writer = g == 0 && tid == 0;
    if (init_here) {
        // ---- the first Golub-Kahan step, k_ks_b<INIT>'s arithmetic in k_ks_b<INIT>'s order: beta0, v = B'u / beta0, d = 0 -- in this launch, so that
        // v and d never exist in memory and the solve is one launch less (round 5).  Scalars: every workgroup, identical bits.
        lds_barrier();
        const double pa = red[0], pc = red[2];
        R = red[1];
        const double beta0 = sqrt(pa + (sr * sr) * R);
        const double n2b = sqrt(ny2 + sr * sr * pc);
        const bool fin = (beta0 == 0.0 || n2b == 0.0);             // x0 already exact, or b = 0
        O.c = 1.0; O.s = 0.0; O.phibar = beta0; O.normr = beta0; O.norma = 0.0; O.factor = beta0;
        O.thet = 0.0; O.rho = 1.0; O.phi = 0.0; O.beta = beta0; O.alpha = 0.0;
        O.ua = fin ? 0.0 : (sr * (1.0 / beta0)) * sr;
        O.ub = sr; O.uc = 0.0; O.ue = 0.0;
        tolb = ks.tol * n2b;
        if (writer) {
            st->sc[0] = O; st->R = R; st->ue_final = 0.0; st->n2b = n2b; st->tolb = tolb;
            st->iter = fin ? 0 : ks.maxit; st->flag = fin ? 0 : 1; st->done = fin ? 1 : 0;
            if (fin) {
                if (ks.hst) { LsqrState* h = ks.hst + b; h->done = 1; h->flag = 0; h->iter = 0; }
                ks_tell_host(ks.hst ? ks.hst + b : nullptr, tag0);
            }
        }
        if (fin) return;
        const double inv_beta0 = 1.0 / beta0;
        ks_group_sums<CP>(tid, ng, rg, ulds, tlds, vlds, part);
        lds_barrier();
        double acc0 = 0.0;
#pragma unroll
        for (int q = 0; q < NEQ; ++q) {
            const int i = tid + KT * q;
            if (i < ne) {
                const int ls = i / s, c = i - ls * s;
                double qx = 0.0, qy = 0.0;
                for (int gi = sgl[ls]; gi < sgl[ls + 1]; ++gi) { const cd pp = part[gi * DC_MAXS + c]; qx += pp.x; qy += pp.y; }
                const double vx = qx * inv_beta0, vy = qy * inv_beta0;
                const double2 ub = rub[q];
                const double2 vr = make_double2(vx + (ub.x * inv_beta0) * sr, vy + (ub.y * inv_beta0) * sr);     // v = A'*u(1:m) + sqrt(r) u(m+1:end)
                rcv[q] = vr;
                acc0 += vr.x * vr.x + vr.y * vr.y;
            }
        }
        const double tot0 = block_sum(acc0, red + 4);
        if (tid == 0) gran_store(gv[0] + g, tot0, tag0 + 1u);      // (iteration 1 reads parity 0 with tag0 + 1: the slot iteration "0" would have used)
        lds_barrier();                                              // (part / ulds / red are rewritten by the first iteration)
    }
    double ue_final = 0.0;
    int conv_iter = -1;
    bool aborted = false;

// (diagnostic, knob lsqr_stamps = 1: 100 MHz stamps of iteration 50, read by tools/lsqr_persist_stamps.py)
#define PS(k) do { if (ks.stamps && ii == 50 && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 512) ks.stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
    for (int ii = 1; ii <= ks.maxit; ++ii) {
        const unsigned tagA = tag0 + 2u * (unsigned)ii, tagB = tagA + 1u;
        PS(0);
        // ================= k_ks_a: u = B v - alpha (u / beta_prev) =================
        // alpha = |v| needs the all-reduce of the previous iteration's partial sums; A v (un-normalised: sum_c V(t,c) v(k,c) per sample) does
        // not, so it is computed in front of the wait and covers the hand-off's latency
#pragma unroll
        for (int q = 0; q < NEQ; ++q) { const int i = tid + KT * q; if (i < ne) vl[i] = rcv[q]; }
        lds_barrier();
        double sre[NSQ], sim[NSQ];
#pragma unroll
        for (int q = 0; q < NSQ; ++q) {
            const int j = tid + KT * q;
            sre[q] = 0.0; sim[q] = 0.0;
            if (j < nsamp) {
                const cd* vrow = vl + res[q].ls * s;
                const double* Vt = vlds + res[q].t * s;
                double re = 0.0, im = 0.0;
#pragma unroll
                for (int c = 0; c < s; ++c) { re += Vt[c] * vrow[c].x; im += Vt[c] * vrow[c].y; }
                sre[q] = re; sim[q] = im;
            }
        }
        PS(1);
        if ((ii > 1 || init_here) && tid < 64) {                    // |v|^2 of the previous iteration (or of the first step above): all-reduce over the slice's workgroups
            double pa;
            const bool ok = gran_sum<(KS_GRAN_MAXG + 63) / 64>(gv[(ii - 1) & 1], G, tagB - 2u, pa);
            if (tid == 0) { red[0] = pa; if (!ok) abort_flag = 1; }
        }
        lds_barrier();
        if (abort_flag) { aborted = true; break; }
        const double pa = red[0];
        const double alpha = sqrt(pa + (O.ua * O.ua) * R);
        const double inv_alpha = 1.0 / alpha, inv_bprev = 1.0 / O.beta;
        double acc_b = 0.0, acc_t = 0.0;
#pragma unroll
        for (int q = 0; q < NEQ; ++q) {
            if (tid + KT * q < ne) {
                const double2 ub = ub_update(make_double2(rcv[q].x * inv_alpha, rcv[q].y * inv_alpha), rub[q], sr, alpha, inv_bprev);
                acc_b += ub.x * ub.x + ub.y * ub.y;
            }
        }
#pragma unroll
        for (int q = 0; q < NSQ; ++q) {
            const int j = tid + KT * q;
            if (j < nsamp) {
                double2 u;
                u.x = sre[q] * inv_alpha - alpha * (rut[q].x * inv_bprev);     // A v - alpha (u / beta_prev)
                u.y = sim[q] * inv_alpha - alpha * (rut[q].y * inv_bprev);
                rut[q] = u;
                ulds[j] = u;
                acc_t += u.x * u.x + u.y * u.y;
            }
        }
        PS(2);
        block_sum2(acc_b, acc_t, red + 4);
        PS(3);
        // (test_drop: the test hook of the time-out path -- one workgroup withholds one partial sum, every waiter must give up cleanly)
        if (tid == 0 && !(test_drop && g == 1 && ii == 2)) { gran_store(gu[ii & 1] + g, acc_b, tagA); gran_store(gu[ii & 1] + G + g, acc_t, tagA); }
        // ================= k_ks_b: what does not need beta first -- sum_t V(t,c) u(t,k) per scatter group =================
        // (block_sum2's barriers follow the stores into ulds)
        ks_group_sums<CP>(tid, ng, rg, ulds, tlds, vlds, part);
        PS(4);
        if (tid < 64) {                                             // |u|^2: all-reduce
            double pb;
            const bool ok = gran_sum<(2 * KS_GRAN_MAXG + 63) / 64>(gu[ii & 1], 2 * G, tagA, pb);
            if (tid == 0) { red[1] = pb; if (!ok) abort_flag = 1; }
        }
        lds_barrier();
        PS(5);
        if (abort_flag) { aborted = true; break; }
        const double pb = red[1];
        // ---- scalars + stopping tests (every thread, identical bits), exactly as k_ks_b<false>
        const double ua_n = O.ua * inv_alpha;
        const double ub_n = __fma_rn(ua_n, sr, -(alpha * (O.ub * inv_bprev)));
        const double beta = sqrt(pb + (ub_n * ub_n) * R);
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
        if (normar == 0.0) conv = true;
        if (normar / (S.norma * O.normr) <= ks.tol) conv = true;
        if (O.normr <= tolb) conv = true;
        S.normr = fabs(S.s) * O.normr;
        S.factor = fabs(S.s * S.phi);
        const double inv_beta = 1.0 / beta, thet = S.thet, inv_rho = 1.0 / S.rho, phi = S.phi;
        S.uc = (ua_n - thet * O.uc) * inv_rho;
        S.ue = O.ue + phi * S.uc;
        S.ua = ((ub_n * inv_beta) * sr) - beta * ua_n;
        S.ub = ub_n;
        if (conv) { conv_iter = ii - 1; break; }
        ue_final = S.ue;
        PS(6);
        // ---- vector updates on the unit's (slot, channel) elements, in registers
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < NEQ; ++q) {
            const int i = tid + KT * q;
            if (i < ne) {
                const int ls = i / s, c = i - ls * s;
                double qx = 0.0, qy = 0.0;
                for (int gi = sgl[ls]; gi < sgl[ls + 1]; ++gi) { const cd p = part[gi * DC_MAXS + c]; qx += p.x; qy += p.y; }
                const double vx = qx * inv_beta, vy = qy * inv_beta;
                const double2 vh = make_double2(rcv[q].x * inv_alpha, rcv[q].y * inv_alpha);   // v = v/alpha
                const double2 ub = ub_update(vh, rub[q], sr, alpha, inv_bprev);
                rub[q] = ub;
                double2 dd = rd[q];
                dd.x = (vh.x - thet * dd.x) * inv_rho;            // d = (v - thet d)/rho
                dd.y = (vh.y - thet * dd.y) * inv_rho;
                rd[q] = dd;
                rx[q].x += phi * dd.x; rx[q].y += phi * dd.y;     // x = x + phi d
                const double2 vr = make_double2((vx + (ub.x * inv_beta) * sr) - beta * vh.x,
                                                (vy + (ub.y * inv_beta) * sr) - beta * vh.y);   // v = B'u - beta v
                rcv[q] = vr;
                acc += vr.x * vr.x + vr.y * vr.y;
            }
        }
        PS(7);
        const double tot = block_sum(acc, red + 4);
        if (tid == 0) gran_store(gv[ii & 1] + g, tot, tagB);
        O = S;
        lds_barrier();                                              // (vl / ulds / part / red are rewritten by the next iteration)
        PS(8);
    }
    if (aborted) {
        // EVERY workgroup that gives up says so (device words: st->pad for k_ks_final_w, which runs after the whole grid and overrides whatever
        // workgroup 0 told the host -- it may have seen the late partial sum and finished; `sticky` for this context's later launches): the
        // unit's x was not stored, so a "done" from workgroup 0 alone would be a silent wrong answer
        if (tid == 0) {
            __hip_atomic_store(&st->pad, 77, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (writer) { st->flag = 77; if (ks.hst) { LsqrState* h = ks.hst + b; h->flag = 77; h->done = 0; h->iter = 0; ks_tell_host(h, tag0); } }
        return;
    }
    // ---- the solution x on the unit's slots (k_ks_final_w assembles the spectrum); scalars for the final kernels and the host
#pragma unroll
    for (int q = 0; q < NEQ; ++q) { const int i = tid + KT * q; if (i < ne) st_wt(ks.cx + cb + i, rx[q]); }
    if (writer) {
        st->ue_final = ue_final;
        if (conv_iter >= 0) { st->done = 1; st->flag = 0; st->iter = conv_iter; }
        if (ks.hst) {                                               // (every field the host reads, from this thread, before the release below)
            LsqrState* h = ks.hst + b;
            h->done = conv_iter >= 0 ? 1 : 0; h->flag = conv_iter >= 0 ? 0 : 1; h->iter = conv_iter >= 0 ? conv_iter : ks.maxit;
        }
        ks_tell_host(ks.hst ? ks.hst + b : nullptr, tag0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ks_final_w : one block per k-row kh.  Assemble xhat_out (the next x-update's xhat0; xhat itself stays untouched so the
// kernel can be re-run), ||y - P xhat||^2 of the
// row's samples (PnP_ADMM.m:106), then the conj-domain inverse w-pass into tmp (k_adj_h finishes the transform).
// ---------------------------------------------------------------------------------------------------------------
template <int R1, int R2>
__global__ __launch_bounds__(KT) void k_ks_final_w(OpDev op, KsDev ks, double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, M = N;
    __shared__ cd lds[DC_MAXS * P::LINE];
    __shared__ double red[KT / 64];
    extern __shared__ __align__(16) unsigned char smem[];
    double* vlds = (double*)smem;
    const int tid = threadIdx.x, kh = blockIdx.x, b = blockIdx.y, s = op.s, sM = s * M;
    const size_t n = (size_t)s * N * M;
    const double ue = ks.st[b].ue_final;
    // a workgroup of the one-launch iteration gave up on this solve (k_ks_persist, `aborted`): the host must not believe any "done"
    if (kh == 0 && tid == 0 && ks.hst && ks.st[b].pad == 77) { LsqrState* h = ks.hst + b; h->flag = 77; h->done = 0; h->iter = 0; }
    double rv[NVQ];
    load_v(op, rv);
    // ---- request everything (clamped, not predicated): the slots first, then the compact x / the two spectra
    constexpr int NQ = (DC_MAXS * N + KT - 1) / KT;
    int slot[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = (tid + KT * q < sM) ? tid + KT * q : sM - 1;
        slot[q] = op.kslot[kh * M + (i - (i / M) * M)];
    }
    double2 cxv[NQ], xv[NQ], zv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = (tid + KT * q < sM) ? tid + KT * q : sM - 1;
        const int c = i / M, kw = i - c * M;
        const size_t g = (size_t)b * n + ((size_t)c * N + kh) * M + kw;
        cxv[q] = ks.cx[((size_t)b * ks.ns + ((slot[q] >= 0) ? slot[q] : 0)) * s + c];
        xv[q] = ks.xhat[g]; zv[q] = ks.zhat[g];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = tid + KT * q;
        if (i < sM) {
            const int c = i / M, kw = i - c * M;
            const size_t g = (size_t)b * n + ((size_t)c * N + kh) * M + kw;
            const double2 val = (slot[q] >= 0) ? cxv[q]
                                               : make_double2(xv[q].x + ue * (zv[q].x - xv[q].x), xv[q].y + ue * (zv[q].y - xv[q].y));
            st_wt(ks.xhat_out + g, val);
            lds[c * P::LINE + kw] = val;
        }
    }
    store_v(op, rv, vlds);
    lds_barrier();
    if (ks.pdiag) {
        const size_t mb = (size_t)b * op.m;
        double acc = 0.0;
        for (int e = op.kptr[kh * M] + tid; e < op.kptr[(kh + 1) * M]; e += KT) {
            const KEntry en = op.ent[e];
            double re = 0.0, im = 0.0;
            for (int c = 0; c < s; ++c) {
                const double v = vlds[en.t * s + c];
                const cd X = lds[c * P::LINE + en.kw];
                re += v * X.x; im += v * X.y;
            }
            const double2 yv = ks.yk[mb + e];
            const double dx = yv.x - re, dy = yv.y - im;
            acc += dx * dx + dy * dy;
        }
        const double tot = block_sum(acc, red);
        if (tid == 0) ks.pdiag[(size_t)b * N + kh] = tot;
    }
    lds_barrier();
    for (int i = tid; i < sM; i += KT) {                   // conjugate: inverse transform by conj-FFT-conj
        const int c = i / M, kw = i - c * M;
        const cd v = lds[c * P::LINE + kw];
        lds[c * P::LINE + kw] = mk(v.x, -v.y);
    }
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, false>(lds, s, op.tw, out, line2, k1)) {
        double2* dst = tmp + (size_t)b * n + ((size_t)line2 * N + kh) * M;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) st_wt(dst + k1 + R1 * k2, out[k2]);
    }
}

constexpr size_t KS_LDS_TOTAL = 160 * 1024;   // LDS of one CU (MI355X_MICROARCH: a single workgroup may declare all of it)

// Dynamic LDS (the row's spectrum lines + V) a kernel may use = the CU's 160 KB minus its static arrays.  V for the
// longest cut of the reference (cut0: T = 1000, s = 10, main_recon_tsmis_FFT.m:41-44) is 80 KB.
int allow_big_lds(qmri_ctx* ctx, const void* fn, size_t* max_dyn = nullptr) {
    hipFuncAttributes fa;
    QMRI_HIP(ctx, hipFuncGetAttributes(&fa, fn));
    const size_t dyn = KS_LDS_TOTAL - std::min(KS_LDS_TOTAL, (size_t)fa.sharedSizeBytes);
    QMRI_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    if (max_dyn) *max_dyn = dyn;
    return QMRI_OK;
}

template <int R1, int R2>
int launch_final_t(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, double2* tmp) {
    if (!ctx->ks_lds_attr[1]) { QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_final_w<R1, R2>)); ctx->ks_lds_attr[1] = true; }
    k_ks_final_w<R1, R2><<<dim3(op.N, B), dim3(KT), (size_t)ks.vcap * 8, ctx->stream>>>(op, ks, tmp);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

template <int N, int R1, int R2, class CP>
int lds_fits_t(qmri_ctx* ctx, int s, int M, int vcap, bool* ok) {
    const size_t vb = (size_t)vcap * 8;
    const void* fns[6] = {(const void*)k_ks_init_a<R1, R2, false>, (const void*)k_ks_a<CP>, (const void*)k_ks_b<CP, true>, (const void*)k_ks_b<CP, false>,
                          (const void*)k_ks_final_w<R1, R2>, (const void*)k_ks_init_a<R1, R2, true>};
    const size_t need[6] = {(size_t)s * M * 16 + vb, vb, vb, vb, vb, (size_t)DC_MAXS * Plan<R1, R2>::LINE * 16 + vb};
    *ok = true;
    for (int i = 0; i < 6; ++i) {
        size_t dyn = 0;
        QMRI_TRY(allow_big_lds(ctx, fns[i], &dyn));
        if (need[i] > dyn) *ok = false;
    }
    return QMRI_OK;
}
template <class CP>
int lds_fits_c(qmri_ctx* ctx, int N, int s, int M, int vcap, bool* ok) {
    switch (N) {
        case 224: return lds_fits_t<224, 16, 14, CP>(ctx, s, M, vcap, ok);
        case 128: return lds_fits_t<128, 16, 8, CP>(ctx, s, M, vcap, ok);
        case 64: return lds_fits_t<64, 8, 8, CP>(ctx, s, M, vcap, ok);
        default: return lds_fits_t<32, 8, 4, CP>(ctx, s, M, vcap, ok);
    }
}

}  // namespace

// Does V (vcap doubles) fit next to the other LDS arrays of every k-space LSQR kernel for this grid, with the unit capacities `caps` (KS_CAPS)?
int ks_lds_fits(qmri_ctx* ctx, int N, int s, int M, int vcap, int caps, bool* ok) {
    switch (caps) {
        case 1: return lds_fits_c<Caps1>(ctx, N, s, M, vcap, ok);
        case 2: return lds_fits_c<Caps2>(ctx, N, s, M, vcap, ok);
        case 3: return lds_fits_c<Caps3>(ctx, N, s, M, vcap, ok);
        default: return lds_fits_c<Caps0>(ctx, N, s, M, vcap, ok);
    }
}

template <class CP> static int ks_attrs_c(qmri_ctx* ctx) {
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_a<CP>));
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_b<CP, true>));
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_b<CP, false>));
    return QMRI_OK;
}
static int ks_attrs(qmri_ctx* ctx, int caps) {
    if (ctx->ks_lds_attr[0]) return QMRI_OK;
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<16, 14, false>)); QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<16, 14, true>));
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<16, 8, false>)); QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<16, 8, true>));
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<8, 8, false>)); QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<8, 8, true>));
    QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<8, 4, false>)); QMRI_TRY(allow_big_lds(ctx, (const void*)k_ks_init_a<8, 4, true>));
    if (caps == 1) QMRI_TRY(ks_attrs_c<Caps1>(ctx));
    else if (caps == 2) QMRI_TRY(ks_attrs_c<Caps2>(ctx));
    else if (caps == 3) QMRI_TRY(ks_attrs_c<Caps3>(ctx));
    else QMRI_TRY(ks_attrs_c<Caps0>(ctx));
    ctx->ks_lds_attr[0] = true;
    return QMRI_OK;
}
// the kernels of one unit shape, by KsDev::caps
#define KS_BY_CAPS(caps_, CALL)                                                      \
    do {                                                                             \
        if ((caps_) == 1) { typedef Caps1 CP; CALL; }                                \
        else if ((caps_) == 2) { typedef Caps2 CP; CALL; }                           \
        else if ((caps_) == 3) { typedef Caps3 CP; CALL; }                           \
        else { typedef Caps0 CP; CALL; }                                             \
    } while (0)

// residual + first Golub-Kahan vectors; ks.xhat / ks.zhat hold the unitary spectra of x0 and z
// hpass_tmp (nullable): the h-pass output of z's transform; the launch then also runs the w-pass and writes ks.zhat (k_ks_init_a<FWDW>)
int ks_launch_init(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, const double2* hpass_tmp, bool first_step) {
    QMRI_TRY(ks_attrs(ctx, ks.caps));
    const size_t vb = (size_t)ks.vcap * 8;
#define KS_INIT(R1_, R2_)                                                                                                       \
    do {                                                                                                                        \
        if (hpass_tmp) k_ks_init_a<R1_, R2_, true><<<dim3(op.N, B), dim3(KT), (size_t)DC_MAXS * Plan<R1_, R2_>::LINE * 16 + vb, ctx->stream>>>(op, ks, hpass_tmp); \
        else k_ks_init_a<R1_, R2_, false><<<dim3(op.N, B), dim3(KT), (size_t)op.s * op.M * 16 + vb, ctx->stream>>>(op, ks, nullptr);                              \
    } while (0)
    switch (op.N) {
        case 224: KS_INIT(16, 14); break;
        case 128: KS_INIT(16, 8); break;
        case 64: KS_INIT(8, 8); break;
        default: KS_INIT(8, 4); break;
    }
#undef KS_INIT
    // (first_step = false: k_ks_persist takes the first Golub-Kahan step itself, ks_launch_persist(..., init_here = true))
    if (first_step) KS_BY_CAPS(ks.caps, (k_ks_b<CP, true><<<dim3(ks.G, B), dim3(KT), vb, ctx->stream>>>(op, ks)));
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// LSQR iteration ks.ii (1-based)
int ks_launch_iter(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B) {
    const size_t vb = (size_t)ks.vcap * 8;
    hipEvent_t e0 = nullptr, e1 = nullptr;                          // profile level 2: one unit per iteration, k_ks_a's start to k_ks_b's end
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_LSQR));
    if (e0) {
        KS_BY_CAPS(ks.caps, (hipExtLaunchKernelGGL((k_ks_a<CP>), dim3(ks.G, B), dim3(KT), (std::uint32_t)vb, ctx->stream, e0, nullptr, 0, op, ks)));
        KS_BY_CAPS(ks.caps, (hipExtLaunchKernelGGL((k_ks_b<CP, false>), dim3(ks.G, B), dim3(KT), (std::uint32_t)vb, ctx->stream, nullptr, e1, 0, op, ks)));
    } else {
        KS_BY_CAPS(ks.caps, (k_ks_a<CP><<<dim3(ks.G, B), dim3(KT), vb, ctx->stream>>>(op, ks)));
        KS_BY_CAPS(ks.caps, (k_ks_b<CP, false><<<dim3(ks.G, B), dim3(KT), vb, ctx->stream>>>(op, ks)));
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// All iterations in one launch (k_ks_persist).  *ran = false when not even one slice's units would be resident at once, or the option is off:
// the caller then iterates with ks_launch_iter.  Round 5: a slice batch goes through the kernel as many slices at a time as are resident
// together (EPI: 250 one-per-CU units = one slice; the spiral: 2 x 250 two-per-CU units = two slices), launch after launch on the stream:
// the state of a solve then never leaves the registers, where the two-launch iteration moves 66 - 80 MB per slice and iteration.
// How many slices of a B-slice solve one k_ks_persist launch takes (0: the kernel does not apply -- too many units, s != 10, maxit < 1)
int ks_persist_plan(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, int* per_launch) {
    *per_launch = 0;
    if (ks.G > KS_GRAN_MAXG || ks.maxit < 1 || op.s != DC_MAXS) return QMRI_OK;    // (the kernel is written for the reference's s = 10)
    const size_t vb = (size_t)ks.vcap * 8;
    if (ctx->ks_persist_cap < 0) {
        const void* fn = nullptr;
        KS_BY_CAPS(ks.caps, (fn = (const void*)k_ks_persist<CP>));
        hipFuncAttributes fa;
        QMRI_HIP(ctx, hipFuncGetAttributes(&fa, fn));
        int nb = 0, ncu = 0;
        if (fa.sharedSizeBytes + vb <= KS_LDS_TOTAL) {
            QMRI_TRY(allow_big_lds(ctx, fn));
            QMRI_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, KT, vb));
        }
        QMRI_HIP(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        // residency actually granted to 256-thread blocks = min(API answer, 8, floor(800 / (ceil(sgpr / 16) * 16 + 16))) per CU
        // (MI355X_MICROARCH.md, "Residency and cooperative launch"); a kernel can use at most 102 SGPRs + VCC, i.e. >= 6 by that term
        ctx->ks_persist_cap = std::min(nb, 6) * ncu;
    }
    *per_launch = std::min(B, ctx->ks_persist_cap / ks.G);
    return QMRI_OK;
}

int ks_launch_persist(qmri_ctx* ctx, const OpDev& op, const KsDev& ks_in, int B, void* gran, unsigned tag0, bool init_here, bool* ran) {
    *ran = false;
    KsDev ks = ks_in;
    int per_launch = 0;
    QMRI_TRY(ks_persist_plan(ctx, op, ks, B, &per_launch));
    if (per_launch < 1) return QMRI_OK;
    const size_t vb = (size_t)ks.vcap * 8;
    KsGran* gu = (KsGran*)gran;
    KsGran* gv = gu + (size_t)B * 4 * ks.G;
    int* sticky = (int*)(gu + (size_t)ctx->op.maxB * 6 * ks.G);                         // (the word behind the granules: ks_gran_bytes)
    hipEvent_t e0 = nullptr, e1 = nullptr;                          // profile level 2: the whole solve's iterations as one unit
    QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_LSQR));
    const int drop = ctx->ks_persist == 2 ? 1 : 0;
    for (int b0 = 0; b0 < B; b0 += per_launch) {
        const int nb = std::min(per_launch, B - b0);
        ks.b0 = b0;
        if (e0) {
            hipEvent_t ea = b0 == 0 ? e0 : nullptr, eb = b0 + nb >= B ? e1 : nullptr;
            KS_BY_CAPS(ks.caps, (hipExtLaunchKernelGGL((k_ks_persist<CP>), dim3(ks.G, nb), dim3(KT), (std::uint32_t)vb, ctx->stream, ea, eb, 0, op, ks, gu, gv, tag0, drop, sticky, init_here ? 1 : 0)));
        } else {
            KS_BY_CAPS(ks.caps, (k_ks_persist<CP><<<dim3(ks.G, nb), dim3(KT), vb, ctx->stream>>>(op, ks, gu, gv, tag0, drop, sticky, init_here ? 1 : 0)));
        }
    }
    QMRI_HIP(ctx, hipGetLastError());
    *ran = true;
    return QMRI_OK;
}
size_t ks_gran_bytes(int G, int B) { return (size_t)B * 6 * G * sizeof(KsGran) + 64; }     // + the sticky time-out word

int ks_launch_final(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, double2* tmp) {
    switch (op.N) {
        case 224: return launch_final_t<16, 14>(ctx, op, ks, B, tmp);
        case 128: return launch_final_t<16, 8>(ctx, op, ks, B, tmp);
        case 64: return launch_final_t<8, 8>(ctx, op, ks, B, tmp);
        case 32: return launch_final_t<8, 4>(ctx, op, ks, B, tmp);
        default:
            qmri_set_error(ctx, "unsupported grid size N=%d (supported: 32, 64, 128, 224)", op.N);
            return QMRI_ERR_UNSUPPORTED;
    }
}

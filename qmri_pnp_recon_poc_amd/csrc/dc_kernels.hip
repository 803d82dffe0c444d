// dc_kernels.hip -- data-consistency kernels of the PnP-ADMM x-update (gfx950).
//
// Reference semantics (file:line relative to the reference root):
//   F.forward  main_recon_tsmis_FFT.m:228   y = P * vec(fft2(x)) / sqrt(N*M)
//   F.adjoint  main_recon_tsmis_FFT.m:229   x = ifft2(reshape(P' * y, N, M, [])) * sqrt(N*M)
//   P          setup_subsampling_spiralgrided.m:36-42 / setup_subsampling_epi.m:31-35
//   afun       PnP_ADMM.m:153-171            B = [A; sqrt(r) I]
//   lsqr call  PnP_ADMM.m:102                (MathWorks lsqr restated from its documentation, as in oracle/orc_lsqr.c)
//
// Arithmetic is complex fp64 throughout, as in the reference.  One operator application is two kernels:
//   forward :  k_fwd_h (FFT along h of 16 contiguous lines, transposed store)  ->  k_fwd_w (FFT along w of the s
//              channel lines of one k-row in LDS + gather-combine with V for every sample of that row)
//   adjoint :  k_adj_w (scatter-combine with V into LDS + inverse FFT along w)  ->  k_adj_h (inverse FFT along h)
// The LSQR vector updates, their norms (per-block partial sums, reduced in a fixed order by every consumer
// block, so results are run-to-run reproducible) and the scalar recurrences are fused into those four kernels:
// one LSQR iteration = 4 launches, no host round trip.  Measurements are kept in k-sorted order on the device
// so both the gather and the scatter stream contiguously; `perm` converts at the ABI boundary.
//
// Layouts: x [c][w][h] complex (MATLAB N x M x s); tmp [c][kh][w]; every per-slice array is slice-major.
#include "dc_device.h"
#include "conv6_act.h"

using namespace dcdev;

namespace {

// ---------------------------------------------------------------------------------------------------
// forward, pass 1: FFT along h (contiguous) of L lines; transposed store tmp[c][kh][w]
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2>
__global__ __launch_bounds__(NT) void k_fwd_h(OpDev op, const double2* __restrict__ src, double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = Cfg<R1, R2>::L;
    __shared__ cd lds[L * P::LINE];
    const int tid = threadIdx.x, b = blockIdx.y;
    const size_t n = (size_t)op.s * N * op.M;
    const size_t base = (size_t)b * n + (size_t)blockIdx.x * L * N;
    for (int i = tid; i < L * N; i += NT) {
        const int line = i / N, nn = i - line * N;
        lds[line * P::LINE + nn] = src[base + i];
    }
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, true>(lds, L, op.tw, out, line2, k1)) {
        const int l = blockIdx.x * L + line2;
        const int c = l / op.M, w = l - c * op.M;
        double2* dst = tmp + (size_t)b * n + (size_t)c * N * op.M + w;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) st_wt(dst + (size_t)(k1 + R1 * k2) * op.M, out[k2]);
    }
}

// ---------------------------------------------------------------------------------------------------
// The step between the denoiser and the next x-update in ONE launch (PnP_ADMM.m:138,144 -> :102):
//   v = double(I) * range + min ;  uold = uold + x - v ;  z = v - uold, with the partial sums of ||z||^2 -- and at once the h-pass of z's
// transform (k_fwd_h): a workgroup's 16 lines of z go from registers to LDS instead of to memory and back.  z itself is never stored
// (only the solve reads it, as a spectrum).  Its first workgroups also finish the per-layer |output| report of the forward pass that
// has just ended (conv6_act.h) -- the launch of k_act_check saved.
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2>
__global__ __launch_bounds__(NT) void k_dual_fwd_h(OpDev op, DualArgs d, ActCheckArgs ac, double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = Cfg<R1, R2>::L;
    __shared__ cd lds[L * P::LINE];
    __shared__ double red[NT / 64];
    __shared__ float redf[4];
    const int tid = threadIdx.x, b = blockIdx.y, M = op.M;
    if (b == 0) for (int layer = blockIdx.x; layer < ac.nlayers; layer += gridDim.x) act_check_layer(ac, layer, redf);     // (uniform per workgroup)
    const size_t n = (size_t)op.s * N * M;
    const size_t base = (size_t)b * n + (size_t)blockIdx.x * L * N;
    // every operand of the workgroup's 16 lines is requested before the first one is used (clamped, not predicated): ONE memory latency for
    // the launch -- with a load / compute / write-through store per loop iteration the 14 iterations paid 14 (16.2 us per launch against 9)
    constexpr int NIT = (L * N + NT - 1) / NT;
    float Iv[NIT], Jv[NIT];
    double2 xv[NIT], uv[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int i = (tid + NT * q < L * N) ? tid + NT * q : 0;
        const int line = i / N, h = i - line * N;
        const int l = blockIdx.x * L + line;
        const int c = l / M, w = l - c * M;
        const size_t pi = (size_t)c * d.pplane + (size_t)(w + 1) * d.php + h + 1;
        Iv[q] = d.out32[(size_t)b * d.out_bs + pi];
        Jv[q] = d.residual_noise ? d.in32[(size_t)b * d.in_bs + pi] : 0.f;
        xv[q] = d.x[base + i];
        uv[q] = d.u[base + i];
    }
    const double lo = d.norm[2 * b], range = d.norm[2 * b + 1];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
        const int i = tid + NT * q;
        if (i < L * N) {
            const int line = i / N, h = i - line * N;
            const float I = d.residual_noise ? Jv[q] - Iv[q] : Iv[q];   // denoiseImage_PnP_ADMM.m:99-104
            const double vv = (double)I * range + lo;                   // undo_norm_zero_to_one  :138,187-192
            double2 un = uv[q];
            un.x = un.x + xv[q].x - vv;                                 // uold = uold + x - v  :144
            un.y = un.y + xv[q].y - 0.0;
            st_wt(d.u + base + i, un);
            const double2 zz = make_double2(vv - un.x, 0.0 - un.y);     // z = v - uold  :102
            lds[line * P::LINE + h] = zz;
            acc += zz.x * zz.x + zz.y * zz.y;
        }
    }
    const double tot = block_sum(acc, red);
    if (tid == 0) d.pz[(size_t)b * gridDim.x + blockIdx.x] = tot;
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, true>(lds, L, op.tw, out, line2, k1)) {
        const int l = blockIdx.x * L + line2;
        const int c = l / M, w = l - c * M;
        double2* dst = tmp + (size_t)b * n + (size_t)c * N * M + w;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) st_wt(dst + (size_t)(k1 + R1 * k2) * M, out[k2]);
    }
}

// ---------------------------------------------------------------------------------------------------
// forward, pass 2: one block per k-row kh.  FFT along w of the s channel lines, then gather-combine
//   y[(t,k)] = (1/sqrt(NM)) sum_c V(t,c) Xhat_c[k]     for every sample of this row
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2, int MODE>
__global__ __launch_bounds__(NT) void k_fwd_w(OpDev op, LsqrDev ls, const double2* tmp,
                                               double2* y_out, double* __restrict__ pdiag,
                                               const double2* __restrict__ chat, double rr) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N;
    __shared__ cd lds[DC_MAXS * P::LINE];
    __shared__ double vlds[DC_VCAP];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y, kh = blockIdx.x, s = op.s, M = op.M;
    const size_t n = (size_t)s * N * M;
    const bool v_in_lds = op.T * s <= DC_VCAP;           // V(t,:) rows are read once per sample: keep them on chip
    if (v_in_lds) for (int i = tid; i < op.T * s; i += NT) vlds[i] = op.Vt[i];
    for (int i = tid; i < s * M; i += NT) {
        const int c = i / M, w = i - c * M;
        lds[c * P::LINE + w] = tmp[(size_t)b * n + ((size_t)c * N + kh) * M + w];
    }
    cd out[R2];
    int line2, k1;
    const bool act = fft_lds<R1, R2, false>(lds, s, op.tw, out, line2, k1);
    __syncthreads();
    if (act) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) lds[line2 * P::LINE + k1 + R1 * k2] = out[k2];
    }
    __syncthreads();
    const double sc = 1.0 / sqrt((double)N * (double)M);

    if (MODE == DC_DIRECT) {
        // closed-form x-update in k-space:  xhat(k) = (G_k + r I)^-1 (chat(k) + r zhat(k))
        // with unitary transforms; never-sampled k: xhat = zhat + chat/r.  Then the inverse FFT along w.
        for (int kw = tid; kw < M; kw += NT) {
            double rx[DC_MAXS], ry[DC_MAXS];
            const int slot = op.kslot[kh * M + kw];
            for (int c = 0; c < s; ++c) {
                const double2 ch = chat[(size_t)b * n + ((size_t)c * N + kh) * M + kw];
                const cd z = lds[c * P::LINE + kw];
                rx[c] = ch.x + rr * (z.x * sc);
                ry[c] = ch.y + rr * (z.y * sc);
            }
            if (slot >= 0) {
                const double* G = op.ginv + (size_t)slot * s * s;
                for (int c = 0; c < s; ++c) {
                    double ax = 0.0, ay = 0.0;
                    for (int e = 0; e < s; ++e) { const double g = G[c * s + e]; ax += g * rx[e]; ay += g * ry[e]; }
                    lds[c * P::LINE + kw] = mk(ax, -ay);       // conjugate: inverse transform by conj-FFT-conj
                }
            } else {
                const double ir = 1.0 / rr;
                for (int c = 0; c < s; ++c) lds[c * P::LINE + kw] = mk(rx[c] * ir, -ry[c] * ir);
            }
        }
        const bool act2 = fft_lds<R1, R2, false>(lds, s, op.tw, out, line2, k1);
        if (act2) {
            double2* dst = y_out + (size_t)b * n + ((size_t)line2 * N + kh) * M;   // y_out doubles as tmp here
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) st_wt(dst + k1 + R1 * k2, out[k2]);
        }
        return;
    }
    if (MODE == DC_SPECTRUM) {
        for (int i = tid; i < s * M; i += NT) {
            const int c = i / M, w = i - c * M;
            const cd z = lds[c * P::LINE + w];
            st_wt(y_out + (size_t)b * n + ((size_t)c * N + kh) * M + w, make_double2(z.x * sc, z.y * sc));
        }
        return;
    }

    const int e0 = op.kptr[kh * M], e1 = op.kptr[(kh + 1) * M];
    const size_t mb = (size_t)b * op.m;
    double acc = 0.0;
    for (int e = e0 + tid; e < e1; e += NT) {
        const KEntry en = op.ent[e];
        const double* vrow = v_in_lds ? vlds + en.t * s : op.Vt + (size_t)en.t * s;
        double re = 0.0, im = 0.0;
        for (int c = 0; c < s; ++c) {
            const double v = vrow[c];
            const cd X = lds[c * P::LINE + en.kw];
            re += v * X.x;
            im += v * X.y;
        }
        re *= sc; im *= sc;
        if (MODE == DC_PLAIN) {
            y_out[mb + op.perm[e]] = make_double2(re, im);
        } else if (MODE == DC_DIAG) {
            const double2 yv = ls.yk[mb + e];                       // ||y - F.forward(x)||  PnP_ADMM.m:106
            const double dx = yv.x - re, dy = yv.y - im;
            acc += dx * dx + dy * dy;
        }
    }
    if (MODE == DC_DIAG) {
        const double tot = block_sum(acc, red);
        if (tid == 0) pdiag[(size_t)b * N + kh] = tot;
    }
}

// ---------------------------------------------------------------------------------------------------
// adjoint, pass 1: one block per k-row kh.  Scatter-combine  Zhat_c[k] = sum_{t: k in Omega_t} V(t,c) y[(t,k)]
// (atomics-free: thread (kw,c) walks the samples of its own k), then inverse FFT along w.
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2>
__global__ __launch_bounds__(NT) void k_adj_w(OpDev op, const double2* __restrict__ y_in,
                                               double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N;
    __shared__ cd lds[DC_MAXS * P::LINE];
    __shared__ double vlds[DC_VCAP];
    __shared__ double2 ulds[DC_CH];
    __shared__ unsigned short tlds[DC_CH];
    const int tid = threadIdx.x, b = blockIdx.y, kh = blockIdx.x, s = op.s, M = op.M;
    const size_t n = (size_t)s * N * M;
    const bool v_in_lds = op.T * s <= DC_VCAP;
    if (v_in_lds) for (int i = tid; i < op.T * s; i += NT) vlds[i] = op.Vt[i];
    // Scatter-combine, atomics-free and in a fixed order: thread (kw, c) owns Zhat_c[kh,kw] and walks the samples of its
    // k in frame order.  The row's samples (k-sorted, hence contiguous) are first staged in LDS by the whole block with
    // coalesced loads, DC_CH at a time, so the serial walk of a heavily sampled k (200 frames at DC) runs at LDS latency,
    // four independent loads at a time, instead of one dependent global round trip per sample.
    const size_t mb = (size_t)b * op.m;
    constexpr int NQ = (N * DC_MAXS + NT - 1) / NT;
    double ar[NQ], ai[NQ];
    int e0[NQ], e1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int item = tid + NT * q;
        ar[q] = 0.0; ai[q] = 0.0;
        if (item < M * s) {
            const int kw = item / s;
            e0[q] = op.kptr[kh * M + kw]; e1[q] = op.kptr[kh * M + kw + 1];
        } else { e0[q] = 0; e1[q] = 0; }
    }
    const int r0 = op.kptr[kh * M], r1 = op.kptr[(kh + 1) * M];
    for (int lo = r0; lo < r1; lo += DC_CH) {
        const int hi = (lo + DC_CH < r1) ? lo + DC_CH : r1;
        __syncthreads();
        for (int i = tid; i < hi - lo; i += NT) {
            const int e = lo + i;
            ulds[i] = y_in[mb + op.perm[e]];
            tlds[i] = op.ent[e].t;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = (tid + NT * q) % s;
            const int a0 = (e0[q] > lo) ? e0[q] : lo, a1 = (e1[q] < hi) ? e1[q] : hi;
            double xr = ar[q], xi = ai[q];
            int e = a0;
            for (; e + 4 <= a1; e += 4) {
                double2 u0 = ulds[e - lo], u1 = ulds[e - lo + 1], u2 = ulds[e - lo + 2], u3 = ulds[e - lo + 3];
                const int t0 = tlds[e - lo], t1 = tlds[e - lo + 1], t2 = tlds[e - lo + 2], t3 = tlds[e - lo + 3];
                const double v0 = v_in_lds ? vlds[t0 * s + c] : op.Vt[(size_t)t0 * s + c];
                const double v1 = v_in_lds ? vlds[t1 * s + c] : op.Vt[(size_t)t1 * s + c];
                const double v2 = v_in_lds ? vlds[t2 * s + c] : op.Vt[(size_t)t2 * s + c];
                const double v3 = v_in_lds ? vlds[t3 * s + c] : op.Vt[(size_t)t3 * s + c];
                xr += v0 * u0.x; xi += v0 * u0.y;
                xr += v1 * u1.x; xi += v1 * u1.y;
                xr += v2 * u2.x; xi += v2 * u2.y;
                xr += v3 * u3.x; xi += v3 * u3.y;
            }
            for (; e < a1; ++e) {
                const double2 u0 = ulds[e - lo];
                const int t0 = tlds[e - lo];
                const double v0 = v_in_lds ? vlds[t0 * s + c] : op.Vt[(size_t)t0 * s + c];
                xr += v0 * u0.x; xi += v0 * u0.y;
            }
            ar[q] = xr; ai[q] = xi;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int item = tid + NT * q;
        if (item < M * s) {
            const int kw = item / s, c = item - kw * s;
            lds[c * P::LINE + kw] = mk(ar[q], -ai[q]);          // conjugate for the inverse transform
        }
    }
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, false>(lds, s, op.tw, out, line2, k1)) {
        double2* dst = tmp + (size_t)b * n + ((size_t)line2 * N + kh) * M;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) st_wt(dst + k1 + R1 * k2, out[k2]);
    }
}

// ---------------------------------------------------------------------------------------------------
// adjoint, pass 2: inverse FFT along h of L lines (c, w0..w0+L), un-conjugate, scale by 1/sqrt(NM)
//   (= ifft2(.)*sqrt(NM)), and the fused LSQR updates of v, d, x.
// ---------------------------------------------------------------------------------------------------
// u / mm (nullable): the ADMM loop's next step needs min / max of real(x + uold) over the whole stack (PnP_ADMM.m:115-121,174-184); the
// workgroup that produces x adds its lines' u and leaves its partial min / max in mm[b][blockIdx.x][2] -- the launch of k_minmax saved.
template <int R1, int R2>
__global__ __launch_bounds__(NT) void k_adj_h(OpDev op, const double2* __restrict__ tmp, double2* __restrict__ dst,
                                               const double2* __restrict__ u, double* __restrict__ mm) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = Cfg<R1, R2>::L;
    __shared__ cd lds[L * P::LINE];
    __shared__ double shm[2 * NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y, M = op.M;
    const size_t n = (size_t)op.s * N * M;
    const int l0 = blockIdx.x * L;
    const int c = l0 / M, w0 = l0 - c * M;
    const double2* srcp = tmp + (size_t)b * n + (size_t)c * N * M + w0;
    for (int i = tid; i < L * N; i += NT) {
        const int kh = i / L, line = i - kh * L;
        lds[line * P::LINE + kh] = srcp[(size_t)kh * M + line];
    }
    cd out[R2];
    int line2, k1;
    double lo = INFINITY, hi = -INFINITY;
    if (fft_lds<R1, R2, false>(lds, L, op.tw, out, line2, k1)) {
        const double sc = 1.0 / sqrt((double)N * (double)M);
        const size_t g0 = (size_t)b * n + (size_t)(l0 + line2) * N;
        double ur[R2];
        if (u) {
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) ur[k2] = u[g0 + k1 + R1 * k2].x;
        }
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const double2 xv = make_double2(out[k2].x * sc, -out[k2].y * sc);
            st_wt(dst + g0 + k1 + R1 * k2, xv);
            if (u) { const double v = xv.x + ur[k2]; lo = fmin(lo, v); hi = fmax(hi, v); }
        }
    }
    if (u) {                                                        // (uniform)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_down(lo, off, 64)); hi = fmax(hi, __shfl_down(hi, off, 64)); }
        const int wid = tid >> 6, lane = tid & 63;
        if (lane == 0) { shm[2 * wid] = lo; shm[2 * wid + 1] = hi; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int i = 1; i < NT / 64; ++i) { lo = fmin(lo, shm[2 * i]); hi = fmax(hi, shm[2 * i + 1]); }
            mm[((size_t)b * gridDim.x + blockIdx.x) * 2] = lo;
            mm[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = hi;
        }
    }
}

// y (ABI order: frame-major) -> k-sorted order; ||y||^2 as DC_SORT_BLOCKS partial sums per slice, added in block order by
// k_sort_y_sum (runs once per reconstruction; one block per slice took 0.33 ms for the 154 200 samples of the headline mask)
__global__ __launch_bounds__(NT) void k_sort_y(OpDev op, LsqrDev ls, const double2* __restrict__ y) {
    __shared__ double red[NT / 64];
    const int b = blockIdx.y;
    const size_t mb = (size_t)b * op.m;
    const int chunk = (op.m + DC_SORT_BLOCKS - 1) / DC_SORT_BLOCKS;
    const int e0 = blockIdx.x * chunk, e1 = min(op.m, e0 + chunk);
    double acc = 0.0;
    for (int e = e0 + threadIdx.x; e < e1; e += NT) {
        const double2 v = y[mb + op.perm[e]];
        ls.yk[mb + e] = v;
        acc += v.x * v.x + v.y * v.y;
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) ls.py[(size_t)b * DC_SORT_BLOCKS + blockIdx.x] = tot;
}
__global__ __launch_bounds__(64) void k_sort_y_sum(LsqrDev ls) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < DC_SORT_BLOCKS; ++k) tot += ls.py[(size_t)b * DC_SORT_BLOCKS + k];
        ls.st[b].ny2 = tot;
    }
}

// z = v - uold  (PnP_ADMM.m:102) with partial ||z||^2
__global__ __launch_bounds__(NT) void k_prepare_z(LsqrDev ls, size_t n, const double2* __restrict__ v,
                                                   const double2* __restrict__ u, double2* __restrict__ z) {
    __shared__ double red[NT / 64];
    const int b = blockIdx.y;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    double acc = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += NT) {
        const double2 a = v[(size_t)b * n + i], c = u[(size_t)b * n + i];
        const double2 zz = make_double2(a.x - c.x, a.y - c.y);
        st_wt(z + (size_t)b * n + i, zz);
        acc += zz.x * zz.x + zz.y * zz.y;
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) ls.pz[(size_t)b * ls.nblk_z + blockIdx.x] = tot;
}

template <int R1, int R2>
int launch_fwd_t(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* src,
                 double2* tmp, double2* y_out, double* pdiag, const double2* chat, double rr) {
    constexpr int L = Cfg<R1, R2>::L;
    dim3 gh(op.s * op.M / L, B), gw(op.N, B), blk(NT);
    hipStream_t st = ctx->stream;
    k_fwd_h<R1, R2><<<gh, blk, 0, st>>>(op, src, tmp);
    switch (mode) {
        case DC_FWD_H_ONLY: break;
        case DC_DIAG: k_fwd_w<R1, R2, DC_DIAG><<<gw, blk, 0, st>>>(op, ls, tmp, nullptr, pdiag, nullptr, 0.0); break;
        case DC_SPECTRUM: k_fwd_w<R1, R2, DC_SPECTRUM><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, nullptr, 0.0); break;
        case DC_DIRECT: k_fwd_w<R1, R2, DC_DIRECT><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, chat, rr); break;
        default: k_fwd_w<R1, R2, DC_PLAIN><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, nullptr, 0.0); break;
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

template <int R1, int R2>
int launch_adj_t(qmri_ctx* ctx, const OpDev& op, int B, const double2* y_in, double2* tmp, double2* dst, bool skip_w,
                 const double2* u = nullptr, double* mm = nullptr) {
    constexpr int L = Cfg<R1, R2>::L;
    dim3 gh(op.s * op.M / L, B), gw(op.N, B), blk(NT);
    hipStream_t st = ctx->stream;
    if (!skip_w) k_adj_w<R1, R2><<<gw, blk, 0, st>>>(op, y_in, tmp);
    k_adj_h<R1, R2><<<gh, blk, 0, st>>>(op, tmp, dst, u, mm);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

}  // namespace

bool dc_size_supported(int N) { return N == 224 || N == 128 || N == 64 || N == 32; }

#define DC_DISPATCH(N_, CALL)                                  \
    switch (N_) {                                              \
        case 224: return CALL(16, 14);                         \
        case 128: return CALL(16, 8);                          \
        case 64: return CALL(8, 8);                            \
        case 32: return CALL(8, 4);                            \
        default:                                               \
            qmri_set_error(ctx, "unsupported grid size N=%d (supported: 32, 64, 128, 224)", N_); \
            return QMRI_ERR_UNSUPPORTED;                       \
    }

int dc_launch_fwd(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* src, double2* tmp,
                  double2* y_out, double* pdiag) {
#define CALL_F(a, b) launch_fwd_t<a, b>(ctx, op, ls, mode, B, src, tmp, y_out, pdiag, nullptr, 0.0)
    DC_DISPATCH(op.N, CALL_F)
#undef CALL_F
}

int dc_launch_adj(qmri_ctx* ctx, const OpDev& op, int B, const double2* y_in, double2* tmp, double2* dst) {
#define CALL_A(a, b) launch_adj_t<a, b>(ctx, op, B, y_in, tmp, dst, false)
    DC_DISPATCH(op.N, CALL_A)
#undef CALL_A
}

int dc_launch_adj_h(qmri_ctx* ctx, const OpDev& op, int B, const double2* tmp, double2* dst, const double2* u, double* mm) {
#define CALL_H(a, b) launch_adj_t<a, b>(ctx, op, B, nullptr, const_cast<double2*>(tmp), dst, true, u, mm)
    DC_DISPATCH(op.N, CALL_H)
#undef CALL_H
}

// workgroups of the h-pass kernels per slice (= partial sums of |z|^2 of k_dual_fwd_h, partial min / max of k_adj_h)
int dc_hpass_blocks(const OpDev& op) {
    switch (op.N) {
        case 224: return op.s * op.M / Cfg<16, 14>::L;
        case 128: return op.s * op.M / Cfg<16, 8>::L;
        case 64: return op.s * op.M / Cfg<8, 8>::L;
        default: return op.s * op.M / Cfg<8, 4>::L;
    }
}

template <int R1, int R2>
static int launch_dual_t(qmri_ctx* ctx, const OpDev& op, int B, const DualArgs& d, const ActCheckArgs& ac, double2* tmp) {
    constexpr int L = Cfg<R1, R2>::L;
    k_dual_fwd_h<R1, R2><<<dim3(op.s * op.M / L, B), dim3(NT), 0, ctx->stream>>>(op, d, ac, tmp);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

// unnormalise + dual update + z = v - u + h-pass of z's transform into tmp (+ the forward pass's |output| report): see k_dual_fwd_h
int dc_launch_dual_fwd_h(qmri_ctx* ctx, const OpDev& op, int B, const DualArgs& d, const ActCheckArgs& ac, double2* tmp) {
#define CALL_U(a, b) launch_dual_t<a, b>(ctx, op, B, d, ac, tmp)
    DC_DISPATCH(op.N, CALL_U)
#undef CALL_U
}

int dc_launch_direct(qmri_ctx* ctx, const OpDev& op, int B, const double2* z, const double2* chat, double r,
                     double2* tmp, double2* x_out) {
    LsqrDev ls{};
    // tmp holds the h-pass output, then (in place, row by row) the conjugate-domain w-pass output of k_fwd_w
#define CALL_D(a, b) (launch_fwd_t<a, b>(ctx, op, ls, DC_DIRECT, B, z, tmp, tmp, nullptr, chat, r) != QMRI_OK \
                          ? (int)QMRI_ERR_HIP                                                                   \
                          : launch_adj_t<a, b>(ctx, op, B, nullptr, tmp, x_out, true))
    DC_DISPATCH(op.N, CALL_D)
#undef CALL_D
}

int dc_launch_sort_y(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* y) {
    k_sort_y<<<dim3(DC_SORT_BLOCKS, B), dim3(NT), 0, ctx->stream>>>(op, ls, y);
    k_sort_y_sum<<<dim3(B), dim3(64), 0, ctx->stream>>>(ls);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

int dc_launch_prepare_z(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* v, const double2* u,
                        double2* z) {
    const size_t n = (size_t)op.s * op.N * op.M;
    k_prepare_z<<<dim3(ls.nblk_z, B), dim3(NT), 0, ctx->stream>>>(ls, n, v, u, z);
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

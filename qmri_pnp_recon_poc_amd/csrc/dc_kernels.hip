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
#include "qmri_internal.h"
#include "fft_codelets.h"

using namespace qfft;

namespace {

constexpr int NT = 256;          // threads per block for every kernel in this file
constexpr int DC_MAXS = 12;      // channel lines the w-pass kernels hold in LDS

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += sh[i];
    return r;
}

// fixed-order reduction of a partial-sum array by a whole block (every block gets the same bits)
__device__ __forceinline__ double reduce_array(const double* __restrict__ p, int n, double* sh) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += NT) a += p[i];
    return block_sum(a, sh);
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
    __syncthreads();
    if (act1) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) a[n1] = lds[line1 * P::LINE + R2 * n1 + n2];
        Dft<R1>::run(a);
#pragma unroll
        for (int q = 1; q < R1; ++q) a[q] = mul(a[q], tw[n2 * q]);
    }
    __syncthreads();
    if (act1) {
#pragma unroll
        for (int q = 0; q < R1; ++q) lds[line1 * P::LINE + P::SP * n2 + q] = a[q];
    }
    __syncthreads();
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

// ---------------------------------------------------------------------------------------------------
// forward, pass 1: FFT along h (contiguous) of L lines; transposed store tmp[c][kh][w]
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2, int MODE>
__global__ __launch_bounds__(NT) void k_fwd_h(OpDev op, LsqrDev ls, const double2* __restrict__ src,
                                               const double2* __restrict__ zsrc, double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = Cfg<R1, R2>::L;
    __shared__ cd lds[L * P::LINE];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y;
    const size_t n = (size_t)op.s * N * op.M;
    const size_t base = (size_t)b * n + (size_t)blockIdx.x * L * N;
    double alpha = 0.0, inv_alpha = 1.0, inv_bprev = 1.0;
    const double sr = ls.sr;
    if (MODE == DC_LSQR_ITER) {
        if (ls.st[b].done) return;
        alpha = sqrt(reduce_array(ls.pv + (size_t)b * ls.nblk_h, ls.nblk_h, red));
        const double bp = sqrt(reduce_array(ls.pu[(ls.ii - 1) & 1] + (size_t)b * ls.npu, ls.npu, red));
        inv_alpha = 1.0 / alpha;
        inv_bprev = 1.0 / bp;
    }
    double acc = 0.0;
    for (int i = tid; i < L * N; i += NT) {
        const int line = i / N, nn = i - line * N;
        const size_t g = base + i;
        cd a;
        if (MODE == DC_LSQR_INIT) {
            // u(m+1:end) = sqrt(r) z - sqrt(r) x0          (b - B*x0, PnP_ADMM.m:102,160-162)
            const double2 xv = src[g], zv = zsrc[g];
            a = xv;
            double2 ub = make_double2(zv.x * sr - xv.x * sr, zv.y * sr - xv.y * sr);
            ls.ub[g] = ub;
            acc += ub.x * ub.x + ub.y * ub.y;
        } else if (MODE == DC_LSQR_ITER) {
            // v = v/alpha ;  u(m+1:end) = sqrt(r) v - alpha * (u/beta_prev)
            double2 v = ls.v[g];
            v.x *= inv_alpha; v.y *= inv_alpha;
            ls.v[g] = v;
            a = v;
            double2 ub = ls.ub[g];
            ub.x = v.x * sr - alpha * (ub.x * inv_bprev);
            ub.y = v.y * sr - alpha * (ub.y * inv_bprev);
            ls.ub[g] = ub;
            acc += ub.x * ub.x + ub.y * ub.y;
        } else {
            a = src[g];
        }
        lds[line * P::LINE + nn] = a;
    }
    if (MODE == DC_LSQR_INIT || MODE == DC_LSQR_ITER) {
        const double tot = block_sum(acc, red);
        if (tid == 0) ls.pu[ls.ii & 1][(size_t)b * ls.npu + blockIdx.x] = tot;
    }
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, true>(lds, L, op.tw, out, line2, k1)) {
        const int l = blockIdx.x * L + line2;
        const int c = l / op.M, w = l - c * op.M;
        double2* dst = tmp + (size_t)b * n + (size_t)c * N * op.M + w;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) dst[(size_t)(k1 + R1 * k2) * op.M] = out[k2];
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
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y, kh = blockIdx.x, s = op.s, M = op.M;
    const size_t n = (size_t)s * N * M;
    double alpha = 0.0, inv_bprev = 1.0;
    if (MODE == DC_LSQR_ITER) {
        if (ls.st[b].done) return;
        alpha = sqrt(reduce_array(ls.pv + (size_t)b * ls.nblk_h, ls.nblk_h, red));
        inv_bprev = 1.0 / sqrt(reduce_array(ls.pu[(ls.ii - 1) & 1] + (size_t)b * ls.npu, ls.npu, red));
    }
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
            for (int k2 = 0; k2 < R2; ++k2) dst[k1 + R1 * k2] = out[k2];
        }
        return;
    }
    if (MODE == DC_SPECTRUM) {
        for (int i = tid; i < s * M; i += NT) {
            const int c = i / M, w = i - c * M;
            const cd z = lds[c * P::LINE + w];
            y_out[(size_t)b * n + ((size_t)c * N + kh) * M + w] = make_double2(z.x * sc, z.y * sc);
        }
        return;
    }

    const int e0 = op.kptr[kh * M], e1 = op.kptr[(kh + 1) * M];
    const size_t mb = (size_t)b * op.m;
    double acc = 0.0;
    for (int e = e0 + tid; e < e1; e += NT) {
        const KEntry en = op.ent[e];
        const double* vrow = op.Vt + (size_t)en.t * s;
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
        } else if (MODE == DC_LSQR_INIT) {
            const double2 yv = ls.yk[mb + e];                       // u(1:m) = y - A*x0
            const double2 u = make_double2(yv.x - re, yv.y - im);
            ls.ut[mb + e] = u;
            acc += u.x * u.x + u.y * u.y;
        } else if (MODE == DC_LSQR_ITER) {
            double2 u = ls.ut[mb + e];                              // u(1:m) = A*v - alpha*(u/beta_prev)
            u.x = re - alpha * (u.x * inv_bprev);
            u.y = im - alpha * (u.y * inv_bprev);
            ls.ut[mb + e] = u;
            acc += u.x * u.x + u.y * u.y;
        } else if (MODE == DC_DIAG) {
            const double2 yv = ls.yk[mb + e];                       // ||y - F.forward(x)||  PnP_ADMM.m:106
            const double dx = yv.x - re, dy = yv.y - im;
            acc += dx * dx + dy * dy;
        }
    }
    if (MODE == DC_LSQR_INIT || MODE == DC_LSQR_ITER) {
        const double tot = block_sum(acc, red);
        if (tid == 0) ls.pu[ls.ii & 1][(size_t)b * ls.npu + ls.nblk_h + kh] = tot;
    } else if (MODE == DC_DIAG) {
        const double tot = block_sum(acc, red);
        if (tid == 0) pdiag[(size_t)b * N + kh] = tot;
    }
}

// ---------------------------------------------------------------------------------------------------
// adjoint, pass 1: one block per k-row kh.  Scatter-combine  Zhat_c[k] = sum_{t: k in Omega_t} V(t,c) y[(t,k)]
// (atomics-free: thread (kw,c) walks the samples of its own k), then inverse FFT along w.
// In LSQR modes this kernel also advances the scalar recurrences and evaluates the stopping tests.
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2, int MODE>
__global__ __launch_bounds__(NT) void k_adj_w(OpDev op, LsqrDev ls, const double2* __restrict__ y_in,
                                               double2* __restrict__ tmp) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N;
    __shared__ cd lds[DC_MAXS * P::LINE];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y, kh = blockIdx.x, s = op.s, M = op.M;
    const size_t n = (size_t)s * N * M;
    double inv_beta = 1.0;
    if (MODE == DC_LSQR_INIT) {
        LsqrState* st = ls.st + b;
        const double beta0 = sqrt(reduce_array(ls.pu[0] + (size_t)b * ls.npu, ls.npu, red));
        const double nz2 = reduce_array(ls.pz + (size_t)b * ls.nblk_z, ls.nblk_z, red);
        inv_beta = (beta0 != 0.0) ? 1.0 / beta0 : 0.0;
        if (kh == 0 && tid == 0) {
            LsqrScalars S;
            S.c = 1.0; S.s = 0.0; S.phibar = beta0; S.normr = beta0; S.norma = 0.0; S.factor = beta0;
            S.thet = 0.0; S.rho = 1.0; S.phi = 0.0; S.beta = beta0; S.alpha = 0.0;
            st->sc[0] = S;
            const double n2b = sqrt(st->ny2 + ls.sr * ls.sr * nz2);
            st->n2b = n2b;
            st->tolb = ls.tol * n2b;
            st->iter = ls.maxit; st->flag = 1;
            st->done = (beta0 == 0.0 || n2b == 0.0) ? 1 : 0;      // x0 already exact, or b = 0
            if (st->done) { st->iter = 0; st->flag = 0; }
        }
    } else if (MODE == DC_LSQR_ITER) {
        LsqrState* st = ls.st + b;
        if (st->done) return;
        const double alpha = sqrt(reduce_array(ls.pv + (size_t)b * ls.nblk_h, ls.nblk_h, red));
        const double beta = sqrt(reduce_array(ls.pu[ls.ii & 1] + (size_t)b * ls.npu, ls.npu, red));
        const LsqrScalars O = st->sc[(ls.ii - 1) & 1];
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
        if (kh == 0 && tid == 0) {
            st->sc[ls.ii & 1] = S;
            if (conv) { st->done = 1; st->flag = 0; st->iter = ls.ii - 1; }
        }
        if (conv) return;
        inv_beta = 1.0 / beta;
    }
    const size_t mb = (size_t)b * op.m;
    for (int item = tid; item < M * s; item += NT) {
        const int kw = item / s, c = item - kw * s;
        const int e0 = op.kptr[kh * M + kw], e1 = op.kptr[kh * M + kw + 1];
        double ar = 0.0, ai = 0.0;
        for (int e = e0; e < e1; ++e) {
            double2 yv;
            if (MODE == DC_PLAIN) yv = y_in[mb + op.perm[e]];
            else { yv = ls.ut[mb + e]; yv.x *= inv_beta; yv.y *= inv_beta; }
            const double v = op.Vt[(size_t)op.ent[e].t * s + c];
            ar += v * yv.x;
            ai += v * yv.y;
        }
        lds[c * P::LINE + kw] = mk(ar, -ai);          // conjugate for the inverse transform
    }
    cd out[R2];
    int line2, k1;
    if (fft_lds<R1, R2, false>(lds, s, op.tw, out, line2, k1)) {
        double2* dst = tmp + (size_t)b * n + ((size_t)line2 * N + kh) * M;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) dst[k1 + R1 * k2] = out[k2];
    }
}

// ---------------------------------------------------------------------------------------------------
// adjoint, pass 2: inverse FFT along h of L lines (c, w0..w0+L), un-conjugate, scale by 1/sqrt(NM)
//   (= ifft2(.)*sqrt(NM)), and the fused LSQR updates of v, d, x.
// ---------------------------------------------------------------------------------------------------
template <int R1, int R2, int MODE>
__global__ __launch_bounds__(NT) void k_adj_h(OpDev op, LsqrDev ls, const double2* __restrict__ tmp,
                                               double2* __restrict__ dst, double2* __restrict__ xio) {
    typedef Plan<R1, R2> P;
    constexpr int N = P::N, L = Cfg<R1, R2>::L;
    __shared__ cd lds[L * P::LINE];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, b = blockIdx.y, M = op.M;
    const size_t n = (size_t)op.s * N * M;
    double inv_beta = 1.0, beta = 0.0, thet = 0.0, inv_rho = 1.0, phi = 0.0;
    const double sr = ls.sr;
    if (MODE == DC_LSQR_INIT) {
        if (ls.st[b].done) return;
        const double beta0 = sqrt(reduce_array(ls.pu[0] + (size_t)b * ls.npu, ls.npu, red));
        inv_beta = 1.0 / beta0;
    } else if (MODE == DC_LSQR_ITER) {
        if (ls.st[b].done) return;
        const LsqrScalars S = ls.st[b].sc[ls.ii & 1];
        beta = S.beta; inv_beta = 1.0 / beta; thet = S.thet; inv_rho = 1.0 / S.rho; phi = S.phi;
    }
    const int l0 = blockIdx.x * L;
    const int c = l0 / M, w0 = l0 - c * M;
    const double2* srcp = tmp + (size_t)b * n + (size_t)c * N * M + w0;
    for (int i = tid; i < L * N; i += NT) {
        const int kh = i / L, line = i - kh * L;
        lds[line * P::LINE + kh] = srcp[(size_t)kh * M + line];
    }
    cd out[R2];
    int line2, k1;
    const bool act = fft_lds<R1, R2, false>(lds, L, op.tw, out, line2, k1);
    const double sc = 1.0 / sqrt((double)N * (double)M);
    double acc = 0.0;
    if (act) {
        const size_t g0 = (size_t)b * n + (size_t)(l0 + line2) * N;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const size_t g = g0 + k1 + R1 * k2;
            const double vx = out[k2].x * sc, vy = -out[k2].y * sc;
            if (MODE == DC_PLAIN) {
                dst[g] = make_double2(vx, vy);
            } else if (MODE == DC_LSQR_INIT) {
                // v = B'*u = A'*u(1:m) + sqrt(r) u(m+1:end)       PnP_ADMM.m:164-167
                const double2 ub = ls.ub[g];
                const double2 vr = make_double2(vx + (ub.x * inv_beta) * sr, vy + (ub.y * inv_beta) * sr);
                ls.v[g] = vr;
                ls.d[g] = make_double2(0.0, 0.0);
                acc += vr.x * vr.x + vr.y * vr.y;
            } else if (MODE == DC_LSQR_ITER) {
                const double2 ub = ls.ub[g];
                const double2 vh = ls.v[g];
                double2 dd = ls.d[g];
                dd.x = (vh.x - thet * dd.x) * inv_rho;            // d = (v - thet d)/rho
                dd.y = (vh.y - thet * dd.y) * inv_rho;
                ls.d[g] = dd;
                double2 xv = xio[g];
                xv.x += phi * dd.x; xv.y += phi * dd.y;           // x = x + phi d
                xio[g] = xv;
                const double2 vr = make_double2((vx + (ub.x * inv_beta) * sr) - beta * vh.x,
                                                (vy + (ub.y * inv_beta) * sr) - beta * vh.y);   // v = B'u - beta v
                ls.v[g] = vr;
                acc += vr.x * vr.x + vr.y * vr.y;
            }
        }
    }
    if (MODE == DC_LSQR_INIT || MODE == DC_LSQR_ITER) {
        const double tot = block_sum(acc, red);
        if (tid == 0) ls.pv[(size_t)b * ls.nblk_h + blockIdx.x] = tot;
    }
}

// y (ABI order: frame-major) -> k-sorted order; ||y||^2 ; one block per slice (runs once per reconstruction)
__global__ __launch_bounds__(NT) void k_sort_y(OpDev op, LsqrDev ls, const double2* __restrict__ y) {
    __shared__ double red[NT / 64];
    const int b = blockIdx.x;
    const size_t mb = (size_t)b * op.m;
    double acc = 0.0;
    for (int e = threadIdx.x; e < op.m; e += NT) {
        const double2 v = y[mb + op.perm[e]];
        ls.yk[mb + e] = v;
        acc += v.x * v.x + v.y * v.y;
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) ls.st[b].ny2 = tot;
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
        z[(size_t)b * n + i] = zz;
        acc += zz.x * zz.x + zz.y * zz.y;
    }
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) ls.pz[(size_t)b * ls.nblk_z + blockIdx.x] = tot;
}

template <int R1, int R2>
int launch_fwd_t(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* src,
                 const double2* zsrc, double2* tmp, double2* y_out, double* pdiag, const double2* chat, double rr) {
    constexpr int L = Cfg<R1, R2>::L;
    dim3 gh(op.s * op.M / L, B), gw(op.N, B), blk(NT);
    hipStream_t st = ctx->stream;
    switch (mode) {
        case DC_LSQR_INIT:
            k_fwd_h<R1, R2, DC_LSQR_INIT><<<gh, blk, 0, st>>>(op, ls, src, zsrc, tmp);
            k_fwd_w<R1, R2, DC_LSQR_INIT><<<gw, blk, 0, st>>>(op, ls, tmp, nullptr, nullptr, nullptr, 0.0);
            break;
        case DC_LSQR_ITER:
            k_fwd_h<R1, R2, DC_LSQR_ITER><<<gh, blk, 0, st>>>(op, ls, nullptr, nullptr, tmp);
            k_fwd_w<R1, R2, DC_LSQR_ITER><<<gw, blk, 0, st>>>(op, ls, tmp, nullptr, nullptr, nullptr, 0.0);
            break;
        case DC_DIAG:
            k_fwd_h<R1, R2, DC_PLAIN><<<gh, blk, 0, st>>>(op, ls, src, nullptr, tmp);
            k_fwd_w<R1, R2, DC_DIAG><<<gw, blk, 0, st>>>(op, ls, tmp, nullptr, pdiag, nullptr, 0.0);
            break;
        case DC_SPECTRUM:
            k_fwd_h<R1, R2, DC_PLAIN><<<gh, blk, 0, st>>>(op, ls, src, nullptr, tmp);
            k_fwd_w<R1, R2, DC_SPECTRUM><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, nullptr, 0.0);
            break;
        case DC_DIRECT:
            k_fwd_h<R1, R2, DC_PLAIN><<<gh, blk, 0, st>>>(op, ls, src, nullptr, tmp);
            k_fwd_w<R1, R2, DC_DIRECT><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, chat, rr);
            break;
        default:
            k_fwd_h<R1, R2, DC_PLAIN><<<gh, blk, 0, st>>>(op, ls, src, nullptr, tmp);
            k_fwd_w<R1, R2, DC_PLAIN><<<gw, blk, 0, st>>>(op, ls, tmp, y_out, nullptr, nullptr, 0.0);
            break;
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

template <int R1, int R2>
int launch_adj_t(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* y_in,
                 double2* tmp, double2* dst, double2* xio, bool skip_w) {
    constexpr int L = Cfg<R1, R2>::L;
    dim3 gh(op.s * op.M / L, B), gw(op.N, B), blk(NT);
    hipStream_t st = ctx->stream;
    switch (mode) {
        case DC_LSQR_INIT:
            k_adj_w<R1, R2, DC_LSQR_INIT><<<gw, blk, 0, st>>>(op, ls, nullptr, tmp);
            k_adj_h<R1, R2, DC_LSQR_INIT><<<gh, blk, 0, st>>>(op, ls, tmp, nullptr, nullptr);
            break;
        case DC_LSQR_ITER:
            k_adj_w<R1, R2, DC_LSQR_ITER><<<gw, blk, 0, st>>>(op, ls, nullptr, tmp);
            k_adj_h<R1, R2, DC_LSQR_ITER><<<gh, blk, 0, st>>>(op, ls, tmp, nullptr, xio);
            break;
        default:
            if (!skip_w) k_adj_w<R1, R2, DC_PLAIN><<<gw, blk, 0, st>>>(op, ls, y_in, tmp);
            k_adj_h<R1, R2, DC_PLAIN><<<gh, blk, 0, st>>>(op, ls, tmp, dst, nullptr);
            break;
    }
    QMRI_HIP(ctx, hipGetLastError());
    return QMRI_OK;
}

}  // namespace

bool dc_size_supported(int N) { return N == 224 || N == 256 || N == 128 || N == 64 || N == 32; }

int dc_nblk_h(int N, int M, int s) {
    int L = 16;
    if (N == 256) L = Cfg<16, 16>::L;
    else if (N == 224) L = Cfg<16, 14>::L;
    else if (N == 128) L = Cfg<16, 8>::L;
    else if (N == 64) L = Cfg<8, 8>::L;
    else if (N == 32) L = Cfg<8, 4>::L;
    return s * M / L;
}

#define DC_DISPATCH(N_, CALL)                                  \
    switch (N_) {                                              \
        case 224: return CALL(16, 14);                         \
        case 256: return CALL(16, 16);                         \
        case 128: return CALL(16, 8);                          \
        case 64: return CALL(8, 8);                            \
        case 32: return CALL(8, 4);                            \
        default:                                               \
            qmri_set_error(ctx, "unsupported grid size N=%d (supported: 32, 64, 128, 224, 256)", N_); \
            return QMRI_ERR_UNSUPPORTED;                       \
    }

int dc_launch_fwd(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* src,
                  const double2* zsrc, double2* tmp, double2* y_out, double* pdiag) {
#define CALL_F(a, b) launch_fwd_t<a, b>(ctx, op, ls, mode, B, src, zsrc, tmp, y_out, pdiag, nullptr, 0.0)
    DC_DISPATCH(op.N, CALL_F)
#undef CALL_F
}

int dc_launch_adj(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* y_in,
                  double2* tmp, double2* dst, double2* x_inout) {
#define CALL_A(a, b) launch_adj_t<a, b>(ctx, op, ls, mode, B, y_in, tmp, dst, x_inout, false)
    DC_DISPATCH(op.N, CALL_A)
#undef CALL_A
}

int dc_launch_direct(qmri_ctx* ctx, const OpDev& op, int B, const double2* z, const double2* chat, double r,
                     double2* tmp, double2* x_out) {
    LsqrDev ls{};
    // tmp holds the h-pass output, then (in place, row by row) the conjugate-domain w-pass output of k_fwd_w
#define CALL_D(a, b) (launch_fwd_t<a, b>(ctx, op, ls, DC_DIRECT, B, z, nullptr, tmp, tmp, nullptr, chat, r) != QMRI_OK \
                          ? (int)QMRI_ERR_HIP                                                                            \
                          : launch_adj_t<a, b>(ctx, op, ls, DC_PLAIN, B, nullptr, tmp, x_out, nullptr, true))
    DC_DISPATCH(op.N, CALL_D)
#undef CALL_D
}

int dc_launch_sort_y(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* y) {
    k_sort_y<<<dim3(B), dim3(NT), 0, ctx->stream>>>(op, ls, y);
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

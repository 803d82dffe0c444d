// lrtv_kernels.hip -- the LRTV solver option of main_recon_tsmis_FFT.m:273-282 behind the same operator:
//   x = FISTA_deep(data, param)     main_files/algorithms/LRTV/FISTA_deep.m:31-104   FISTA with backtracking on
//                                   0.5 |y - F.forward(x)|^2 + K |x|_TV, TV prox on the stacked real / imaginary image
//   prox_tv / norm_tv / gradient_op / div_op   unlocbox/prox/prox_tv.m:99-203, utils/norm_tv.m:45-55, gradient_op.m:41-49, div_op.m:42-55
// Oracle: oracle/orc_lrtv.c + oracle.fista_lrtv.  fp64 throughout, like the reference.
//
// The stacked image is [reshape(real(x),N,[]); reshape(imag(x),N,[])] (FISTA_deep.m:66,75): R = 2N rows, C = M*L columns,
// column-major -- finite differences also run across the real/imaginary seam and across channel seams, as in the reference.
// One inner iteration of prox_tv is ONE kernel: a block computes sol = b - gamma div(r, s) on its 64 x 16 tile plus one halo row
// and column (LDS), the tile's share of the objective, and the dual update with projection and FISTA momentum (r, s are
// ping-pong buffers: neighbouring tiles still read the old ones).  The stopping rule of an iteration is evaluated at the start
// of the NEXT launch, by every block, from the per-block partial sums (no fences, no atomics: the kernel boundary orders it);
// once it is met the launches return at once, so `sol` keeps the iterate that met the tolerance -- the host only looks at the
// flag every few launches.  HBM/L2 streaming work
// (10 arrays of R*C doubles per iteration), no matrix cores.
#include <algorithm>
#include <cmath>
#include <vector>
#include "qmri_internal.h"
#include <hip/hip_ext.h>

#pragma clang fp contract(off)      // the oracle is compiled without FMA contraction: keep a*b+c unfused here too

namespace {

constexpr int TVR = 64, TVC = 16, TVT = 256;       // tile rows x columns, threads
constexpr int RED_BLOCKS = 256;                    // fixed grid of the streaming reductions (partials are added in block order)

struct TvState { double obj[2]; int iter, done; };     // obj[k & 1] = objective of iteration k (obj of "iteration 0" = 0), written by block 0

__device__ __forceinline__ double tv_sol_at(const double* __restrict__ b, const double* __restrict__ r, const double* __restrict__ s,
                                            int R, int C, double gamma, int i, int j) {
    const size_t p = (size_t)j * R + i;
    double dv;
    if (i == 0) dv = r[p];
    else if (i == R - 1) dv = -r[p - 1];
    else dv = r[p] - r[p - 1];
    if (j == 0) dv += s[p];
    else if (j == C - 1) dv += -s[p - R];
    else dv += s[p] - s[p - R];
    return b[p] - gamma * dv;
}

// deterministic block sum of two values: wave trees in lane order, then the waves in wave order
__device__ __forceinline__ void tv_block_sum2(double& a, double& b, double* sh /* [2 * waves] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x = 0.0, y = 0.0;
        for (int w = 0; w < nw; ++w) { x += sh[2 * w]; y += sh[2 * w + 1]; }
        sh[0] = x; sh[1] = y;
    }
    __syncthreads();
    a = sh[0]; b = sh[1];
    __syncthreads();
}

__global__ __launch_bounds__(TVT) void k_tv_iter(const double* __restrict__ b, const double* __restrict__ r_in, const double* __restrict__ s_in,
                                                  double* __restrict__ r_out, double* __restrict__ s_out, double* __restrict__ pold,
                                                  double* __restrict__ qold, double* __restrict__ sol, double* __restrict__ partials,
                                                  TvState* __restrict__ st, int R, int C, double gamma, double mom, double tol, int maxit, int launch) {
    // Launch number `launch` (0-based) performs iteration launch + 1.  It first evaluates the stopping rule of iteration `launch`
    // (prox_tv.m:164-175) from the partial sums the previous launch left -- every block adds the same numbers in the same order,
    // so all blocks decide alike; the kernel boundary is the only synchronisation (a device-scope fence per block, as a "last
    // block reduces" scheme needs, costs an L2 write-back each on this part).
    if (st->done) return;                              // the stopping rule was met by an earlier launch
    __shared__ double tile[(TVC + 1) * (TVR + 1)];
    __shared__ double red[2 * (TVT / 64)];
    const unsigned nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (launch > 0) {
        const double* pp = partials + (size_t)((launch - 1) & 1) * 2 * nblk;
        double a = 0.0, t = 0.0;
        for (unsigned k = threadIdx.x; k < nblk; k += TVT) { a += pp[2 * k]; t += pp[2 * k + 1]; }
        tv_block_sum2(a, t, red);
        const double obj = 0.5 * a + gamma * t;
        const double rel = fabs(obj - st->obj[(launch - 1) & 1]) / obj;
        const bool stop = rel < tol || launch >= maxit;
        if (bid == 0 && threadIdx.x == 0) { st->obj[launch & 1] = obj; st->iter = launch; if (stop) st->done = 1; }
        if (stop) return;                              // `sol` keeps the iterate of iteration `launch`
    }
    const int tr = blockIdx.x, tc = blockIdx.y;
    const int i0 = tr * TVR, j0 = tc * TVC;
    // phase 1: sol on the tile.  Thread t owns row di = t & 63 of columns (t >> 6) + 4k: a wave reads 64 consecutive rows of one
    // column (512 contiguous bytes).  b, r, s stay in registers for phase 2.
    constexpr int NPT = TVC / (TVT / TVR);             // points per thread (4)
    static_assert(TVT % TVR == 0 && TVC % (TVT / TVR) == 0, "tile / thread mapping");
    const int di = threadIdx.x & (TVR - 1), c0 = threadIdx.x / TVR;
    const int i = i0 + di;
    double bv[NPT], rv[NPT], sv[NPT], vv[NPT], po[NPT], qo[NPT];     // (pold / qold requested here too: one memory latency, not two)
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int dj = c0 + (TVT / TVR) * k, j = j0 + dj;
        bv[k] = rv[k] = sv[k] = vv[k] = po[k] = qo[k] = 0.0;
        if (i < R && j < C) {
            const size_t p = (size_t)j * R + i;
            bv[k] = b[p]; rv[k] = r_in[p]; sv[k] = s_in[p]; po[k] = pold[p]; qo[k] = qold[p];
            double dv;                                 // div_op.m:47-54
            if (i == 0) dv = rv[k];
            else if (i == R - 1) dv = -r_in[p - 1];
            else dv = rv[k] - r_in[p - 1];
            if (j == 0) dv += sv[k];
            else if (j == C - 1) dv += -s_in[p - R];
            else dv += sv[k] - s_in[p - R];
            vv[k] = bv[k] - gamma * dv;
            sol[p] = vv[k];
        }
        tile[dj * (TVR + 1) + di] = vv[k];
    }
    // ... plus one halo row (below the tile) and one halo column (right of it): 16 + 64 points, one per thread
    if (threadIdx.x < TVC + TVR) {
        const bool row = threadIdx.x < TVC;
        const int hdi = row ? TVR : (int)threadIdx.x - TVC, hdj = row ? (int)threadIdx.x : TVC;
        const int hi = i0 + hdi, hj = j0 + hdj;
        tile[hdj * (TVR + 1) + hdi] = (hi < R && hj < C) ? tv_sol_at(b, r_in, s_in, R, C, gamma, hi, hj) : 0.0;
    }
    __syncthreads();
    // phase 2: objective shares and the dual update (prox_tv.m:160-186)
    const double c = 1.0 / (8.0 * gamma);
    double fid = 0.0, tv = 0.0;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int dj = c0 + (TVT / TVR) * k, j = j0 + dj;
        if (i >= R || j >= C) continue;
        const size_t p = (size_t)j * R + i;
        const double v = vv[k];
        const double dx = (i < R - 1) ? tile[dj * (TVR + 1) + di + 1] - v : 0.0;
        const double dy = (j < C - 1) ? tile[(dj + 1) * (TVR + 1) + di] - v : 0.0;
        const double d = bv[k] - v;
        fid += d * d;
        tv += sqrt(dx * dx + dy * dy);
        const double rr = rv[k] - c * dx, ss = sv[k] - c * dy;
        // projection onto the unit ball: weights = max(1, |(r, s)|) (prox_tv.m:178-181).  Inside the ball the division is by 1,
        // i.e. exact: skip the fp64 sqrt and divisions there (the result is bit-identical)
        const double n2 = rr * rr + ss * ss;
        double pp = rr, qq = ss;
        if (n2 > 1.0) { const double w = sqrt(n2); pp = rr / w; qq = ss / w; }
        r_out[p] = pp + mom * (pp - po[k]); pold[p] = pp;
        s_out[p] = qq + mom * (qq - qo[k]); qold[p] = qq;
    }
    tv_block_sum2(fid, tv, red);
    // phase 3: this block's shares, for the next launch
    if (threadIdx.x == 0) { double* po_ = partials + (size_t)(launch & 1) * 2 * nblk; po_[2 * bid] = fid; po_[2 * bid + 1] = tv; }
}

// ---- streaming pieces of the outer FISTA loop ------------------------------------------------------------------
// stacked image of x - step * g (g == nullptr: of x).  x, g: [L][M][N] complex; bimg: column-major 2N x (M L)
__global__ __launch_bounds__(256) void k_lrtv_stack(const double2* __restrict__ x, const double2* __restrict__ g, double step, double* __restrict__ bimg,
                                                     int N, size_t ncols) {
    const size_t total = ncols * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t col = e / N; const int n = (int)(e - col * N);
        double2 v = x[e];
        if (g) { const double2 q = g[e]; v.x = v.x - q.x * step; v.y = v.y - q.y * step; }
        bimg[col * 2 * N + n] = v.x;
        bimg[col * 2 * N + N + n] = v.y;
    }
}
__global__ __launch_bounds__(256) void k_lrtv_unstack(const double* __restrict__ bimg, double2* __restrict__ x, int N, size_t ncols) {
    const size_t total = ncols * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t col = e / N; const int n = (int)(e - col * N);
        x[e] = make_double2(bimg[col * 2 * N + n], bimg[col * 2 * N + N + n]);
    }
}
// norm_tv partial sums
__global__ __launch_bounds__(256) void k_lrtv_normtv(const double* __restrict__ I, int R, int C, double* __restrict__ partials) {
    __shared__ double red[2 * 4];
    const size_t total = (size_t)R * C;
    double acc = 0.0, zero = 0.0;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const int j = (int)(p / R), i = (int)(p - (size_t)j * R);
        const double dx = (i < R - 1) ? I[p + 1] - I[p] : 0.0;
        const double dy = (j < C - 1) ? I[p + R] - I[p] : 0.0;
        acc += sqrt(dx * dx + dy * dy);
    }
    tv_block_sum2(acc, zero, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// err = Fx - y (in place in fx), partial sums of |err|^2
__global__ __launch_bounds__(256) void k_lrtv_residual(double2* __restrict__ fx, const double2* __restrict__ y, size_t m, double* __restrict__ partials) {
    __shared__ double red[2 * 4];
    double acc = 0.0, zero = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < m; e += (size_t)gridDim.x * 256) {
        double2 v = fx[e]; const double2 q = y[e];
        v.x -= q.x; v.y -= q.y;
        fx[e] = v;
        acc += v.x * v.x + v.y * v.y;
    }
    tv_block_sum2(acc, zero, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// backtracking: partial sums of real(g' d) and |d|^2, d = x2 - x (FISTA_deep.m:86)
__global__ __launch_bounds__(256) void k_lrtv_backtrack(const double2* __restrict__ x, const double2* __restrict__ x2, const double2* __restrict__ g, size_t n,
                                                         double* __restrict__ partials) {
    __shared__ double red[2 * 4];
    double ip = 0.0, nn = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const double2 a = x[e], c = x2[e], q = g[e];
        const double dx = c.x - a.x, dy = c.y - a.y;
        ip += q.x * dx + q.y * dy;
        nn += dx * dx + dy * dy;
    }
    tv_block_sum2(ip, nn, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = ip; partials[2 * blockIdx.x + 1] = nn; }
}
// x = x2 + c (x2 - x2_prev); x2_prev = x2 (FISTA_deep.m:96-98)
__global__ __launch_bounds__(256) void k_lrtv_momentum(double2* __restrict__ x, const double2* __restrict__ x2, double2* __restrict__ x2prev, double c, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const double2 a = x2[e], p = x2prev[e];
        x[e] = make_double2(a.x + c * (a.x - p.x), a.y + c * (a.y - p.y));
        x2prev[e] = a;
    }
}

struct TvWork {                                      // device buffers of one prox_tv call on an R x C image
    double *r[2] = {nullptr, nullptr}, *s[2] = {nullptr, nullptr}, *pold = nullptr, *qold = nullptr, *partials = nullptr;
    TvState* st = nullptr;
    size_t n = 0; unsigned nblk = 0;
    int pred = 0;                                    // iterations the previous call needed: the first batch of launches of the next one
};

int tv_alloc(qmri_ctx* ctx, TvWork& w, int R, int C) {
    w.n = (size_t)R * C;
    w.nblk = (unsigned)(((R + TVR - 1) / TVR) * ((C + TVC - 1) / TVC));
    for (double** p : {&w.r[0], &w.r[1], &w.s[0], &w.s[1], &w.pold, &w.qold}) QMRI_HIP(ctx, hipMalloc((void**)p, w.n * sizeof(double)));
    QMRI_HIP(ctx, hipMalloc((void**)&w.partials, (size_t)4 * std::max(w.nblk, (unsigned)RED_BLOCKS) * sizeof(double)));   // two sets of (fid, tv) per block
    QMRI_HIP(ctx, hipMalloc((void**)&w.st, sizeof(TvState)));
    return QMRI_OK;
}
void tv_free(TvWork& w) {
    for (double* p : {w.r[0], w.r[1], w.s[0], w.s[1], w.pold, w.qold, w.partials}) if (p) (void)hipFree(p);
    if (w.st) (void)hipFree(w.st);
    w = TvWork();
}

// sol = prox_{gamma TV}(b) on device buffers; returns the iteration count and the final objective
int tv_prox_dev(qmri_ctx* ctx, TvWork& w, const double* d_b, int R, int C, double gamma, double tol, int maxit, double* d_sol, int* iters, double* obj) {
    if (gamma == 0.0) {                                                   // test_gamma: nothing to do
        QMRI_HIP(ctx, hipMemcpyAsync(d_sol, d_b, w.n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        *iters = 0; *obj = 0.0;
        return QMRI_OK;
    }
    for (double* p : {w.r[0], w.s[0], w.pold, w.qold}) QMRI_HIP(ctx, hipMemsetAsync(p, 0, w.n * sizeof(double), ctx->stream));
    QMRI_HIP(ctx, hipMemsetAsync(w.st, 0, sizeof(TvState), ctx->stream));
    const dim3 grid((R + TVR - 1) / TVR, (C + TVC - 1) / TVC);
    double told = 1.0;
    TvState h{};
    int launched = 0;
    constexpr int CHUNK = 4;                                              // launches between two looks at the flag ...
    while (launched < maxit + 1) {                                        // (launch `maxit` only evaluates iteration maxit)
        // ... the first batch is what the previous call needed plus the launch that evaluates the rule (consecutive FISTA steps
        // need nearly the same number of inner iterations)
        const int want = (launched == 0 && w.pred > 0) ? w.pred + 1 : CHUNK;
        const int n = std::min(want, maxit + 1 - launched), chunk_start = launched;
        for (int k = 0; k < n; ++k, ++launched) {
            const double t = (1.0 + std::sqrt(4.0 * told * told)) / 2.0;  // prox_tv.m:183 (as written)
            const double mom = (told - 1.0) / t;
            told = t;
            const int in = launched & 1;
            hipEvent_t e0 = nullptr, e1 = nullptr;                        // profile level 2: the kernel's own dispatch timestamps
            QMRI_TRY(qmri_prof_pair(ctx, &e0, &e1, PROF_TV));
            if (e0) hipExtLaunchKernelGGL(k_tv_iter, grid, dim3(TVT), 0, ctx->stream, e0, e1, 0, d_b, (const double*)w.r[in], (const double*)w.s[in], w.r[in ^ 1],
                                          w.s[in ^ 1], w.pold, w.qold, d_sol, w.partials, w.st, R, C, gamma, mom, tol, maxit, launched);
            else k_tv_iter<<<grid, dim3(TVT), 0, ctx->stream>>>(d_b, w.r[in], w.s[in], w.r[in ^ 1], w.s[in ^ 1], w.pold, w.qold, d_sol, w.partials, w.st,
                                                                R, C, gamma, mom, tol, maxit, launched);
        }
        QMRI_HIP(ctx, hipGetLastError());
        QMRI_HIP(ctx, hipMemcpyAsync(&h, w.st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // profile: launches 0 .. iter-1 performed an iteration; the later ones only evaluated the rule or returned at once
        QMRI_TRY(qmri_prof_chain_finish(ctx, std::max(0, std::min(n, (h.done ? h.iter : launched) - chunk_start))));
        if (h.done) break;
    }
    *iters = h.iter; *obj = h.obj[h.iter & 1];
    w.pred = h.iter;
    return QMRI_OK;
}

double host_sum(const std::vector<double>& v, size_t n, size_t stride = 1, size_t off = 0) {
    double a = 0.0;
    for (size_t i = 0; i < n; ++i) a += v[i * stride + off];
    return a;
}

}  // namespace

extern "C" int qmri_norm_tv(qmri_ctx* ctx, const double* I, int R, int C, double* out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_CHECK_ARG(ctx, I && out && R >= 2 && C >= 2, "qmri_norm_tv: I / out NULL or image smaller than 2 x 2");
    const size_t n = (size_t)R * C;
    double *d_I = nullptr, *d_p = nullptr;
    QMRI_HIP(ctx, hipMalloc((void**)&d_I, n * sizeof(double)));
    QMRI_HIP(ctx, hipMalloc((void**)&d_p, RED_BLOCKS * sizeof(double)));
    std::vector<double> hp(RED_BLOCKS);
    hipError_t e = hipMemcpyAsync(d_I, I, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) { k_lrtv_normtv<<<dim3(RED_BLOCKS), dim3(256), 0, ctx->stream>>>(d_I, R, C, d_p); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), d_p, RED_BLOCKS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_I); (void)hipFree(d_p);
    QMRI_HIP(ctx, e);
    *out = host_sum(hp, RED_BLOCKS);
    return QMRI_OK;
}

extern "C" int qmri_prox_tv(qmri_ctx* ctx, const double* b, int R, int C, double gamma, double tol, int maxit, double* sol,
                            int32_t* iters_out, double* obj_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_CHECK_ARG(ctx, b && sol && R >= 2 && C >= 2, "qmri_prox_tv: b / sol NULL or image smaller than 2 x 2");
    QMRI_CHECK_ARG(ctx, gamma >= 0.0 && tol > 0.0 && maxit >= 1, "qmri_prox_tv: gamma must be >= 0 (test_gamma), tol > 0, maxit >= 1");
    const size_t n = (size_t)R * C;
    TvWork w;
    double *d_b = nullptr, *d_sol = nullptr;
    int st = tv_alloc(ctx, w, R, C);
    int it = 0; double obj = 0.0;
    if (st == QMRI_OK && (hipMalloc((void**)&d_b, n * sizeof(double)) != hipSuccess || hipMalloc((void**)&d_sol, n * sizeof(double)) != hipSuccess)) {
        qmri_set_error(ctx, "qmri_prox_tv: hipMalloc failed"); st = QMRI_ERR_NOMEM;
    }
    if (st == QMRI_OK && hipMemcpyAsync(d_b, b, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { qmri_set_error(ctx, "qmri_prox_tv: H2D copy failed"); st = QMRI_ERR_HIP; }
    if (st == QMRI_OK) st = tv_prox_dev(ctx, w, d_b, R, C, gamma, tol, maxit, d_sol, &it, &obj);
    if (st == QMRI_OK && (hipMemcpyAsync(sol, d_sol, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                          hipStreamSynchronize(ctx->stream) != hipSuccess)) { qmri_set_error(ctx, "qmri_prox_tv: D2H copy failed"); st = QMRI_ERR_HIP; }
    tv_free(w);
    if (d_b) (void)hipFree(d_b);
    if (d_sol) (void)hipFree(d_sol);
    if (iters_out) *iters_out = it;
    if (obj_out) *obj_out = obj;
    return st;
}

extern "C" int qmri_lrtv(qmri_ctx* ctx, const void* y, const qmri_lrtv_params* prm, void* x_out, qmri_lrtv_info* info) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, y && prm && x_out, "qmri_lrtv: y / params / x_out must not be NULL");
    QMRI_CHECK_ARG(ctx, prm->K >= 0.0 && prm->iters >= 1 && prm->tol > 0.0, "qmri_lrtv: K >= 0, iters >= 1, tol > 0");
    const int N = o.N, R = 2 * o.N, C = o.M * o.s;
    const size_t n = (size_t)o.N * o.M * o.s, m = (size_t)o.m, ncols = (size_t)C;
    double step = prm->step > 0.0 ? prm->step : (double)n / (double)m;     // param.step = numel(X0)/numel(Y), main_recon_tsmis_FFT.m:277
    const double prox_tol = prm->prox_tol > 0.0 ? prm->prox_tol : 10e-4;   // prox_tv.m:99
    const int prox_maxit = prm->prox_maxit > 0 ? prm->prox_maxit : 200;    // prox_tv.m:101
    TvWork w;
    double2 *d_x = nullptr, *d_x2 = nullptr, *d_x2p = nullptr, *d_g = nullptr, *d_y = nullptr, *d_fx = nullptr;
    double *d_b = nullptr, *d_sol = nullptr;
    std::vector<void*> owned;
    auto alloc = [&](void** p, size_t bytes) { if (hipMalloc(p, bytes) != hipSuccess) return false; owned.push_back(*p); return true; };
    int st = tv_alloc(ctx, w, R, C);
    if (st == QMRI_OK && !(alloc((void**)&d_x, n * sizeof(double2)) && alloc((void**)&d_x2, n * sizeof(double2)) && alloc((void**)&d_x2p, n * sizeof(double2)) &&
                           alloc((void**)&d_g, n * sizeof(double2)) && alloc((void**)&d_y, m * sizeof(double2)) && alloc((void**)&d_fx, m * sizeof(double2)) &&
                           alloc((void**)&d_b, 2 * n * sizeof(double)) && alloc((void**)&d_sol, 2 * n * sizeof(double)))) {
        qmri_set_error(ctx, "qmri_lrtv: hipMalloc failed"); st = QMRI_ERR_NOMEM;
    }
    std::vector<double> hp(2 * RED_BLOCKS);
    qmri_lrtv_info inf{};
    auto body = [&]() -> int {
        QMRI_HIP(ctx, hipMemcpyAsync(d_y, y, m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        QMRI_HIP(ctx, hipMemsetAsync(d_x, 0, n * sizeof(double2), ctx->stream));            // x = zeros(N,M,L), FISTA_deep.m:45
        QMRI_HIP(ctx, hipMemsetAsync(d_x2p, 0, n * sizeof(double2), ctx->stream));
        auto partial_sum = [&](size_t cnt) -> int {
            QMRI_HIP(ctx, hipMemcpyAsync(hp.data(), w.partials, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return QMRI_OK;
        };
        auto half_sq_residual = [&](const double2* xin, double& out) -> int {               // 0.5 |F.forward(x) - y|^2, residual left in d_fx
            QMRI_TRY(qmri_forward_dev(ctx, xin, d_fx, 1));
            k_lrtv_residual<<<dim3(RED_BLOCKS), dim3(256), 0, ctx->stream>>>(d_fx, d_y, m, w.partials);
            QMRI_HIP(ctx, hipGetLastError());
            QMRI_TRY(partial_sum(RED_BLOCKS));
            out = 0.5 * host_sum(hp, RED_BLOCKS);
            return QMRI_OK;
        };
        double obj_prev = 0.0;
        long t = 1;
        for (int it = 1; it <= prm->iters; ++it) {
            double cvxobj = 0.0;
            QMRI_TRY(half_sq_residual(d_x, cvxobj));                                        // err = Fx - y; cvxobj, :58-62
            QMRI_TRY(qmri_adjoint_dev(ctx, d_fx, d_g, 1));                                  // grad1 = F.adjoint(err)
            k_lrtv_stack<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_x, nullptr, 0.0, d_b, N, ncols);
            k_lrtv_normtv<<<dim3(RED_BLOCKS), dim3(256), 0, ctx->stream>>>(d_b, R, C, w.partials);   // val = norm_tv(stacked x), :66
            QMRI_HIP(ctx, hipGetLastError());
            QMRI_TRY(partial_sum(RED_BLOCKS));
            const double val = host_sum(hp, RED_BLOCKS);
            for (;;) {                                                                      // backtracking line search, :69-94
                const double2* x2 = d_x2;
                if (prm->K > 0.0) {
                    k_lrtv_stack<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_x, d_g, step, d_b, N, ncols);   // x2 = x - grad1 * step, stacked
                    QMRI_HIP(ctx, hipGetLastError());
                    int pit = 0; double pobj = 0.0;
                    QMRI_TRY(tv_prox_dev(ctx, w, d_b, R, C, step * prm->K, prox_tol, prox_maxit, d_sol, &pit, &pobj));
                    inf.prox_calls += 1; inf.prox_iters_total += pit;
                    k_lrtv_unstack<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_sol, d_x2, N, ncols);
                } else {
                    k_lrtv_stack<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_x, d_g, step, d_b, N, ncols);
                    k_lrtv_unstack<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_b, d_x2, N, ncols);
                }
                QMRI_HIP(ctx, hipGetLastError());
                if (!prm->backtrack) break;
                double tmp = 0.0;
                QMRI_TRY(half_sq_residual(x2, tmp));
                k_lrtv_backtrack<<<dim3(RED_BLOCKS), dim3(256), 0, ctx->stream>>>(d_x, d_x2, d_g, n, w.partials);
                QMRI_HIP(ctx, hipGetLastError());
                QMRI_TRY(partial_sum(2 * RED_BLOCKS));
                const double ip = host_sum(hp, RED_BLOCKS, 2, 0), nn = host_sum(hp, RED_BLOCKS, 2, 1);
                if (tmp > cvxobj + ip + 1.0 / (2.0 * step) * nn) { step = step / 2.0; inf.halvings += 1; }   // 'reducing stepsize...'
                else break;
                if (inf.halvings > 60) { qmri_set_error(ctx, "qmri_lrtv: step size underflow in the line search"); return QMRI_ERR_UNSUPPORTED; }
            }
            const double cm = (double)(t - 1) / (double)(t + 2);
            k_lrtv_momentum<<<dim3(RED_BLOCKS * 4), dim3(256), 0, ctx->stream>>>(d_x, d_x2, d_x2p, cm, n);   // :96-98
            QMRI_HIP(ctx, hipGetLastError());
            t += 1;
            const double obj = cvxobj + prm->K * val;
            inf.iters = it; inf.obj = obj;
            if (std::fabs(obj - obj_prev) / obj < prm->tol) break;                          // :103
            obj_prev = obj;
        }
        inf.step = step;
        QMRI_HIP(ctx, hipMemcpyAsync(x_out, d_x, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return QMRI_OK;
    };
    if (st == QMRI_OK) st = body();
    if (st != QMRI_OK) (void)hipStreamSynchronize(ctx->stream);
    tv_free(w);
    for (void* p : owned) (void)hipFree(p);
    if (info) *info = inf;
    return st;
}

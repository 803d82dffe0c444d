/*
 * orc_admm.c -- the PnP-ADMM outer loop.  Test infrastructure only.
 *
 * Restated from main_files/algorithms/PnP_ADMM/PnP_ADMM.m:
 *   :76-78    x = param.X0; v = x; uold = 0
 *   :93-146   for i = 1:max_iter  { lsqr x-update :102 ; diagnostics :106-109 ; v = real(x+uold) :115-118 ;
 *             norm_zero_to_one :121,174-184 ; net(v) / net(cat(3,v,noise_map)) :124-135 ;
 *             undo_norm_zero_to_one :138,187-192 ; uold = uold + x - v :144 }
 *   returns x (the last lsqr solution, complex).
 * Parameters main_recon_tsmis_FFT.m:285-292; noise map build_noise_map.m:19 (constant plane).
 */
#include "orc_internal.h"
#include <omp.h>

/* wall-clock split of the most recent orc_pnp_admm call (bench.py cpu_baseline; not thread-safe, one call at a time):
 * [0] lsqr x-update, [1] the two diagnostics, [2] denoiser incl. casts, [3] normalise / un-normalise / dual update */
static double g_stage_s[4];
void orc_admm_stage_seconds(double* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_stage_s[i]; }

void orc_pnp_admm(const orc_op* op, const orc_net* net, const double* y, const orc_admm_params* p,
                  const double* x0, const double* gt, double* x_out, double* diag_out,
                  int32_t* lsqr_iters_out) {
    const int N = op->N, M = op->M, s = op->s;
    const size_t plane = (size_t)N * M, n = plane * s, n2 = 2 * n, m2 = 2 * (size_t)op->m;
    double* x = x_out;
    double* v = (double*)orc_xmalloc(sizeof(double) * n2);      /* complex */
    double* u = (double*)calloc(n2, sizeof(double));            /* uold = 0 */
    double* z = (double*)orc_xmalloc(sizeof(double) * n2);
    double* ym = (double*)orc_xmalloc(sizeof(double) * m2);
    const int Cin = s + (p->multi_level ? 1 : 0);
    double* w = (double*)orc_xmalloc(sizeof(double) * plane * Cin);
    double* wo = (double*)orc_xmalloc(sizeof(double) * n);
    if (!u) abort();

    if (x0) memcpy(x, x0, sizeof(double) * n2);                 /* x = param.X0 */
    else orc_adjoint(op, y, x);                                 /* param.X0 = F.adjoint(Y)  main_recon:292 */
    memcpy(v, x, sizeof(double) * n2);                          /* v = x */

    double ny = 0.0, ngt = 0.0;
    if (p->want_diag) {
        for (size_t i = 0; i < m2; ++i) ny += y[i] * y[i];
        ny = sqrt(ny);
        if (gt) { for (size_t i = 0; i < n2; ++i) ngt += gt[i] * gt[i]; ngt = sqrt(ngt); }
    }

    for (int i = 0; i < 4; ++i) g_stage_s[i] = 0.0;
    for (int it = 0; it < p->iters; ++it) {
        /* Step 1: x = argmin ||y - Ax||^2 + r ||x - (v - uold)||^2 */
        double t0 = omp_get_wtime();
        for (size_t i = 0; i < n2; ++i) z[i] = v[i] - u[i];
        int li = 0, lf = 0;
        if (p->solver == 0) orc_lsqr_xupdate(op, y, z, p->gamma, p->cg_tol, p->cg_maxit, x, &li, &lf, NULL);
        else orc_direct_xupdate(op, y, z, p->gamma, x);
        if (lsqr_iters_out) lsqr_iters_out[it] = li;
        double t1 = omp_get_wtime();
        g_stage_s[0] += t1 - t0;

        if (p->want_diag && diag_out) {                         /* :106-107 */
            orc_forward(op, x, ym);
            double a = 0.0, b = 0.0;
            for (size_t i = 0; i < m2; ++i) { double d = y[i] - ym[i]; a += d * d; }
            diag_out[2 * it] = sqrt(a) / ny;
            if (gt) { for (size_t i = 0; i < n2; ++i) { double d = gt[i] - x[i]; b += d * d; } diag_out[2 * it + 1] = sqrt(b) / ngt; }
            else diag_out[2 * it + 1] = NAN;
        }

        t0 = omp_get_wtime();
        g_stage_s[1] += t0 - t1;
        /* Step 2: v = real(x + uold); normalise to [0,1] over the whole stack (:174-184) */
        double lo = INFINITY, hi = -INFINITY;
        for (size_t i = 0; i < n; ++i) {
            double val = x[2 * i] + u[2 * i];
            w[i] = val;
            if (val < lo) lo = val;
            if (val > hi) hi = val;
        }
        const double range = hi - lo;
        for (size_t i = 0; i < n; ++i) w[i] = (w[i] - lo) / range;
        if (p->multi_level)                                     /* cat(3, v, noise_map) :132 */
            for (size_t i = 0; i < plane; ++i) w[n + i] = p->noise_std;
        t1 = omp_get_wtime();
        g_stage_s[3] += t1 - t0;
        orc_denoise(net, w, N, M, Cin, 1, p->residual_noise, wo);   /* net(v) */
        t0 = omp_get_wtime();
        g_stage_s[2] += t0 - t1;
        /* undo normalisation :138,187-192 ; v becomes real */
        for (size_t i = 0; i < n; ++i) { v[2 * i] = wo[i] * range + lo; v[2 * i + 1] = 0.0; }
        /* Step 3: uold = uold + x - v */
        for (size_t i = 0; i < n2; ++i) u[i] = u[i] + x[i] - v[i];
        g_stage_s[3] += omp_get_wtime() - t0;
    }
    free(v); free(u); free(z); free(ym); free(w); free(wo);
}

/* orc_lrtv.c -- CPU oracle for the LRTV solver option: the TV pieces of unlocbox the reference calls.
 * TEST INFRASTRUCTURE ONLY (see qmri_oracle.h).
 *
 * Restates, for a real column-major R x C image (FISTA_deep.m:66,75 stacks real and imaginary parts of the N x M x L
 * TSMI into one (2N) x (M L) image, so differences also run across the real/imaginary seam and across channel seams):
 *   gradient_op   unlocbox/utils/gradient_op.m:41-49   dx(i,j) = I(i+1,j) - I(i,j) (0 in the last row), dy likewise in j
 *   div_op        unlocbox/utils/div_op.m:42-55        the negative adjoint of gradient_op
 *   norm_tv       unlocbox/utils/norm_tv.m:45-55       sum sqrt(dx^2 + dy^2)
 *   prox_tv       unlocbox/prox/prox_tv.m:99-203       Beck-Teboulle dual FISTA: tol 10e-4 on the relative objective
 *                                                      change, maxit 200, weights [1 1], t = (1 + sqrt(4 t^2)) / 2 as written
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "qmri_oracle.h"

static inline double sol_at(const double* b, const double* r, const double* s, int R, int C, double gamma, int i, int j) {
    /* b - gamma * div_op(r, s) at (i, j) */
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

double orc_norm_tv(const double* I, int R, int C) {
    double tot = 0.0;
#pragma omp parallel for reduction(+ : tot) schedule(static)
    for (int j = 0; j < C; ++j) {
        double acc = 0.0;
        for (int i = 0; i < R; ++i) {
            const size_t p = (size_t)j * R + i;
            const double dx = (i < R - 1) ? I[p + 1] - I[p] : 0.0;
            const double dy = (j < C - 1) ? I[p + R] - I[p] : 0.0;
            acc += sqrt(dx * dx + dy * dy);
        }
        tot += acc;
    }
    return tot;
}

/* sol = prox_{gamma TV}(b).  Returns the iteration count (1-based index of the iteration whose `sol` is returned). */
int orc_prox_tv(const double* b, int R, int C, double gamma, double tol, int maxit, double* sol, double* obj_out) {
    const size_t n = (size_t)R * C;
    if (gamma == 0.0) { memcpy(sol, b, n * sizeof(double)); if (obj_out) *obj_out = 0.0; return 0; }     /* test_gamma */
    double* r = calloc(n, sizeof(double));
    double* s = calloc(n, sizeof(double));
    double* pold = calloc(n, sizeof(double));
    double* qold = calloc(n, sizeof(double));
    double told = 1.0, prev_obj = 0.0, obj = 0.0;
    int iter;
    for (iter = 1; iter <= maxit; ++iter) {
#pragma omp parallel for schedule(static)
        for (int j = 0; j < C; ++j)
            for (int i = 0; i < R; ++i) sol[(size_t)j * R + i] = sol_at(b, r, s, R, C, gamma, i, j);
        double fid = 0.0;
#pragma omp parallel for reduction(+ : fid) schedule(static)
        for (int j = 0; j < C; ++j) {
            double acc = 0.0;
            for (int i = 0; i < R; ++i) { const double d = b[(size_t)j * R + i] - sol[(size_t)j * R + i]; acc += d * d; }
            fid += acc;
        }
        obj = 0.5 * fid + gamma * orc_norm_tv(sol, R, C);
        const double rel_obj = fabs(obj - prev_obj) / obj;
        prev_obj = obj;
        if (rel_obj < tol) break;
        const double c = 1.0 / (8.0 * gamma);
        const double t = (1.0 + sqrt(4.0 * told * told)) / 2.0;
        const double mom = (told - 1.0) / t;
#pragma omp parallel for schedule(static)
        for (int j = 0; j < C; ++j)
            for (int i = 0; i < R; ++i) {
                const size_t p = (size_t)j * R + i;
                const double dx = (i < R - 1) ? sol[p + 1] - sol[p] : 0.0;
                const double dy = (j < C - 1) ? sol[p + R] - sol[p] : 0.0;
                const double rr = r[p] - c * dx, ss = s[p] - c * dy;
                const double w = fmax(1.0, sqrt(rr * rr + ss * ss));
                const double pp = rr / w, qq = ss / w;
                r[p] = pp + mom * (pp - pold[p]); pold[p] = pp;
                s[p] = qq + mom * (qq - qold[p]); qold[p] = qq;
            }
        told = t;
    }
    if (iter > maxit) iter = maxit;
    free(r); free(s); free(pold); free(qold);
    if (obj_out) *obj_out = obj;
    return iter;
}

/* ---- TSMI synthesis (main_synthesize_tsmis.m:54,82-100, mode 'real'): nearest look-up-table entry in (T1, T2) by exhaustive
 * search (knnsearch, Euclidean, first index among equal distances), X = D(I,:) * normD(I) * |PD| * sign(first channel). */
void orc_synthesize_tsmi(const double* qmap, int Npix, const float* D, const float* normD, const float* lut, int K, int s,
                         float* X, int32_t* idx) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < Npix; ++p) {
        const double q1 = qmap[p], q2 = qmap[(size_t)Npix + p];
        double best = INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const double d1 = q1 - (double)lut[k], d2 = q2 - (double)lut[(size_t)K + k];
            const double d = d1 * d1 + d2 * d2;
            if (d < best) { best = d; bi = k; }
        }
        if (idx) idx[p] = bi + 1;
        const float nd = normD[bi], pd = (float)fabs(qmap[(size_t)2 * Npix + p]);
        const float x0 = D[bi] * nd * pd;
        const float sg = (x0 > 0.f) ? 1.f : ((x0 < 0.f) ? -1.f : 0.f);
        for (int c = 0; c < s; ++c) X[(size_t)c * Npix + p] = D[(size_t)K * c + bi] * nd * pd * sg;
    }
}

/* mode 'complex' of main_synthesize_tsmis.m:100-103: X = (real(D(I,:)) .* normD(I)) .* qm(:,3), PD complex, no abs, no sign alignment;
 * stored as cat(3, real(X), imag(X)): X is Npix x 2s.  pd_imag may be NULL (real PD: the imaginary channels are zero). */
void orc_synthesize_tsmi_complex(const double* qmap, const double* pd_imag, int Npix, const float* D, const float* normD, const float* lut,
                                 int K, int s, float* X, int32_t* idx) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < Npix; ++p) {
        const double q1 = qmap[p], q2 = qmap[(size_t)Npix + p];
        double best = INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const double d1 = q1 - (double)lut[k], d2 = q2 - (double)lut[(size_t)K + k];
            const double d = d1 * d1 + d2 * d2;
            if (d < best) { best = d; bi = k; }
        }
        if (idx) idx[p] = bi + 1;
        const float nd = normD[bi], pr = (float)qmap[(size_t)2 * Npix + p], pi = pd_imag ? (float)pd_imag[p] : 0.f;
        for (int c = 0; c < s; ++c) {
            const float base = D[(size_t)K * c + bi] * nd;
            X[(size_t)c * Npix + p] = base * pr;
            X[(size_t)(s + c) * Npix + p] = base * pi;
        }
    }
}

/*
 * orc_lsqr.c -- the x-update of PnP-ADMM.  Test infrastructure only.
 *
 * Call site restated:   PnP_ADMM.m:102
 *     x = lsqr(@(z,flag)afun(z,opt,flag), [y(:); (v(:)-uold(:))*sqrt(r)], cg_tol, 100, [], [], x(:));
 * Stacked operator:     PnP_ADMM.m:153-171   B = [A; sqrt(r) I],  B' w = A' w1 + sqrt(r) w2.
 *
 * MATLAB's lsqr itself is MathWorks code whose source is NOT in the reference.  It is restated here from
 * its published description (Paige & Saunders 1982 bidiagonalisation; MATLAB doc page "lsqr": initial guess
 * handled through the residual u = b - B*x0, convergence declared when the normal-equation estimate
 * norm(B'r)/(norm(B)*norm(r)) <= tol or when norm(r) <= tol*norm(b), flag 1 at maxit, flag 3 after three
 * stagnating steps, `iter` = number of completed x updates).  The exact stop rule therefore is
 * PARITY UNPINNED against MATLAB; tests bound the ambiguity with the closed-form minimiser below
 * (any correct stop rule at tol=1e-4 lands within ~2.2e-4 relative of it).
 */
#include "orc_internal.h"
#include <float.h>

static double nrm2sq(const double* a, size_t n) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (size_t i = 0; i < n; ++i) s += a[i] * a[i];
    return s;
}

void orc_lsqr_xupdate(const orc_op* op, const double* y, const double* z, double r, double tol,
                      int maxit, double* x, int* iters, int* flag_out, double* relres) {
    const size_t n2 = 2 * (size_t)op->N * op->M * op->s;   /* doubles in an n-vector (complex) */
    const size_t m2 = 2 * (size_t)op->m;
    const double sr = sqrt(r);
    double* ut = (double*)orc_xmalloc(sizeof(double) * m2);   /* u(1:m)     */
    double* ub = (double*)orc_xmalloc(sizeof(double) * n2);   /* u(m+1:end) */
    double* v = (double*)orc_xmalloc(sizeof(double) * n2);
    double* d = (double*)calloc(n2, sizeof(double));
    double* tm = (double*)orc_xmalloc(sizeof(double) * m2);
    double* tn = (double*)orc_xmalloc(sizeof(double) * n2);
    if (!d) abort();

    /* b = [y; sqrt(r) z];  n2b = norm(b) */
    double n2b = 0.0;
    {
        double sy = nrm2sq(y, m2), sz = nrm2sq(z, n2);
        n2b = sqrt(sy + r * sz);
    }
    int flag = 1;
    const double tolb = tol * n2b;
    /* u = b - B*x0  (afun 'notransp', PnP_ADMM.m:160-162) */
    orc_forward(op, x, tm);
    for (size_t i = 0; i < m2; ++i) ut[i] = y[i] - tm[i];
    for (size_t i = 0; i < n2; ++i) ub[i] = z[i] * sr - x[i] * sr;
    double beta = sqrt(nrm2sq(ut, m2) + nrm2sq(ub, n2));
    double normr = beta;
    if (beta != 0.0) {
        for (size_t i = 0; i < m2; ++i) ut[i] /= beta;
        for (size_t i = 0; i < n2; ++i) ub[i] /= beta;
    }
    double c = 1.0, s = 0.0, phibar = beta;
    /* v = B'*u  (afun 'transp', PnP_ADMM.m:164-167) */
    orc_adjoint(op, ut, v);
    for (size_t i = 0; i < n2; ++i) v[i] += ub[i] * sr;
    double alpha = sqrt(nrm2sq(v, n2));
    if (alpha != 0.0)
        for (size_t i = 0; i < n2; ++i) v[i] /= alpha;
    double normar = alpha * beta;
    int iter = maxit;
    if (normar == 0.0 || n2b == 0.0) {
        /* x0 already solves the least-squares problem (or b = 0) */
        flag = 0; iter = 0;
        goto done;
    }
    {
        double norma = 0.0;
        int stag = 0;
        const int maxstagsteps = 3;
        for (int ii = 1; ii <= maxit; ++ii) {
            /* u = B*v - alpha*u */
            orc_forward(op, v, tm);
            for (size_t i = 0; i < m2; ++i) ut[i] = tm[i] - alpha * ut[i];
            for (size_t i = 0; i < n2; ++i) ub[i] = v[i] * sr - alpha * ub[i];
            beta = sqrt(nrm2sq(ut, m2) + nrm2sq(ub, n2));
            for (size_t i = 0; i < m2; ++i) ut[i] /= beta;
            for (size_t i = 0; i < n2; ++i) ub[i] /= beta;
            norma = sqrt(norma * norma + alpha * alpha + beta * beta);
            double thet = -s * alpha;
            double rhot = c * alpha;
            double rho = sqrt(rhot * rhot + beta * beta);
            c = rhot / rho;
            s = -beta / rho;
            double phi = c * phibar;
            if (phi == 0.0) stag = 1;
            phibar = s * phibar;
            for (size_t i = 0; i < n2; ++i) d[i] = (v[i] - thet * d[i]) / rho;
            double normd = sqrt(nrm2sq(d, n2));
            double normx = sqrt(nrm2sq(x, n2));
            if (fabs(phi) * normd < DBL_EPSILON * normx) stag++; else stag = 0;
            /* convergence in min ||b - B x|| : normal-equation estimate */
            if (normar / (norma * normr) <= tol) { flag = 0; iter = ii - 1; break; }
            /* convergence in B x = b */
            if (normr <= tolb) { flag = 0; iter = ii - 1; break; }
            if (stag >= maxstagsteps) { flag = 3; iter = ii - 1; break; }
            for (size_t i = 0; i < n2; ++i) x[i] += phi * d[i];
            normr = fabs(s) * normr;
            /* v = B'*u - beta*v */
            orc_adjoint(op, ut, tn);
            for (size_t i = 0; i < n2; ++i) v[i] = (tn[i] + ub[i] * sr) - beta * v[i];
            alpha = sqrt(nrm2sq(v, n2));
            for (size_t i = 0; i < n2; ++i) v[i] /= alpha;
            normar = alpha * fabs(s * phi);
        }
    }
done:
    if (iters) *iters = iter;
    if (flag_out) *flag_out = flag;
    if (relres) *relres = (n2b > 0.0) ? normr / n2b : 0.0;
    free(ut); free(ub); free(v); free(d); free(tm); free(tn);
}

/* ---- closed-form minimiser -------------------------------------------------------------------------
 * A'A = F' blockdiag_k(G_k) F with F the unitary 2-D DFT and G_k = sum_{t: k in Omega_t} V(t,:)' V(t,:),
 * so  xhat(k) = (G_k + r I)^-1 ( chat(k) + r zhat(k) ),  chat = F A' y.   
 * Not part of the reference; it bounds what any correct lsqr stop rule can return. */
static void chol_solve(int s, double* G, double* bre, double* bim) {
    /* in-place Cholesky G = L L^T (lower), then two triangular solves for real and imaginary parts */
    for (int j = 0; j < s; ++j) {
        double dsum = G[j * s + j];
        for (int k = 0; k < j; ++k) dsum -= G[j * s + k] * G[j * s + k];
        double ljj = sqrt(dsum);
        G[j * s + j] = ljj;
        for (int i = j + 1; i < s; ++i) {
            double v = G[i * s + j];
            for (int k = 0; k < j; ++k) v -= G[i * s + k] * G[j * s + k];
            G[i * s + j] = v / ljj;
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        double* b = pass ? bim : bre;
        for (int i = 0; i < s; ++i) {
            double v = b[i];
            for (int k = 0; k < i; ++k) v -= G[i * s + k] * b[k];
            b[i] = v / G[i * s + i];
        }
        for (int i = s - 1; i >= 0; --i) {
            double v = b[i];
            for (int k = i + 1; k < s; ++k) v -= G[k * s + i] * b[k];
            b[i] = v / G[i * s + i];
        }
    }
}

void orc_direct_xupdate(const orc_op* op, const double* y, const double* z, double r, double* x) {
    const int N = op->N, M = op->M, s = op->s, T = op->T;
    const size_t plane = (size_t)N * M, n = plane * s;
    cplx* c = (cplx*)orc_xmalloc(sizeof(cplx) * n);
    cplx* zh = (cplx*)orc_xmalloc(sizeof(cplx) * n);
    orc_adjoint(op, y, (double*)c);
    orc_fft2(N, M, s, -1, (const double*)c, (double*)c);
    orc_fft2(N, M, s, -1, z, (double*)zh);
    const double sc = 1.0 / sqrt((double)plane);       /* unitary scaling */
#pragma omp parallel for schedule(dynamic, 64)
    for (int k = 0; k < (int)plane; ++k) {
        double G[32 * 32], bre[32], bim[32];
        for (int a = 0; a < s * s; ++a) G[a] = 0.0;
        for (int e = op->k_ptr[k]; e < op->k_ptr[k + 1]; ++e) {
            int t = op->frame_of[op->k_meas[e]];
            for (int a = 0; a < s; ++a)
                for (int b = 0; b < s; ++b)
                    G[a * s + b] += op->V[t + (size_t)T * a] * op->V[t + (size_t)T * b];
        }
        for (int a = 0; a < s; ++a) {
            G[a * s + a] += r;
            bre[a] = (c[a * plane + k].re + r * zh[a * plane + k].re) * sc;
            bim[a] = (c[a * plane + k].im + r * zh[a * plane + k].im) * sc;
        }
        chol_solve(s, G, bre, bim);
        for (int a = 0; a < s; ++a) { zh[a * plane + k].re = bre[a]; zh[a * plane + k].im = bim[a]; }
    }
    orc_fft2(N, M, s, +1, (const double*)zh, x);
    const double sc2 = sqrt((double)plane);
    for (size_t i = 0; i < 2 * n; ++i) x[i] *= sc2;
    free(c); free(zh);
}

/*
 * orc_operator.c -- forward / adjoint operator.  Test infrastructure only.
 *
 * Restated from
 *   main_recon_tsmis_FFT.m:228  F.forward = @(x) P.for(reshape(fft2(x),[],1))/sqrt(N*M)
 *   main_recon_tsmis_FFT.m:229  F.adjoint = @(x) ifft2(reshape(P.adj(x),N,M,[]))*sqrt(N*M)
 *   setup_subsampling_spiralgrided.m:36-42 / setup_subsampling_epi.m:31-35:
 *       P = [P; tmp * kron(conj(V(i,:)), speye(N*M))]   -> row (t,k), column (c,k) holds conj(V(t,c))
 * V is real (main_recon_tsmis_FFT.m:129) so conj(V) = V.
 */
#include "orc_internal.h"

orc_op* orc_op_create(int N, int M, int s, int T, const double* V, const int32_t* frame_ptr,
                      const int32_t* kidx) {
    orc_op* op = (orc_op*)orc_xmalloc(sizeof(orc_op));
    op->N = N; op->M = M; op->s = s; op->T = T;
    op->m = frame_ptr[T];
    op->V = (double*)orc_xmalloc(sizeof(double) * T * s);
    memcpy(op->V, V, sizeof(double) * T * s);
    op->frame_ptr = (int32_t*)orc_xmalloc(sizeof(int32_t) * (T + 1));
    memcpy(op->frame_ptr, frame_ptr, sizeof(int32_t) * (T + 1));
    op->kidx = (int32_t*)orc_xmalloc(sizeof(int32_t) * op->m);
    memcpy(op->kidx, kidx, sizeof(int32_t) * op->m);
    op->frame_of = (int32_t*)orc_xmalloc(sizeof(int32_t) * op->m);
    for (int t = 0; t < T; ++t)
        for (int i = frame_ptr[t]; i < frame_ptr[t + 1]; ++i) op->frame_of[i] = t;
    int NM = N * M;
    op->k_ptr = (int32_t*)calloc((size_t)NM + 1, sizeof(int32_t));
    op->k_meas = (int32_t*)orc_xmalloc(sizeof(int32_t) * op->m);
    for (int i = 0; i < op->m; ++i) op->k_ptr[kidx[i] + 1]++;
    for (int k = 0; k < NM; ++k) op->k_ptr[k + 1] += op->k_ptr[k];
    int32_t* fill = (int32_t*)orc_xmalloc(sizeof(int32_t) * NM);
    memcpy(fill, op->k_ptr, sizeof(int32_t) * NM);
    for (int i = 0; i < op->m; ++i) op->k_meas[fill[kidx[i]]++] = i;
    free(fill);
    return op;
}

void orc_op_destroy(orc_op* op) {
    if (!op) return;
    free(op->V); free(op->frame_ptr); free(op->kidx); free(op->frame_of);
    free(op->k_ptr); free(op->k_meas); free(op);
}

int orc_op_m(const orc_op* op) { return op->m; }

void orc_forward(const orc_op* op, const double* x, double* y) {
    const int N = op->N, M = op->M, s = op->s, T = op->T;
    const size_t plane = (size_t)N * M;
    cplx* X = (cplx*)orc_xmalloc(sizeof(cplx) * plane * s);
    orc_fft2(N, M, s, -1, x, (double*)X);                       /* fft2(x) */
    const double sc = 1.0 / sqrt((double)N * (double)M);        /* /sqrt(N*M) */
    cplx* Y = (cplx*)y;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < op->m; ++i) {
        int t = op->frame_of[i], k = op->kidx[i];
        double re = 0.0, im = 0.0;
        for (int c = 0; c < s; ++c) {                            /* P*x: sum_c conj(V(t,c)) * Xhat_c[k] */
            double v = op->V[t + (size_t)T * c];
            re += v * X[c * plane + k].re;
            im += v * X[c * plane + k].im;
        }
        Y[i].re = re * sc; Y[i].im = im * sc;
    }
    free(X);
}

void orc_adjoint(const orc_op* op, const double* y, double* x) {
    const int N = op->N, M = op->M, s = op->s, T = op->T;
    const size_t plane = (size_t)N * M;
    const cplx* Y = (const cplx*)y;
    cplx* Z = (cplx*)calloc(plane * s, sizeof(cplx));
    if (!Z) abort();
#pragma omp parallel for schedule(static)
    for (int k = 0; k < (int)plane; ++k) {                       /* P'*y: Zhat_c[k] += V(t,c) * y[(t,k)] */
        for (int e = op->k_ptr[k]; e < op->k_ptr[k + 1]; ++e) {
            int i = op->k_meas[e];
            int t = op->frame_of[i];
            for (int c = 0; c < s; ++c) {
                double v = op->V[t + (size_t)T * c];
                Z[c * plane + k].re += v * Y[i].re;
                Z[c * plane + k].im += v * Y[i].im;
            }
        }
    }
    orc_fft2(N, M, s, +1, (const double*)Z, x);                  /* ifft2 (carries 1/(NM)) */
    const double sc = sqrt((double)N * (double)M);               /* *sqrt(N*M) */
    for (size_t i = 0; i < 2 * plane * s; ++i) x[i] *= sc;
    free(Z);
}

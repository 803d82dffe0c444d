/*
 * orc_fft.c -- mixed-radix complex DFT used by the operator restatement.  Test infrastructure only.
 *
 * Stands in for MATLAB's fft2/ifft2 (main_recon_tsmis_FFT.m:228-229; MathWorks/FFTW, source not in the
 * reference).  Definition restated from the documentation:
 *     fft :  X[k] = sum_n x[n] exp(-2*pi*i*n*k/N)          (sign = -1, unnormalised)
 *     ifft:  x[n] = (1/N) sum_k X[k] exp(+2*pi*i*n*k/N)    (sign = +1; the 1/N is applied by the caller)
 * Algorithm: recursive decimation-in-time over the prime factors of N (224 = 2^5 * 7), twiddles from
 * one table of cos/sin evaluated in double.
 */
#include "orc_internal.h"

typedef struct {
    int n;
    cplx* tw;      /* tw[k] = exp(sign*2*pi*i*k/n), k = 0..n-1 */
} fft_plan;

static int smallest_factor(int n) {
    for (int p = 2; p * p <= n; ++p)
        if (n % p == 0) return p;
    return n;
}

static inline cplx cmul(cplx a, cplx b) {
    cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}

/* out[0..n) = DFT of in[0], in[stride], ...; tw table is for the ROOT size N0, tstep = N0/n. */
static void fft_rec(int n, const cplx* in, ptrdiff_t stride, cplx* out, const cplx* tw, int N0, cplx* scratch) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = smallest_factor(n);
    int q = n / p;
    for (int j = 0; j < p; ++j)
        fft_rec(q, in + j * stride, stride * p, out + (ptrdiff_t)j * q, tw, N0, scratch);
    int tstep = N0 / n;     /* W_n^a = tw[a*tstep] */
    int pstep = N0 / p;     /* W_p^a = tw[a*pstep] */
    for (int k = 0; k < q; ++k) {
        cplx t[16];
        cplx* tt = (p <= 16) ? t : scratch;
        for (int j = 0; j < p; ++j)
            tt[j] = cmul(out[(ptrdiff_t)j * q + k], tw[((long)j * k % n) * tstep]);
        for (int l = 0; l < p; ++l) {
            cplx acc = tt[0];
            for (int j = 1; j < p; ++j) {
                cplx w = tw[((long)j * l % p) * pstep];
                acc.re += tt[j].re * w.re - tt[j].im * w.im;
                acc.im += tt[j].re * w.im + tt[j].im * w.re;
            }
            /* X[k + l*q] */
            if (p <= 16) scratch[l] = acc; else scratch[p + l] = acc;
        }
        for (int l = 0; l < p; ++l)
            out[(ptrdiff_t)l * q + k] = (p <= 16) ? scratch[l] : scratch[p + l];
    }
}

void orc_fft_lines(int n, int howmany, int sign, const cplx* in, ptrdiff_t in_dist, ptrdiff_t in_stride,
                   cplx* out, ptrdiff_t out_dist, ptrdiff_t out_stride) {
    const double PI = 3.14159265358979323846;
    cplx* tw = (cplx*)orc_xmalloc(sizeof(cplx) * n);
    for (int k = 0; k < n; ++k) {
        double a = 2.0 * PI * (double)k / (double)n;
        tw[k].re = cos(a);
        tw[k].im = (sign < 0) ? -sin(a) : sin(a);
    }
#pragma omp parallel
    {
        cplx* buf = (cplx*)orc_xmalloc(sizeof(cplx) * n);
        cplx* scratch = (cplx*)orc_xmalloc(sizeof(cplx) * (2 * n + 32));
#pragma omp for schedule(static)
        for (int l = 0; l < howmany; ++l) {
            fft_rec(n, in + (ptrdiff_t)l * in_dist, in_stride, buf, tw, n, scratch);
            cplx* o = out + (ptrdiff_t)l * out_dist;
            for (int k = 0; k < n; ++k) o[(ptrdiff_t)k * out_stride] = buf[k];
        }
        free(buf); free(scratch);
    }
    free(tw);
}

/* 2-D DFT of each of s channels of an N x M column-major complex array; sign=+1 includes 1/(N*M). */
void orc_fft2(int N, int M, int s, int sign, const double* in_, double* out_) {
    const cplx* in = (const cplx*)in_;
    cplx* out = (cplx*)out_;
    size_t plane = (size_t)N * M;
    cplx* tmp = (cplx*)orc_xmalloc(sizeof(cplx) * plane * s);
    /* along rows index (dimension 1, contiguous, length N): M*s lines */
    orc_fft_lines(N, M * s, sign, in, N, 1, tmp, N, 1);
    /* along columns index (dimension 2, stride N, length M): for each channel, N lines */
    for (int c = 0; c < s; ++c)
        orc_fft_lines(M, N, sign, tmp + c * plane, 1, N, out + c * plane, 1, N);
    if (sign > 0) {
        double sc = 1.0 / ((double)N * (double)M);
        for (size_t i = 0; i < plane * s; ++i) { out[i].re *= sc; out[i].im *= sc; }
    }
    free(tmp);
}

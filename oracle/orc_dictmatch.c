/*
 * orc_dictmatch.c -- MRF dictionary template match.  Test infrastructure only.
 *
 * Restated from main_files/dictionary_matching/mrf_dtm_cpu.m:
 *   :50-54   x = reshape(data.X,[N,T]); mask forced to all ones; x = single(x)
 *   :74      blockSize = min(max(floor(par.fp.blockSize/K),1),Npix)
 *   :91      ip = dict.D * ctranspose(x(cind,:))          -> ip(j,p) = sum_c D(j,c) * conj(x(p,c))
 *   :92      [mt,dm] = max(abs(ip),[],1)                   -> first index wins ties
 *   :94-96   pd = ip(dm); Xfit = pd .* D(dm,:); pd = pd ./ normD(dm)
 *   :136-160 qmap = lut(dm,:) with NaN -> 0; pd; mt; dm (1-based)
 * The blocked loop only bounds MATLAB's temporary; results do not depend on it, so block_size is accepted
 * and the same per-pixel arithmetic is applied to every pixel.
 * Arithmetic made explicit (MATLAB's BLAS order is unspecified): ip accumulates c = 0..s-1 as an fmaf chain,
 * abs(ip) = sqrtf(fmaf(im,im,re*re)) in single, and the maximum is taken over those single-precision magnitudes as
 * max(abs(ip)) does (:92): two atoms whose |ip|^2 differ in the last bits but whose magnitudes round to the same single
 * are a tie, and the first index wins.  The same chain is what a f32 MFMA executes, so the product's atom indices can
 * be compared bit-exactly.
 */
#include "orc_internal.h"

void orc_dict_match(const double* X, int Npix, int s, const float* D, const float* normD,
                    const float* lut, int K, int Q, double block_size, float* qmap, float* pd,
                    float* mt, int32_t* dm, float* Xfit) {
    (void)block_size;
    /* s is whatever size(data.X, end) is (:41-50): 10 compressed channels in the shipped script, T uncompressed frames otherwise.
     * Speed only (the arithmetic per (atom, pixel) is the chain above, untouched): D is read through a row-major copy so that a chain
     * walks contiguous memory, and four atoms' chains run side by side to keep the FMA units busy. */
    float* Dt = (float*)malloc(sizeof(float) * (size_t)K * (size_t)(s > 0 ? s : 1));
    for (int c = 0; c < s; ++c)
        for (int j = 0; j < K; ++j) Dt[(size_t)j * s + c] = D[(size_t)j + (size_t)K * c];
#pragma omp parallel
  {
    float* xr = (float*)malloc(sizeof(float) * (size_t)(s > 0 ? s : 1));
    float* xi = (float*)malloc(sizeof(float) * (size_t)(s > 0 ? s : 1));
#pragma omp for schedule(static)
    for (int p = 0; p < Npix; ++p) {
        for (int c = 0; c < s; ++c) {
            xr[c] = (float)X[2 * ((size_t)p + (size_t)Npix * c)];        /* single(x) :54 */
            xi[c] = -(float)X[2 * ((size_t)p + (size_t)Npix * c) + 1];   /* conj(x) :91 */
        }
        float best = -1.0f, bre = 0.f, bim = 0.f;
        int bj = 0;
        for (int j0 = 0; j0 < K; j0 += 4) {
            const int nj = (K - j0 < 4) ? K - j0 : 4;
            float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
            if (nj == 4) {
                const float *d0 = Dt + (size_t)j0 * s, *d1 = d0 + s, *d2 = d1 + s, *d3 = d2 + s;
                for (int c = 0; c < s; ++c) {
                    re[0] = fmaf(d0[c], xr[c], re[0]); im[0] = fmaf(d0[c], xi[c], im[0]);
                    re[1] = fmaf(d1[c], xr[c], re[1]); im[1] = fmaf(d1[c], xi[c], im[1]);
                    re[2] = fmaf(d2[c], xr[c], re[2]); im[2] = fmaf(d2[c], xi[c], im[2]);
                    re[3] = fmaf(d3[c], xr[c], re[3]); im[3] = fmaf(d3[c], xi[c], im[3]);
                }
            } else {
                for (int q = 0; q < nj; ++q)
                    for (int c = 0; c < s; ++c) {
                        const float d = Dt[(size_t)(j0 + q) * s + c];
                        re[q] = fmaf(d, xr[c], re[q]);
                        im[q] = fmaf(d, xi[c], im[q]);
                    }
            }
            for (int q = 0; q < nj; ++q) {
                const float mag = sqrtf(fmaf(im[q], im[q], re[q] * re[q]));                 /* abs(ip) :92 */
                if (mag > best) { best = mag; bj = j0 + q; bre = re[q]; bim = im[q]; }      /* strict: the first index wins ties */
            }
        }
        if (dm) dm[p] = bj + 1;
        if (mt) mt[p] = best;
        if (pd) { pd[2 * p] = bre / normD[bj]; pd[2 * p + 1] = bim / normD[bj]; }
        if (qmap)
            for (int q = 0; q < Q; ++q) {
                float v = lut[(size_t)bj + (size_t)K * q];
                qmap[(size_t)p + (size_t)Npix * q] = isnan(v) ? 0.f : v;
            }
        if (Xfit)
            for (int c = 0; c < s; ++c) {
                const float d = Dt[(size_t)bj * s + c];
                Xfit[2 * ((size_t)p + (size_t)Npix * c)] = bre * d;                         /* ip(dm) .* D(dm,:) :95 */
                Xfit[2 * ((size_t)p + (size_t)Npix * c) + 1] = bim * d;
            }
    }
    free(xr); free(xi);
  }
    free(Dt);
}

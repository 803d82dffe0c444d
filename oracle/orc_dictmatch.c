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
#pragma omp parallel for schedule(static)
    for (int p = 0; p < Npix; ++p) {
        float xr[64], xi[64];
        for (int c = 0; c < s; ++c) {
            xr[c] = (float)X[2 * ((size_t)p + (size_t)Npix * c)];        /* single(x) :54 */
            xi[c] = (float)X[2 * ((size_t)p + (size_t)Npix * c) + 1];
        }
        float best = -1.0f, bre = 0.f, bim = 0.f;
        int bj = 0;
        for (int j = 0; j < K; ++j) {
            float re = 0.f, im = 0.f;
            for (int c = 0; c < s; ++c) {
                const float d = D[(size_t)j + (size_t)K * c];
                re = fmaf(d, xr[c], re);
                im = fmaf(d, -xi[c], im);                                /* conj(x) */
            }
            const float mag = sqrtf(fmaf(im, im, re * re));             /* abs(ip) :92 */
            if (mag > best) { best = mag; bj = j; bre = re; bim = im; }  /* strict: the first index wins ties */
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
                const float d = D[(size_t)bj + (size_t)K * c];
                Xfit[2 * ((size_t)p + (size_t)Npix * c)] = bre * d;
                Xfit[2 * ((size_t)p + (size_t)Npix * c) + 1] = bim * d;
            }
    }
}

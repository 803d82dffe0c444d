/* Internal declarations shared by the oracle translation units. Test infrastructure only. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "qmri_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } cplx;

struct orc_op {
    int N, M, s, T, m;
    double* V;            /* T x s column-major, real (main_recon_tsmis_FFT.m:129) */
    int32_t* frame_ptr;   /* T+1 */
    int32_t* kidx;        /* m, 0-based column-major k index */
    int32_t* frame_of;    /* m: frame of each measurement */
    /* inverse lists per k (for the closed-form solve): CSR over k */
    int32_t* k_ptr;       /* N*M+1 */
    int32_t* k_meas;      /* m: measurement indices hitting k */
};

/* orc_fft.c */
void orc_fft_lines(int n, int howmany, int sign, const cplx* in, ptrdiff_t in_dist, ptrdiff_t in_stride,
                   cplx* out, ptrdiff_t out_dist, ptrdiff_t out_stride);

static inline void* orc_xmalloc(size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p) abort();
    return p;
}

#endif

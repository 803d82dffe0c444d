/*
 * orc_masks.c -- sampling-mask builders.  Test infrastructure only.
 *
 * Restated from
 *   main_files/subsampling_patterns/setup_subsampling_spiralgrided.m:7-34
 *   main_files/subsampling_patterns/setup_subsampling_epi.m:20-33
 * MATLAB built-ins restated from documentation: linspace, round (half away from zero == C round()),
 * fftshift (swap halves in both dims), find (ascending column-major order).
 * Parity unpinned against MATLAB itself (no MATLAB output exists in this pipeline); cross-checked bit for bit against
 * the independent literal restatement of the same .m lines (tools/gen_matlab_rows.py -> tests/golden/matlab_rows_*.npz,
 * tests/test_oracle_matlab_rows.py), which is builder-written code, not reference output.
 */
#include "orc_internal.h"

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

int orc_spiral_mask(int N, int S, int T, int32_t* frame_ptr, int32_t* kidx, int cap) {
    const double PI = 3.14159265358979323846;
    const double delta = PI / 180.0 * 7.5;                    /* :7 */
    double* theta = (double*)orc_xmalloc(sizeof(double) * S);
    double* rr = (double*)orc_xmalloc(sizeof(double) * S);
    unsigned char* B = (unsigned char*)orc_xmalloc((size_t)N * N);
    /* t = linspace(0, 2*pi, S)  (:16): d1 + (0:S-1)*(d2-d1)/(S-1), last element exactly d2 */
    double rmin = INFINITY, rmax = -INFINITY;
    for (int j = 0; j < S; ++j) {
        double t = (S > 1) ? (double)j * (2.0 * PI) / (double)(S - 1) : 2.0 * PI;
        if (j == S - 1) t = 2.0 * PI;
        theta[j] = 8.0 * t;                                   /* :17 */
        rr[j] = pow(1.05, theta[j]);                          /* :18 */
        if (rr[j] < rmin) rmin = rr[j];
        if (rr[j] > rmax) rmax = rr[j];
    }
    for (int j = 0; j < S; ++j) rr[j] = (rr[j] - rmin) / (rmax - rmin);   /* :19 */

    int m = 0, overflow = 0;
    const int half = N / 2;
    for (int i = 0; i < T; ++i) {
        frame_ptr[i] = m;
        memset(B, 0, (size_t)N * N);
        for (int j = 0; j < S; ++j) {
            double cx = rr[j] * cos(theta[j] + (double)i * delta);        /* :25 */
            double cy = rr[j] * sin(theta[j] + (double)i * delta);
            double gx = round(cx * N / 2.0) + N / 2.0 + 1.0;               /* :28 (1-based row)  */
            double gy = round(cy * N / 2.0) + N / 2.0 + 1.0;               /*     (1-based col)  */
            if (gx > N) gx = N;                                            /* :29 */
            if (gy > N) gy = N;
            int r0 = (int)gx - 1, c0 = (int)gy - 1;                        /* ind = cx + N*(cy-1) :30 */
            /* fftshift (:33): element (r,c) moves to ((r+N/2) mod N, (c+N/2) mod N) for even N;
               for odd N MATLAB shifts by floor(N/2), same formula. */
            int r1 = (r0 + half) % N, c1 = (c0 + half) % N;
            B[(size_t)c1 * N + r1] = 1;                                    /* temp(ind) = 1 :32 */
        }
        for (int k = 0; k < N * N; ++k) {                                  /* find(temp==1) :34 */
            if (B[k]) {
                if (m < cap) kidx[m] = k; else overflow = 1;
                ++m;
            }
        }
    }
    frame_ptr[T] = m;
    free(theta); free(rr); free(B);
    return overflow ? -m : m;
}

int orc_epi_mask(int N, int M, double percentage, int T, int32_t* frame_ptr, int32_t* kidx, int cap) {
    int step = (int)round(1.0 / percentage);          /* :20 */
    int no_of_steps = N / step;                       /* floor(N/step) :21 */
    int* rows = (int*)orc_xmalloc(sizeof(int) * (no_of_steps > 0 ? no_of_steps : 1));
    int32_t* srt = (int32_t*)orc_xmalloc(sizeof(int32_t) * (no_of_steps > 0 ? no_of_steps : 1));
    /* comb(1:step:step*nb_meas/M) = 1  (:25) -> 0-based rows 0, step, 2*step, ... */
    int nrows = 0;
    for (int r = 0; r < step * no_of_steps; r += step) rows[nrows++] = r;
    int m = 0, overflow = 0;
    for (int i = 0; i < T; ++i) {
        frame_ptr[i] = m;
        /* comb = comb([N,1:N-1])  (:28): cyclic shift by one row BEFORE use, first frame included */
        for (int q = 0; q < nrows; ++q) rows[q] = (rows[q] + 1) % N;
        for (int q = 0; q < nrows; ++q) srt[q] = rows[q];
        qsort(srt, nrows, sizeof(int32_t), cmp_i32);
        /* template = comb*ones(1,M); find(template(:)==1)  (:29-30): ascending column-major */
        for (int c = 0; c < M; ++c)
            for (int q = 0; q < nrows; ++q) {
                if (m < cap) kidx[m] = srt[q] + N * c; else overflow = 1;
                ++m;
            }
    }
    frame_ptr[T] = m;
    free(rows); free(srt);
    return overflow ? -m : m;
}

// fft_codelets.h -- in-register forward DFT codelets (double precision) for the data-consistency kernels.
//
// These implement the per-channel fft2/ifft2 of the reference's forward operator
// (main_recon_tsmis_FFT.m:228-229) for N = R1*R2 with R1, R2 in {2,4,7,8,14,16}; 224 = 16 * 14.
// Sign convention: X[k] = sum_n x[n] exp(-2*pi*i*n*k/R).  The inverse is obtained by conjugating on
// the way in and out.  The header compiles for host and device (QMRI_HD) so that tests/ can check the
// codelets against a naive DFT with g++ on a machine without a GPU.
#pragma once

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define QMRI_HD __host__ __device__ __forceinline__
typedef double2 cd;
#else
#define QMRI_HD inline
struct cd { double x, y; };
#endif

namespace qfft {

QMRI_HD cd mk(double x, double y) { cd r; r.x = x; r.y = y; return r; }
QMRI_HD cd add(cd a, cd b) { return mk(a.x + b.x, a.y + b.y); }
QMRI_HD cd sub(cd a, cd b) { return mk(a.x - b.x, a.y - b.y); }
QMRI_HD cd mul(cd a, cd b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
QMRI_HD cd mulmi(cd a) { return mk(a.y, -a.x); }          // a * (-i)
QMRI_HD cd scale(cd a, double s) { return mk(a.x * s, a.y * s); }
QMRI_HD cd conj(cd a) { return mk(a.x, -a.y); }
// a * (c - i s)  == a * exp(-i theta) with c = cos theta, s = sin theta
QMRI_HD cd mulw(cd a, double c, double s) { return mk(a.x * c + a.y * s, a.y * c - a.x * s); }

template <int R> struct Dft;

template <> struct Dft<1> { static QMRI_HD void run(cd*) {} };

template <> struct Dft<2> {
    static QMRI_HD void run(cd* a) {
        cd t = a[0];
        a[0] = add(t, a[1]);
        a[1] = sub(t, a[1]);
    }
};

template <> struct Dft<4> {
    static QMRI_HD void run(cd* a) {
        cd t0 = add(a[0], a[2]), t1 = sub(a[0], a[2]);
        cd t2 = add(a[1], a[3]), t3 = mulmi(sub(a[1], a[3]));
        a[0] = add(t0, t2); a[2] = sub(t0, t2);
        a[1] = add(t1, t3); a[3] = sub(t1, t3);
    }
};

template <> struct Dft<8> {
    static QMRI_HD void run(cd* a) {
        const double h = 0.7071067811865476;
        cd e[4] = { a[0], a[2], a[4], a[6] };
        cd o[4] = { a[1], a[3], a[5], a[7] };
        Dft<4>::run(e);
        Dft<4>::run(o);
        o[1] = mulw(o[1], h, h);          // w8^1 = (1 - i)/sqrt2
        o[2] = mulmi(o[2]);               // w8^2 = -i
        o[3] = mulw(o[3], -h, h);         // w8^3 = (-1 - i)/sqrt2
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = add(e[k], o[k]); a[k + 4] = sub(e[k], o[k]); }
    }
};

template <> struct Dft<16> {
    static QMRI_HD void run(cd* a) {
        const double c1 = 0.9238795325112867, s1 = 0.3826834323650898, h = 0.7071067811865476;
        cd e[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { e[k] = a[2 * k]; o[k] = a[2 * k + 1]; }
        Dft<8>::run(e);
        Dft<8>::run(o);
        o[1] = mulw(o[1], c1, s1);
        o[2] = mulw(o[2], h, h);
        o[3] = mulw(o[3], s1, c1);
        o[4] = mulmi(o[4]);
        o[5] = mulw(o[5], -s1, c1);
        o[6] = mulw(o[6], -h, h);
        o[7] = mulw(o[7], -c1, s1);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = add(e[k], o[k]); a[k + 8] = sub(e[k], o[k]); }
    }
};

template <> struct Dft<7> {
    // X_k = a0 + sum_j p_j cos(2 pi j k/7) - i sum_j q_j sin(2 pi j k/7),  p_j = a_j + a_{7-j}, q_j = a_j - a_{7-j}
    static QMRI_HD void run(cd* a) {
        const double C1 = 0.6234898018587336, C2 = -0.22252093395631434, C3 = -0.900968867902419;
        const double S1 = 0.7818314824680298, S2 = 0.9749279121818236, S3 = 0.43388373911755823;
        cd p1 = add(a[1], a[6]), p2 = add(a[2], a[5]), p3 = add(a[3], a[4]);
        cd q1 = sub(a[1], a[6]), q2 = sub(a[2], a[5]), q3 = sub(a[3], a[4]);
        cd a0 = a[0];
        cd r1 = mk(a0.x + C1 * p1.x + C2 * p2.x + C3 * p3.x, a0.y + C1 * p1.y + C2 * p2.y + C3 * p3.y);
        cd r2 = mk(a0.x + C2 * p1.x + C3 * p2.x + C1 * p3.x, a0.y + C2 * p1.y + C3 * p2.y + C1 * p3.y);
        cd r3 = mk(a0.x + C3 * p1.x + C1 * p2.x + C2 * p3.x, a0.y + C3 * p1.y + C1 * p2.y + C2 * p3.y);
        cd i1 = mk(S1 * q1.x + S2 * q2.x + S3 * q3.x, S1 * q1.y + S2 * q2.y + S3 * q3.y);
        cd i2 = mk(S2 * q1.x - S3 * q2.x - S1 * q3.x, S2 * q1.y - S3 * q2.y - S1 * q3.y);
        cd i3 = mk(S3 * q1.x - S1 * q2.x + S2 * q3.x, S3 * q1.y - S1 * q2.y + S2 * q3.y);
        a[0] = mk(a0.x + p1.x + p2.x + p3.x, a0.y + p1.y + p2.y + p3.y);
        // X_k = r_k - i*i_k ; X_{7-k} = r_k + i*i_k    with -i*(x+iy) = (y, -x)
        a[1] = mk(r1.x + i1.y, r1.y - i1.x); a[6] = mk(r1.x - i1.y, r1.y + i1.x);
        a[2] = mk(r2.x + i2.y, r2.y - i2.x); a[5] = mk(r2.x - i2.y, r2.y + i2.x);
        a[3] = mk(r3.x + i3.y, r3.y - i3.x); a[4] = mk(r3.x - i3.y, r3.y + i3.x);
    }
};

template <> struct Dft<14> {
    static QMRI_HD void run(cd* a) {
        const double c[7] = { 1.0, 0.9009688679024191, 0.6234898018587336, 0.22252093395631445,
                              -0.22252093395631434, -0.6234898018587335, -0.900968867902419 };
        const double s[7] = { 0.0, 0.4338837391175581, 0.7818314824680298, 0.9749279121818236,
                              0.9749279121818236, 0.7818314824680299, 0.43388373911755823 };
        cd e[7], o[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { e[k] = a[2 * k]; o[k] = a[2 * k + 1]; }
        Dft<7>::run(e);
        Dft<7>::run(o);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            cd t = (k == 0) ? o[0] : mulw(o[k], c[k], s[k]);
            a[k] = add(e[k], t);
            a[k + 7] = sub(e[k], t);
        }
    }
};

// Index maps of the two-step (R1 x R2) transform, N = R1*R2, input index n = R2*n1 + n2, output k = k1 + R1*k2:
//   step 1 (thread n2): a[n1] = x[R2*n1 + n2]; DFT_R1; a[k1] *= W_N^(n2*k1); store S[n2][k1]
//   step 2 (thread k1): b[n2] = S[n2][k1];     DFT_R2; X[k1 + R1*k2] = b[k2]
// S is kept at pitch R1+1 so that both the strided writes and the row reads are bank-conflict free.
template <int R1, int R2> struct Plan {
    static constexpr int N = R1 * R2;
    static constexpr int SP = R1 + 1;                 // pitch of the intermediate S[n2][.]
    static constexpr int LINE = ((R2 * SP > N ? R2 * SP : N) | 1);   // LDS complex elements reserved per line (odd)
};

}  // namespace qfft

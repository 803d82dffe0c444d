// Host check of csrc/fft_codelets.h: every codelet and the two-step index maps against a naive DFT.
// Built and run by tests/test_fft_codelets.py with g++ (no GPU needed).
#include "fft_codelets.h"
#include <cmath>
#include <cstdio>
#include <vector>

using namespace qfft;

static double naive_err(const std::vector<cd>& x, const std::vector<cd>& X) {
    const int n = (int)x.size();
    double err = 0, ref = 0;
    for (int k = 0; k < n; ++k) {
        double re = 0, im = 0;
        for (int j = 0; j < n; ++j) {
            double a = -2.0 * M_PI * (double)((long)j * k % n) / n;
            re += x[j].x * cos(a) - x[j].y * sin(a);
            im += x[j].x * sin(a) + x[j].y * cos(a);
        }
        err = fmax(err, hypot(re - X[k].x, im - X[k].y));
        ref = fmax(ref, hypot(re, im));
    }
    return err / ref;
}

template <int R> static double test_codelet() {
    std::vector<cd> x(R), X(R);
    for (int i = 0; i < R; ++i) x[i] = mk(sin(1.0 + 3.7 * i), cos(0.3 + 2.1 * i * i));
    X = x;
    Dft<R>::run(X.data());
    return naive_err(x, X);
}

template <int R1, int R2> static double test_plan() {
    typedef Plan<R1, R2> P;
    const int N = P::N;
    std::vector<cd> x(N), X(N), S(P::LINE), tw(N);
    for (int i = 0; i < N; ++i) {
        x[i] = mk(sin(0.5 + 1.3 * i) + 0.01 * i, cos(0.1 + 0.7 * i));
        tw[i] = mk(cos(2.0 * M_PI * i / N), -sin(2.0 * M_PI * i / N));
    }
    for (int n2 = 0; n2 < R2; ++n2) {            // step 1
        cd a[R1];
        for (int n1 = 0; n1 < R1; ++n1) a[n1] = x[R2 * n1 + n2];
        Dft<R1>::run(a);
        for (int k1 = 0; k1 < R1; ++k1) S[P::SP * n2 + k1] = mul(a[k1], tw[(n2 * k1) % N]);
    }
    for (int k1 = 0; k1 < R1; ++k1) {            // step 2
        cd b[R2];
        for (int n2 = 0; n2 < R2; ++n2) b[n2] = S[P::SP * n2 + k1];
        Dft<R2>::run(b);
        for (int k2 = 0; k2 < R2; ++k2) X[k1 + R1 * k2] = b[k2];
    }
    return naive_err(x, X);
}

int main() {
    double e;
    int bad = 0;
#define CHECK(name, expr) e = (expr); printf("%-14s %.3e\n", name, e); if (!(e < 1e-13)) bad++;
    CHECK("dft2", test_codelet<2>());
    CHECK("dft4", test_codelet<4>());
    CHECK("dft7", test_codelet<7>());
    CHECK("dft8", test_codelet<8>());
    CHECK("dft14", test_codelet<14>());
    CHECK("dft16", test_codelet<16>());
    CHECK("plan16x14", (test_plan<16, 14>()));
    CHECK("plan8x4", (test_plan<8, 4>()));
    CHECK("plan8x8", (test_plan<8, 8>()));
    CHECK("plan16x8", (test_plan<16, 8>()));
    CHECK("plan16x16", (test_plan<16, 16>()));
    CHECK("plan8x7", (test_plan<8, 7>()));
    return bad;
}

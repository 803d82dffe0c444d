// host_asan_driver.cpp -- drives the HOST code of libqmri (built with -fsanitize=address,undefined and no device code: `make -C
// qmri_pnp_recon_poc_amd/csrc asan-host`) on a machine without a GPU.  Any sanitizer report aborts the process; the driver's own checks
// return non-zero.  Run by tests/test_host_logic.py::test_product_host_code_under_address_and_ub_sanitizer.
//
//   host_asan_driver <valid.onnx> [<mangled.onnx> ...]
//
// Covered: the mask builders (setup_subsampling_spiralgrided.m / setup_subsampling_epi.m restated in api_core.cpp) incl. the capacity error,
// qmri_net_nparams, the ONNX reader on a well-formed file and on truncated / bit-flipped ones (an untrusted input), the weight packers of all
// four layer kinds in both operand-splitting schemes (conv_kernels.hip / conv6_kernels.hip host code), and the argument / state checks of the
// entry points that refuse to run without a context or a device.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qmri_internal.h"

static int fails = 0;
#define EXPECT(cond)                                                             \
    do {                                                                         \
        if (!(cond)) { std::fprintf(stderr, "driver check failed, line %d: %s\n", __LINE__, #cond); ++fails; } \
    } while (0)

static void masks() {
    for (int N : {8, 32, 64, 224}) {
        const int T = (N == 224) ? 48 : 9, S = (N == 224) ? 771 : 57;
        std::vector<int32_t> fp(T + 1);
        int m = 0;
        EXPECT(qmri_build_spiral(nullptr, N, S, T, fp.data(), nullptr, 0, &m) == QMRI_ERR_INVALID_ARG && m > 0);     // sizing call
        std::vector<int32_t> k(m);
        int m2 = 0;
        EXPECT(qmri_build_spiral(nullptr, N, S, T, fp.data(), k.data(), m, &m2) == QMRI_OK && m2 == m && fp[T] == m);
        for (int i = 0; i < m; ++i) EXPECT(k[i] >= 0 && k[i] < N * N);
        if (m > 1) { std::vector<int32_t> small(m - 1); EXPECT(qmri_build_spiral(nullptr, N, S, T, fp.data(), small.data(), m - 1, &m2) == QMRI_ERR_INVALID_ARG); }
        EXPECT(qmri_build_epi(nullptr, N, N, 1.0 / 5.0, T, fp.data(), nullptr, 0, &m) == QMRI_ERR_INVALID_ARG && m > 0);
        std::vector<int32_t> ke(m);
        EXPECT(qmri_build_epi(nullptr, N, N, 1.0 / 5.0, T, fp.data(), ke.data(), m, &m2) == QMRI_OK && m2 == m);
        for (int i = 0; i < m; ++i) EXPECT(ke[i] >= 0 && ke[i] < N * N);
    }
    int m = 0;
    int32_t fp1[2];
    EXPECT(qmri_build_spiral(nullptr, 0, 10, 1, fp1, nullptr, 0, &m) == QMRI_ERR_INVALID_ARG);
    EXPECT(qmri_build_epi(nullptr, 8, 8, 0.0, 1, fp1, nullptr, 0, &m) != QMRI_OK);
}

static void packers() {
    // the four layer kinds at ragged channel counts, both schemes; weights spanning many decades so that the f16 scaling path runs
    struct Case { ConvKind kind; int cin, cout; };
    const Case cases[] = {{CONV_3X3, 10, 64}, {CONV_3X3, 64, 10}, {CONV_3X3, 11, 24}, {CONV_3X3, 128, 128}, {CONV_3X3, 72, 40},
                          {CONV_DOWN, 64, 128}, {CONV_DOWN, 24, 40}, {CONV_UP, 128, 64}, {CONV_UP, 40, 24}};
    unsigned long long state = 88172645463325252ull;
    auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (float)((state >> 11) * (1.0 / 9007199254740992.0)) - 0.5f; };
    for (const Case& c : cases) {
        const int taps = (c.kind == CONV_3X3) ? 9 : 4;
        std::vector<float> w((size_t)c.cin * c.cout * taps);
        for (size_t i = 0; i < w.size(); ++i) w[i] = rnd() * std::pow(10.f, (float)((int)(i % 9) - 6));
        ConvLayer L;
        conv_plan_layer(L, c.kind, c.cin, c.cout);
        std::vector<float> p32;
        const size_t n = conv_pack_weights(L, w.data(), p32);
        EXPECT(n == p32.size() && n > 0);
        double s32 = 0.0, sw = 0.0;
        for (float v : p32) s32 += v;
        for (float v : w) sw += v;
        EXPECT(std::fabs(s32 - sw) <= 1e-6 * (1.0 + std::fabs(sw)) + 1e-3);      // a permutation with zero padding
        EXPECT(conv6_weights_fit_f16(w.data(), w.size()));
        for (int sp : {2, 3}) {
            L.sp6 = sp;
            std::vector<uint16_t> p6;
            if (L.kind == CONV_3X3 || L.kind == CONV_3X3N) conv6_plan_pack(L, w.data(), p6); else conv6s_plan_pack(L, w.data(), p6);
            EXPECT(!p6.empty() && L.nchunk6 > 0 && L.n_ct6 > 0);
        }
    }
    const float big[3] = {1.f, 7.0e4f, -2.f}, nan_[2] = {0.f, NAN};
    EXPECT(!conv6_weights_fit_f16(big, 3) && !conv6_weights_fit_f16(nan_, 2));
}

static void onnx(int argc, char** argv) {
    qmri_net_desc d{};
    size_t nf = 0;
    EXPECT(qmri_onnx_read_unetres("/nonexistent/file.onnx", &d, nullptr, 0, &nf) != QMRI_OK);
    EXPECT(qmri_onnx_read_unetres(nullptr, &d, nullptr, 0, &nf) != QMRI_OK);
    if (argc < 2) return;
    EXPECT(qmri_onnx_read_unetres(argv[1], &d, nullptr, 0, &nf) == QMRI_OK && nf > 0);
    EXPECT(nf == qmri_net_nparams(&d));
    std::vector<float> w(nf);
    size_t nf2 = 0;
    EXPECT(qmri_onnx_read_unetres(argv[1], &d, w.data(), nf, &nf2) == QMRI_OK && nf2 == nf);
    if (nf > 1) { std::vector<float> small(nf - 1); EXPECT(qmri_onnx_read_unetres(argv[1], &d, small.data(), nf - 1, &nf2) != QMRI_OK); }
    int ok = 0, bad = 0;
    for (int i = 2; i < argc; ++i) {                                     // mangled files: any status is fine, a sanitizer report is not
        std::vector<float> buf(nf + 16);
        qmri_net_desc dd{};
        size_t n3 = 0;
        const int st = qmri_onnx_read_unetres(argv[i], &dd, buf.data(), buf.size(), &n3);
        if (st == QMRI_OK) { ++ok; EXPECT(n3 <= buf.size()); } else { ++bad; EXPECT(std::strlen(qmri_last_error(nullptr)) > 0); }
    }
    std::printf("onnx: %d mangled files read, %d refused\n", ok, bad);
}

static void refusals() {
    qmri_net_desc d{};
    d.arch = QMRI_ARCH_UNETRES; d.in_nc = 10; d.out_nc = 10; d.nc[0] = 64; d.nc[1] = 128; d.nc[2] = 256; d.nc[3] = 512; d.nb = 4;
    EXPECT(qmri_net_nparams(&d) == 32648448u);
    d.in_nc = 11;
    EXPECT(qmri_net_nparams(&d) == 32649024u);
    EXPECT(qmri_net_nparams(nullptr) == 0);
    EXPECT(qmri_abi_version() == QMRI_ABI_VERSION);
    qmri_ctx* ctx = nullptr;
    const int st = qmri_create(0, &ctx);                                   // no device here: must fail loudly, with a message
    if (st != QMRI_OK) { EXPECT(ctx == nullptr && std::strlen(qmri_last_error(nullptr)) > 0); }
    else qmri_destroy(ctx);                                                // (a GPU box: fine too)
    EXPECT(qmri_create(0, nullptr) != QMRI_OK);
    double x[4] = {0, 0, 0, 0};
    float f[4];
    int32_t i4[4];
    int m = 0;
    EXPECT(qmri_operator_m(nullptr, &m) != QMRI_OK);
    EXPECT(qmri_forward(nullptr, x, 1, x) != QMRI_OK && qmri_adjoint(nullptr, x, x) != QMRI_OK);
    EXPECT(qmri_set_dictionary(nullptr, 1, 1, 1, f, f, f) != QMRI_OK);
    EXPECT(qmri_dict_match(nullptr, x, 1, f, f, f, i4) != QMRI_OK && qmri_dict_match_xfit(nullptr, x, 1, f, f, f, i4, f) != QMRI_OK);
    EXPECT(qmri_denoise(nullptr, x, 1, 1, 1, 1, x) != QMRI_OK);
    EXPECT(qmri_pnp_admm(nullptr, x, nullptr, nullptr, nullptr, x, nullptr, nullptr) != QMRI_OK);
    EXPECT(qmri_synchronize(nullptr) != QMRI_OK);
    char err[64];
    EXPECT(qmri_recon_batch(0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, err, sizeof err) == QMRI_ERR_INVALID_ARG);
}

int main(int argc, char** argv) {
    masks();
    packers();
    onnx(argc, argv);
    refusals();
    if (fails) { std::fprintf(stderr, "%d driver checks failed\n", fails); return 1; }
    std::printf("HOST_ASAN_DRIVER_OK\n");
    return 0;
}

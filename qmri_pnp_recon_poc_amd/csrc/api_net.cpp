// api_net.cpp -- denoiser plugin (param.net), PnP-ADMM driver, dictionary match and slice batching of libqmri.so.
//
// Replaces (reference file:line): param.net main_recon_tsmis_FFT.m:138-171 + denoiseImage_PnP_ADMM.m:1-117;
// PnP_ADMM.m:1-148; mrf_dtm_cpu.m:1-166.  UNetRes layer order follows state_dict() of network_unet.py:68-117.
#include "qmri_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <chrono>
#include <cstdlib>

void qmri_free_operator(qmri_ctx* ctx);
int qmri_prepare_direct(qmri_ctx* ctx, double r);
int qmri_lsqr_run(qmri_ctx* ctx, int B, const double2* d_z, double r, double tol, int maxit, double2* d_x,
                  int32_t* iters_out, int32_t* flag_out, double* pdiag, LsqrState* hslot, bool* deferred, const LsqrFuse* fuse);

// ---------------------------------------------------------------------------------------------------
// denoiser
// ---------------------------------------------------------------------------------------------------
extern "C" size_t qmri_net_nparams(const qmri_net_desc* d) {
    if (!d) return 0;
    size_t n = 0;
    if (d->arch == QMRI_ARCH_UNETRES) {
        const int32_t* nc = d->nc;
        n += (size_t)nc[0] * d->in_nc * 9;
        for (int l = 0; l < 3; ++l) n += (size_t)2 * d->nb * nc[l] * nc[l] * 9 + (size_t)nc[l + 1] * nc[l] * 4;
        n += (size_t)2 * d->nb * nc[3] * nc[3] * 9;
        for (int l = 3; l > 0; --l) n += (size_t)nc[l] * nc[l - 1] * 4 + (size_t)2 * d->nb * nc[l - 1] * nc[l - 1] * 9;
        n += (size_t)d->out_nc * nc[0] * 9;
    } else if (d->arch == QMRI_ARCH_SEQ_CONV) {
        if (d->nb == 1) return (size_t)d->out_nc * d->in_nc * 9;
        n = (size_t)d->nc[0] * d->in_nc * 9 + (size_t)(d->nb - 2) * d->nc[0] * d->nc[0] * 9 + (size_t)d->out_nc * d->nc[0] * 9;
    }
    return n;
}

void qmri_free_net(qmri_ctx* ctx) {
    NetPlan& p = ctx->net;
    for (ConvLayer& L : p.layers) { if (L.wp) (void)hipFree(L.wp); if (L.d_tab) (void)hipFree(L.d_tab); if (L.wp6) (void)hipFree(L.wp6); }
    for (float* b : p.allocs) if (b) (void)hipFree(b);
    if (p.d_wflat) (void)hipFree(p.d_wflat);
    if (p.d_counter) (void)hipFree(p.d_counter);
    if (p.d_stamps) (void)hipFree(p.d_stamps);
    if (p.d_c6part) (void)hipFree(p.d_c6part);
    if (p.d_res_xbuf) (void)hipFree(p.d_res_xbuf);
    if (p.d_res_stamps) (void)hipFree(p.d_res_stamps);
    if (p.d_io) (void)hipFree(p.d_io);
    if (p.d_range_flag) (void)hipFree(p.d_range_flag);
    if (p.h_range_flag) (void)hipHostFree(p.h_range_flag);
    if (p.d_act_slots) (void)hipFree(p.d_act_slots);
    if (p.d_act_count) (void)hipFree(p.d_act_count);
    if (p.d_act_ref) (void)hipFree(p.d_act_ref);
    p = NetPlan();
}

// zero-initialised padded activation tensor; Cal channels are allocated (>= C; the extra ones stay zero forever because
// kernels only ever write channels < C and plane interiors), plus slack for tiles that overhang the image
static int alloc_tensor(qmri_ctx* ctx, PTensor& t, int C, int Cal, int H, int W, size_t B) {
    t.C = C; t.Cal = std::max(C, Cal); t.H = H; t.W = W;
    t.h0 = 32; t.hp = ((t.h0 + H + 1 + 31) / 32) * 32;
    const size_t count = B * t.batch_stride() + 8192;
    hipError_t e = hipMalloc((void**)&t.p, count * sizeof(float));
    if (e != hipSuccess) { qmri_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(float), hipGetErrorString(e)); return QMRI_ERR_NOMEM; }
    ctx->net.allocs.push_back(t.p);
    QMRI_HIP(ctx, hipMemset(t.p, 0, count * sizeof(float)));
    return QMRI_OK;
}

static int pack_layer6(qmri_ctx* ctx, ConvLayer& L, const float* w) {
    std::vector<uint16_t> p6;
    if (L.kind == CONV_3X3 || L.kind == CONV_3X3N) conv6_plan_pack(L, w, p6);     // (conv_plan_layer renames narrow 3x3 layers)
    else conv6s_plan_pack(L, w, p6);
    hipError_t e = hipMalloc(&L.wp6, p6.size() * sizeof(uint16_t));
    if (e != hipSuccess) { qmri_set_error(ctx, "hipMalloc (weights) failed: %s", hipGetErrorString(e)); return QMRI_ERR_NOMEM; }
    QMRI_HIP(ctx, hipMemcpy(L.wp6, p6.data(), p6.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return QMRI_OK;
}

static size_t layer_weight_count(const ConvLayer& L) { return (size_t)L.Cin * L.Cout * ((L.kind == CONV_3X3 || L.kind == CONV_3X3N) ? 9 : 4); }

// knob pack_gpu = 1 (default, round 6): the layer is only PLANNED here; net_pack_all_dev splits and orders every layer's weights on the device.
// pack_gpu = 0: the host packers (round 1; kept as the reference of the packing and for the sanitised host build), layer by layer.
static int add_layer(qmri_ctx* ctx, ConvKind kind, int Cin, int Cout, const float*& w) {
    ConvLayer L;
    conv_plan_layer(L, kind, Cin, Cout);
    L.w_off = (size_t)(w - ctx->net.w_begin);
    L.sp6 = ctx->net.sp6;
    if (!ctx->net.d_wflat) {
        std::vector<float> packed;
        L.wp_floats = conv_pack_weights(L, w, packed);
        hipError_t e = hipMalloc((void**)&L.wp, packed.size() * sizeof(float));
        if (e != hipSuccess) { qmri_set_error(ctx, "hipMalloc (weights) failed: %s", hipGetErrorString(e)); return QMRI_ERR_NOMEM; }
        QMRI_HIP(ctx, hipMemcpy(L.wp, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        QMRI_TRY(pack_layer6(ctx, L, w));                      // the same weights, split for the bf16 / f16 matrix-core kernels
    }
    w += layer_weight_count(L);
    L.index = (int)ctx->net.layers.size();
    ctx->net.layers.push_back(L);
    return QMRI_OK;
}

static int pack_layer6_dev(qmri_ctx* ctx, ConvLayer& L) {
    const NetPlan& p = ctx->net;
    const float mx = (L.index >= 0 && (size_t)L.index < p.w_max.size()) ? p.w_max[L.index] : 0.f;
    if (L.kind == CONV_3X3 || L.kind == CONV_3X3N) return conv6_pack_dev(ctx, L, p.d_wflat + L.w_off, mx);
    return conv6s_pack_dev(ctx, L, p.d_wflat + L.w_off, mx);
}

// The device side of qmri_set_denoiser's weight handling: every layer's largest |w| (the f16 scheme's per-layer scale and its range check), then
// the three packings per layer as kernels reading the flat blob in device memory.  One host synchronisation (the maxima).
static int net_pack_all_dev(qmri_ctx* ctx) {
    NetPlan& p = ctx->net;
    const size_t nl = p.layers.size();
    unsigned* d_max = nullptr;
    QMRI_HIP(ctx, hipMalloc((void**)&d_max, std::max<size_t>(nl, 1) * sizeof(unsigned)));
    int rc = QMRI_OK;
    do {
        if (hipMemsetAsync(d_max, 0, nl * sizeof(unsigned), ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        for (size_t l = 0; l < nl && rc == QMRI_OK; ++l)
            rc = ew_launch_absmax(ctx, p.d_wflat + p.layers[l].w_off, nullptr, layer_weight_count(p.layers[l]), d_max + l);
        if (rc != QMRI_OK) break;
        std::vector<unsigned> bits(nl);
        if (hipMemcpyAsync(bits.data(), d_max, nl * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        p.w_max.resize(nl);
        bool fit = true;
        for (size_t l = 0; l < nl; ++l) {
            std::memcpy(&p.w_max[l], &bits[l], 4);
            if (!(p.w_max[l] <= 60000.f)) fit = false;             // (conv6_weights_fit_f16: the f16 pieces carry |w| <= 6e4; NaN too)
        }
        if (p.sp6 == 2 && !fit) p.sp6 = 3;                         // weights beyond the f16 range: bf16 scheme
        for (ConvLayer& L : p.layers) {
            L.sp6 = p.sp6;
            if ((rc = conv_pack_weights_dev(ctx, L, p.d_wflat + L.w_off)) != QMRI_OK) break;
            if ((rc = pack_layer6_dev(ctx, L)) != QMRI_OK) break;
        }
    } while (0);
    (void)hipFree(d_max);
    if (rc == QMRI_ERR_HIP && ctx->err.empty()) qmri_set_error(ctx, "HIP failure while packing the denoiser's weights on the device");
    return rc;
}

// Re-pack every layer for the other operand-splitting scheme (sp = 2: f16 x 3 products, sp = 3: bf16 x 6 products).
static int net_set_scheme(qmri_ctx* ctx, int sp) {
    NetPlan& p = ctx->net;
    if (p.sp6 == sp) return QMRI_OK;
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (ConvLayer& L : p.layers) {
        if (L.wp6) { (void)hipFree(L.wp6); L.wp6 = nullptr; }
        L.sp6 = sp;
        if (p.d_wflat) QMRI_TRY(pack_layer6_dev(ctx, L));
        else QMRI_TRY(pack_layer6(ctx, L, p.w_host.data() + L.w_off));
    }
    if (!p.d_wflat) QMRI_HIP(ctx, hipDeviceSynchronize());         // (the host packers' copies travel on the NULL stream: see qmri_set_denoiser)
    p.sp6 = sp;
    return QMRI_OK;
}

// After a synchronisation: did a layer's output leave the range the f16 split carries (|x| <= 6e4, finite)?  If so the
// network is switched to the bf16 scheme (8 exponent bits, no range limit) and the caller runs its work again.
static int net_range_tripped(qmri_ctx* ctx, bool& tripped) {
    NetPlan& p = ctx->net;
    tripped = false;
    if (p.sp6 != 2 || !p.d_range_flag) return QMRI_OK;
    unsigned f = 0;
    // (on the context's own stream: the callers have synchronised it, and the kernels of the repeated run that raise this flag again are ordered
    //  behind the reset -- a NULL-stream memset is not ordered with a non-blocking stream, see qmri_set_denoiser)
    QMRI_HIP(ctx, hipMemcpyAsync(&f, p.d_range_flag, sizeof f, hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
#ifdef QMRI_TIMING_ONLY
    f = 0;
#endif
    if (!f) return QMRI_OK;
    QMRI_HIP(ctx, hipMemsetAsync(p.d_range_flag, 0, sizeof f, ctx->stream));
    if (p.h_range_flag) std::memset(p.h_range_flag, 0, (size_t)p.h_range_words * sizeof(unsigned));
    tripped = true;
    if (f & 4u) {                                                   // a hand-off of the resident-tile launch timed out: its results are garbage, and so
        // may the other bits be.  Co-residency of its workgroups is assumed from tiles <= CUs; another stream, context or process on the device can break
        // it for a while (tests/test_gpu_net.py: two contexts).  The call is repeated with one launch per layer; the resident form is tried again after
        // res_rearm clean passes (knob, default 64) -- unless it has timed out three times since qmri_set_denoiser: then it stays off.
        p.res_off = true;
        p.res_timeouts += 1;
        p.res_clean = 0;
        if (p.res_timeouts <= 3)
            fprintf(stderr, "libqmri: a tile hand-off of the resident-tile convolution launch timed out (%d since set-up); repeating with one launch per layer%s\n",
                    p.res_timeouts, p.res_timeouts >= 3 ? ", the resident form stays off" : "");
        return QMRI_OK;
    }
    if (qmri_knob(K_VERBOSE)) fprintf(stderr, "libqmri: range guard of the f16 scheme tripped (flag %u: 1 overflow, 2 a layer collapsed): the network moves to the bf16 scheme\n", f);
    QMRI_TRY(net_set_scheme(ctx, 3));
    p.fallbacks += 1;
    return QMRI_OK;
}

// any bit in the pinned host words of the range guards (k_act_check, conv6_kernels.hip)
static bool host_range_tripped(const NetPlan& p) {
#ifdef QMRI_TIMING_ONLY
    return false;
#endif
    if (!p.h_range_flag) return false;
    const int n = std::min(p.h_range_words, (int)p.layers.size() + 1);
    for (int i = 0; i < n; ++i) if (p.h_range_flag[i]) return true;
    return false;
}

template <typename T> static int dev_alloc(qmri_ctx* ctx, T** p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { qmri_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e)); return QMRI_ERR_NOMEM; }
    return QMRI_OK;
}

static int net_forward_padded(qmri_ctx* ctx, int B);

// The f16 operand split represents values below ~2.4e-4 with an absolute, not a relative error.  Inputs
// are at unit scale by construction ([0, 1] in the ADMM loop, rescaled in qmri_denoise) and weights are rescaled per layer; what
// is left is the network's own gain from layer to layer.  Whether that matters for THIS network is measured once at set-up
// time: one forward on a unit-scale probe with the f16 kernels, one with the f32-MFMA kernels (their weights are packed anyway),
// and if the outputs differ by more than fp32 summation-order noise -- or the overflow guard trips -- the network runs on the
// bf16 scheme (no range limits) from the start.  (The run-time overflow guard stays: it covers inputs the probe did not see.)
static int net_calibrate_scheme(qmri_ctx* ctx) {
    NetPlan& p = ctx->net;
    struct ResOff { NetPlan& n; bool was; ~ResOff() { n.res_off = was; } } res_guard{p, p.res_off};
    p.res_off = true;                                                // (one launch per layer here: same bits, and a hand-off time-out could not be told from a range problem)
    const size_t n = (size_t)p.desc.in_nc * p.H * p.W, nout = p.out32.batch_stride();
    std::vector<float> h(n);
    uint32_t st = 0x2545F491u;
    for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) * (1.0f / 16777216.0f); }   // uniform [0, 1)
    float *d_tmp = nullptr, *d_ref = nullptr;
    unsigned* d_m = nullptr;
    int rc = QMRI_OK;
    do {
        if (hipMalloc((void**)&d_tmp, n * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_ref, nout * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&d_m, 2 * sizeof(unsigned)) != hipSuccess) { rc = QMRI_ERR_NOMEM; break; }
        if (hipMemsetAsync(d_m, 0, 2 * sizeof(unsigned), ctx->stream) != hipSuccess ||
            hipMemcpyAsync(d_tmp, h.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        if ((rc = ew_launch_pack(ctx, 1, p.desc.in_nc, p.H, p.W, d_tmp, 0, p.in32)) != QMRI_OK) break;
        p.force_f32 = true;                                          // reference: exact fp32 products
        rc = net_forward_padded(ctx, 1);
        p.force_f32 = false;
        if (rc != QMRI_OK) break;
        if (hipMemcpyAsync(d_ref, p.out32.p, nout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        p.act_record = true;                                         // ... which also record every layer's magnitude (ACT_LOW guard)
        rc = net_forward_padded(ctx, 1);                             // the f16 kernels
        p.act_record = false;
        if (rc != QMRI_OK) break;
        if ((rc = ew_launch_absmax(ctx, d_ref, nullptr, nout, d_m)) != QMRI_OK) break;
        if ((rc = ew_launch_absmax(ctx, p.out32.p, d_ref, nout, d_m + 1)) != QMRI_OK) break;
        unsigned m[2] = {0, 0}, flag = 0;
        if (hipMemcpyAsync(m, d_m, sizeof m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&flag, p.d_range_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
        float ref_max, diff_max;
        std::memcpy(&ref_max, &m[0], 4); std::memcpy(&diff_max, &m[1], 4);
#ifdef QMRI_TIMING_ONLY   // builds with parts of a kernel removed (tools/ab_*.sh): wrong results by design -- without this the probe would put them on the bf16 scheme
        const bool ok = true;
#else
        const bool ok = flag == 0 && std::isfinite(ref_max) && std::isfinite(diff_max) && diff_max <= 2e-5f * ref_max;
#endif
        if (qmri_knob(K_VERBOSE))
            fprintf(stderr, "libqmri: calibration probe: max |out| %.3g, max |f16 - f32| %.3g, overflow flag %u -> %s scheme\n", ref_max, diff_max, flag,
                    ok ? "f16 x 3" : "bf16 x 6");
        if (!ok) {
            if (hipMemsetAsync(p.d_range_flag, 0, sizeof flag, ctx->stream) != hipSuccess) { rc = QMRI_ERR_HIP; break; }
            rc = net_set_scheme(ctx, 3);
        }
    } while (0);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_ref) (void)hipFree(d_ref);
    if (d_m) (void)hipFree(d_m);
    if (rc == QMRI_ERR_HIP && ctx->err.empty()) qmri_set_error(ctx, "HIP failure in the denoiser calibration pass");
    if (rc == QMRI_ERR_NOMEM && ctx->err.empty()) qmri_set_error(ctx, "hipMalloc failed in the denoiser calibration pass");
    return rc;
}

extern "C" int qmri_set_denoiser(qmri_ctx* ctx, const qmri_net_desc* desc, const float* weights, size_t nbytes, int H, int W,
                                 int max_batch) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_CHECK_ARG(ctx, desc && weights, "desc / weights must not be NULL");
    QMRI_CHECK_ARG(ctx, H > 0 && W > 0 && max_batch > 0, "H, W, max_batch must be positive");
    QMRI_CHECK_ARG(ctx, desc->in_nc > 0 && desc->out_nc > 0 && desc->nb >= 1, "in_nc, out_nc, nb must be positive");
    if (desc->arch != QMRI_ARCH_UNETRES && desc->arch != QMRI_ARCH_SEQ_CONV) {
        qmri_set_error(ctx, "unknown network architecture %d", desc->arch);
        return QMRI_ERR_UNSUPPORTED;
    }
    if (desc->arch == QMRI_ARCH_UNETRES && (H % 8 != 0 || W % 8 != 0)) {
        // UNetRes has three 2x down-samplers and no padding logic (network_unet.py:106-117)
        qmri_set_error(ctx, "UNetRes needs H and W divisible by 8 (got %d x %d)", H, W);
        return QMRI_ERR_UNSUPPORTED;
    }
    if (nbytes != 4 * qmri_net_nparams(desc)) {
        qmri_set_error(ctx, "weight blob is %zu bytes, architecture needs %zu", nbytes, 4 * qmri_net_nparams(desc));
        return QMRI_ERR_INVALID_ARG;
    }
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    qmri_free_net(ctx);
    NetPlan& p = ctx->net;
    p.desc = *desc; p.H = H; p.W = W; p.maxB = max_batch;
    p.w_begin = weights;
    p.sp6 = conv6_default_sp();
    const auto t_upload = std::chrono::steady_clock::now();
    if (qmri_knob(K_PACK_GPU)) {
        // the caller's blob goes to the device once, as it is; every packing is a kernel over it (net_pack_all_dev) and a later change of scheme re-packs from it
        QMRI_HIP(ctx, hipMalloc((void**)&p.d_wflat, nbytes));
        QMRI_HIP(ctx, hipMemcpyAsync(p.d_wflat, weights, nbytes, hipMemcpyHostToDevice, ctx->stream));   // (on the stream the packing kernels run on: ordered with them)
    } else {
        p.w_host.assign(weights, weights + nbytes / 4);
        if (p.sp6 == 2 && !conv6_weights_fit_f16(weights, nbytes / 4)) p.sp6 = 3;     // weights beyond the f16 range: bf16 scheme
    }
    QMRI_HIP(ctx, hipMalloc((void**)&p.d_range_flag, sizeof(unsigned)));
    QMRI_HIP(ctx, hipMemset(p.d_range_flag, 0, sizeof(unsigned)));
    p.h_range_words = 4096;
    QMRI_HIP(ctx, hipHostMalloc((void**)&p.h_range_flag, (size_t)p.h_range_words * sizeof(unsigned), hipHostMallocDefault));
    std::memset(p.h_range_flag, 0, (size_t)p.h_range_words * sizeof(unsigned));
    const float* w = weights;
    const int nb = desc->nb;
    const size_t B = (size_t)max_batch;
    // (qmri_get_health: where the set-up time goes -- the layers' weight packing and upload, the tensors, the calibration probe)
    typedef std::chrono::steady_clock Clk;
    const auto t_begin = Clk::now();
    double ms_pack = std::chrono::duration<double, std::milli>(t_begin - t_upload).count();      // (the blob's upload / host copy and range check)
    auto add_layer_timed = [&](qmri_ctx* c, ConvKind kind, int cin, int cout, const float*& wp) {
        const auto t0 = Clk::now();
        const int rc = add_layer(c, kind, cin, cout, wp);
        ms_pack += std::chrono::duration<double, std::milli>(Clk::now() - t0).count();
        return rc;
    };
    if (desc->arch == QMRI_ARCH_UNETRES) {
        const int32_t* nc = desc->nc;
        QMRI_TRY(add_layer_timed(ctx, CONV_3X3, desc->in_nc, nc[0], w));
        for (int l = 0; l < 3; ++l) {
            for (int b = 0; b < 2 * nb; ++b) QMRI_TRY(add_layer_timed(ctx, CONV_3X3, nc[l], nc[l], w));
            QMRI_TRY(add_layer_timed(ctx, CONV_DOWN, nc[l], nc[l + 1], w));
        }
        for (int b = 0; b < 2 * nb; ++b) QMRI_TRY(add_layer_timed(ctx, CONV_3X3, nc[3], nc[3], w));
        for (int l = 3; l > 0; --l) {
            QMRI_TRY(add_layer_timed(ctx, CONV_UP, nc[l], nc[l - 1], w));
            for (int b = 0; b < 2 * nb; ++b) QMRI_TRY(add_layer_timed(ctx, CONV_3X3, nc[l - 1], nc[l - 1], w));
        }
        QMRI_TRY(add_layer_timed(ctx, CONV_3X3, nc[0], desc->out_nc, w));
        for (int l = 0; l < 4; ++l) {
            // channels allocated = the largest padded Cin of any layer that reads a level-l tensor
            int cal = conv_cin_pad(CONV_3X3, nc[l]);
            if (l < 3) cal = std::max(cal, conv_cin_pad(CONV_DOWN, nc[l]));
            if (l > 0) cal = std::max(cal, conv_cin_pad(CONV_UP, nc[l]));
            QMRI_TRY(alloc_tensor(ctx, p.x[l], nc[l], cal, H >> l, W >> l, B));
            QMRI_TRY(alloc_tensor(ctx, p.a[l], nc[l], cal, H >> l, W >> l, B));
            QMRI_TRY(alloc_tensor(ctx, p.t[l], nc[l], cal, H >> l, W >> l, B));
        }
        if (nc[0] == 64 && H % 16 == 0 && W % 16 == 0 && (H / 16) * (W / 16) <= 1024) {                   // k_conv6r's exchange buffer (one slice)
            p.res_tiles = (H / 16) * (W / 16);
            QMRI_TRY(dev_alloc(ctx, &p.d_res_xbuf, conv6r_xbuf_bytes(p.res_tiles)));
            QMRI_HIP(ctx, hipMemset(p.d_res_xbuf, 0, conv6r_xbuf_bytes(p.res_tiles)));
            p.res_epoch = 0; p.res_off = false;
            if (qmri_knob(K_RES_STAMPS)) { QMRI_HIP(ctx, hipMalloc(&p.d_res_stamps, 1024 * sizeof(unsigned long long))); QMRI_HIP(ctx, hipMemset(p.d_res_stamps, 0, 1024 * sizeof(unsigned long long))); }
        }
    } else {
        const int width = desc->nc[0];
        if (nb == 1) QMRI_TRY(add_layer_timed(ctx, CONV_3X3, desc->in_nc, desc->out_nc, w));
        else {
            QMRI_TRY(add_layer_timed(ctx, CONV_3X3, desc->in_nc, width, w));
            for (int l = 1; l < nb - 1; ++l) QMRI_TRY(add_layer_timed(ctx, CONV_3X3, width, width, w));
            QMRI_TRY(add_layer_timed(ctx, CONV_3X3, width, desc->out_nc, w));
        }
        QMRI_TRY(alloc_tensor(ctx, p.a[0], width, conv_cin_pad(CONV_3X3, width), H, W, B));
        QMRI_TRY(alloc_tensor(ctx, p.t[0], width, conv_cin_pad(CONV_3X3, width), H, W, B));
    }
    if (p.d_wflat) {
        const auto t0 = Clk::now();
        QMRI_TRY(net_pack_all_dev(ctx));
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (the time of the packing kernels belongs to this figure)
        ms_pack += std::chrono::duration<double, std::milli>(Clk::now() - t0).count();
    }
    p.w_begin = nullptr;                                           // (the caller's pointer is not kept)
    QMRI_TRY(alloc_tensor(ctx, p.in32, desc->in_nc, conv_cin_pad(CONV_3X3, desc->in_nc), H, W, B));
    QMRI_TRY(alloc_tensor(ctx, p.out32, desc->out_nc, desc->out_nc, H, W, B));
    // BLOCKED interior tensors (PTensor::blk): every layer on the conv6 kernels, every interior channel count a multiple of 8
    p.blk_ok = conv6_enabled();
    for (const ConvLayer& L : p.layers) if (!L.wp6) p.blk_ok = false;
    if (desc->arch == QMRI_ARCH_UNETRES) { for (int l = 0; l < 4; ++l) if (desc->nc[l] % 8) p.blk_ok = false; }
    else if (nb > 1 && desc->nc[0] % 8) p.blk_ok = false;
    for (int l = 0; l < 4; ++l) if ((p.x[l].p && p.x[l].Cal % 8) || (p.a[l].p && p.a[l].Cal % 8) || (p.t[l].p && p.t[l].Cal % 8)) p.blk_ok = false;
    p.interior_fmt = -1;
    QMRI_TRY(dev_alloc(ctx, &p.d_counter, (size_t)1));
    QMRI_HIP(ctx, hipMemset(p.d_counter, 0, sizeof(unsigned)));
    if (qmri_knob(K_CONV_STAMPS)) { QMRI_HIP(ctx, hipMalloc(&p.d_stamps, 4096 * 11 * sizeof(unsigned long long))); }
    p.counter_base = 0;
    p.ready = true;
    // (device-wide: the host packers' blocking copies travel on the NULL stream, which this context's non-blocking stream is not ordered with -- beside
    //  another process on the device the set-up probe below was seen to read weights that had not landed and to choose the bf16 scheme for a network
    //  that does not need it, five times out of six: tools/probe_under_contention.py, profiles/r06_e_*)
    QMRI_HIP(ctx, hipDeviceSynchronize());
    const auto t_cal = Clk::now();
    if (p.sp6 == 2) QMRI_TRY(net_calibrate_scheme(ctx));
    const auto t_end = Clk::now();
    p.setup_ms[0] = ms_pack;
    p.setup_ms[2] = std::chrono::duration<double, std::milli>(t_end - t_cal).count();
    p.setup_ms[1] = std::chrono::duration<double, std::milli>(t_cal - t_upload).count() - ms_pack;
    return QMRI_OK;
}

// one conv launch with optional per-launch timing of the dominant kernel (profile level 2)
static int run_conv(qmri_ctx* ctx, ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                    const PTensor* add2, int relu) {
    return conv_launch(ctx, L, B, in, out, add1, add2, relu);     // (profile level 2: the launchers mark their kernels)
}

// nb ResBlocks: cur <- cur + conv(relu(conv(cur)))  (basicblock.py:211-223).  `src` is the block input of the first
// ResBlock (may be a skip tensor that must stay intact); results land in `cur`; `skip` is added by the last conv.
static int run_resblocks(qmri_ctx* ctx, size_t& li, int nb, int B, const PTensor& src, const PTensor& cur, const PTensor& tmp,
                         const PTensor* skip) {
    NetPlan& p = ctx->net;
    if (p.d_res_xbuf && src.H == p.H && !p.force_f32) {            // the full-resolution level: one launch with resident tiles where it applies
        bool done = false;
        Conv6rRun r;
        r.res = &p.layers[li]; r.nres = 2 * nb; r.src = &src; r.cur = &cur; r.skip = skip;
        QMRI_TRY(conv6r_try(ctx, r, B, &done));
        if (done) { li += (size_t)(2 * nb); return QMRI_OK; }
    }
    const PTensor* in = &src;
    for (int b = 0; b < nb; ++b) {
        QMRI_TRY(run_conv(ctx, p.layers[li++], B, *in, tmp, nullptr, nullptr, 1));
        QMRI_TRY(run_conv(ctx, p.layers[li++], B, tmp, cur, in, (b == nb - 1) ? skip : nullptr, 0));
        in = &cur;
    }
    return QMRI_OK;
}

static int net_forward_layers(qmri_ctx* ctx, int B);

// network forward on the context's padded tensors: in32 -> out32
static int net_forward_padded(qmri_ctx* ctx, int B) {
    {
        // Interior tensors (everything between the head's input and the tail's output) are BLOCKED when every layer runs on the
        // matrix-core kernels (PTensor::blk), planar otherwise (the f32-MFMA kernels of the calibration pass, knob conv_f32, odd
        // channel counts).  The two formats put the zero halo at different addresses: a change of format re-zeroes the tensors.
        NetPlan& p = ctx->net;
        const bool blk = p.blk_ok && !p.force_f32 && conv6_enabled();
        if (p.interior_fmt != -1 && p.interior_fmt != (blk ? 1 : 0)) {
            PTensor* ts[12];
            int nt = 0;
            for (int l = 0; l < 4; ++l) { ts[nt++] = &p.x[l]; ts[nt++] = &p.a[l]; ts[nt++] = &p.t[l]; }
            for (int i = 0; i < nt; ++i)
                if (ts[i]->p) QMRI_HIP(ctx, hipMemsetAsync(ts[i]->p, 0, ((size_t)p.maxB * ts[i]->batch_stride() + 8192) * sizeof(float), ctx->stream));
        }
        p.interior_fmt = blk ? 1 : 0;
        for (int l = 0; l < 4; ++l) { p.x[l].blk = blk; p.a[l].blk = blk; p.t[l].blk = blk; }
    }
    const bool report = !ctx->net.force_f32;                        // (the calibration's f32 pass reports nothing)
    const bool timed = ctx->prof_level >= 2;                        // profile level 2: the whole pass between two stream events (besides the per-launch pairs)
    if (timed) QMRI_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (report) QMRI_TRY(conv6_act_begin(ctx, (int)ctx->net.layers.size()));
    QMRI_TRY(net_forward_layers(ctx, B));
    if (report) QMRI_TRY(conv6_act_end(ctx));                       // low-magnitude guard of the f16 scheme (conv6_kernels.hip, ACT_LOW)
    if (timed) QMRI_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    QMRI_TRY(qmri_prof_chain_finish(ctx));                          // (level 2: synchronises)
    if (timed) {
        float ms = 0.f;
        QMRI_HIP(ctx, hipEventSynchronize(ctx->ev[3]));
        QMRI_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
        ctx->prof.ms_net_forward += ms; ctx->prof.n_net_forward += 1;
    }
    return QMRI_OK;
}

static int net_forward_layers(qmri_ctx* ctx, int B) {
    NetPlan& p = ctx->net;
    const int nb = p.desc.nb;
    size_t li = 0;
    if (p.desc.arch == QMRI_ARCH_SEQ_CONV) {
        const size_t nl = p.layers.size();
        const PTensor* in = &p.in32;
        const PTensor* bufs[2] = { &p.a[0], &p.t[0] };
        for (size_t l = 0; l < nl; ++l) {
            const PTensor* out = (l == nl - 1) ? &p.out32 : bufs[l & 1];
            QMRI_TRY(run_conv(ctx, p.layers[l], B, *in, *out, nullptr, nullptr, l != nl - 1));
            in = out;
        }
        return QMRI_OK;
    }
    // UNetRes.forward, network_unet.py:106-117
    // (the head belongs to the down path's resident-tile launch of the full-resolution level where that applies: knob res_head = 0 keeps it apart)
    const bool res_head = qmri_knob(K_RES_HEAD) != 0, res_tail = qmri_knob(K_RES_TAIL) != 0, res_down = qmri_knob(K_RES_DOWN) != 0;
    bool head_done = false, down_done = false;
    if (res_head && p.d_res_xbuf && !p.force_f32 && p.layers.size() >= (size_t)(2 + 2 * nb)) {
        Conv6rRun r;
        r.head = &p.layers[0]; r.head_in = &p.in32; r.res = &p.layers[1]; r.nres = 2 * nb; r.src = &p.x[0]; r.cur = &p.a[0];
        if (res_down) {                                             // ... and the level's down-sampling convolution behind them
            r.down = &p.layers[1 + 2 * nb]; r.down_out = &p.x[1];
            QMRI_TRY(conv6r_try(ctx, r, B, &down_done));
            if (down_done) { head_done = true; li = (size_t)(2 + 2 * nb); }
            r.down = nullptr; r.down_out = nullptr;
        }
        if (!down_done) {
            QMRI_TRY(conv6r_try(ctx, r, B, &head_done));
            if (head_done) li = (size_t)(1 + 2 * nb);
        }
    }
    if (!head_done) QMRI_TRY(run_conv(ctx, p.layers[li++], B, p.in32, p.x[0], nullptr, nullptr, 0));      // x1 = m_head(x0)
    for (int l = 0; l < 3; ++l) {                                                                          // x_{l+2} = m_down_{l+1}(x_{l+1})
        if (!(l == 0 && head_done)) QMRI_TRY(run_resblocks(ctx, li, nb, B, p.x[l], p.a[l], p.t[l], nullptr));
        if (!(l == 0 && down_done)) QMRI_TRY(run_conv(ctx, p.layers[li++], B, p.a[l], p.x[l + 1], nullptr, nullptr, 0));
    }
    QMRI_TRY(run_resblocks(ctx, li, nb, B, p.x[3], p.a[3], p.t[3], &p.x[3]));                               // m_body(x4) + x4
    for (int l = 3; l > 0; --l) {                                                                          // m_up_l(x + x_{l+1})
        QMRI_TRY(run_conv(ctx, p.layers[li++], B, p.a[l], p.a[l - 1], nullptr, nullptr, 0));               // transposed conv
        if (l == 1 && res_tail && p.d_res_xbuf && !p.force_f32 && li + (size_t)(2 * nb) < p.layers.size()) {   // ... the level's ResBlocks and the tail in one launch
            bool done = false;
            Conv6rRun r;
            r.res = &p.layers[li]; r.nres = 2 * nb; r.src = &p.a[0]; r.cur = &p.a[0]; r.skip = &p.x[0]; r.tail = &p.layers[li + 2 * nb]; r.tail_out = &p.out32;
            QMRI_TRY(conv6r_try(ctx, r, B, &done));
            if (done) { li += (size_t)(2 * nb + 1); return QMRI_OK; }
        }
        QMRI_TRY(run_resblocks(ctx, li, nb, B, p.a[l - 1], p.a[l - 1], p.t[l - 1], &p.x[l - 1]));
    }
    QMRI_TRY(run_conv(ctx, p.layers[li++], B, p.a[0], p.out32, nullptr, nullptr, 0));                      // m_tail(x + x1)
    return QMRI_OK;
}

// (A hipGraph replay of the forward pass -- ~65 dependent launches with fixed arguments -- was measured in round 1 and is not faster: 413.7 vs 413.8
//  ADMM it/s; the 3-4 us between dependent kernels are spent on the device, not on the host.  The capture path is gone since round 5.)
static int net_forward(qmri_ctx* ctx, int B) {
    NetPlan& p = ctx->net;
    const int st = net_forward_padded(ctx, B);
    // the resident-tile launch is switched off by a hand-off time-out (net_range_tripped); after K_RES_REARM clean forward passes it is tried again
    if (st == QMRI_OK && p.res_off && p.res_timeouts > 0 && p.res_timeouts < 3 && !p.res_forced_off && ++p.res_clean >= std::max(1, qmri_knob(K_RES_REARM))) {
        p.res_off = false;
        p.res_clean = 0;
    }
    return st;
}

extern "C" int qmri_net_forward_dev(qmri_ctx* ctx, const float* d_in, int B, float* d_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    NetPlan& p = ctx->net;
    if (!p.ready) { qmri_set_error(ctx, "denoiser not set: call qmri_set_denoiser first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, d_in && d_out && B >= 1 && B <= p.maxB, "qmri_net_forward_dev arguments / batch > max_batch");
    {   // the guarded second attempt reads d_in again: input and output must not overlap
        const size_t hw = (size_t)p.H * p.W * B * sizeof(float);
        const char *a = (const char*)d_in, *b = (const char*)d_out;
        QMRI_CHECK_ARG(ctx, a + hw * p.desc.in_nc <= b || b + hw * p.desc.out_nc <= a, "qmri_net_forward_dev: d_in and d_out must not overlap");
    }
    // A pass is repeated for two separate reasons -- a hand-off time-out of the resident-tile launch (then one launch per layer), the f16 range guard
    // (then the bf16 scheme) -- and one may follow the other: up to three passes, and a call that still wants another one is an error, not QMRI_OK.
    bool again = false;
    for (int attempt = 0; attempt < 3; ++attempt) {
        QMRI_TRY(ew_launch_pack(ctx, B, p.desc.in_nc, p.H, p.W, d_in, 0, p.in32));
        QMRI_TRY(net_forward(ctx, B));
        QMRI_TRY(ew_launch_unpack(ctx, B, p.desc.out_nc, p.H, p.W, p.out32, p.in32, 0, d_out, 0));
        again = false;
        if (p.sp6 != 2) break;                              // (the bf16 scheme has no range to guard: stays asynchronous)
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        QMRI_TRY(net_range_tripped(ctx, again));
        if (!again) break;
    }
    if (again) { qmri_set_error(ctx, "the network's guards asked for a fourth pass (resident-tile hand-off / f16 range): giving up"); return QMRI_ERR_HIP; }
    return QMRI_OK;
}

// test / A-B hook: the resident-tile launch of the full-resolution ResBlocks (conv6_kernels.hip k_conv6r)
extern "C" int qmri_debug_conv_resident(qmri_ctx* ctx, int on, int* timeouts_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    NetPlan& p = ctx->net;
    if (timeouts_out) *timeouts_out = p.res_timeouts;
    if (!p.ready) return QMRI_OK;
    p.res_off = (on == 0);
    p.res_forced_off = (on == 0);                                   // (a caller's choice is not re-armed behind its back)
    if (on) { p.res_timeouts = std::min(p.res_timeouts, 2); p.res_clean = 0; }
    p.res_drop = (on == 2) ? 1 : 0;
    return QMRI_OK;
}

extern "C" int qmri_denoiser_scheme(const qmri_ctx* ctx, int* scheme_out, int* fallbacks_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    if (!ctx->net.ready) return QMRI_ERR_STATE;
    if (scheme_out) *scheme_out = ctx->net.sp6;
    if (fallbacks_out) *fallbacks_out = ctx->net.fallbacks;
    return QMRI_OK;
}

extern "C" int qmri_get_health(const qmri_ctx* ctx, qmri_health* out) {
    if (!ctx || !out) return QMRI_ERR_INVALID_ARG;
    const NetPlan& p = ctx->net;
    std::memset(out, 0, sizeof *out);
    out->denoiser_scheme = p.ready ? p.sp6 : 0;
    out->denoiser_fallbacks = p.ready ? p.fallbacks : 0;
    out->resident_armed = (p.ready && !p.res_off && p.sp6 == 2 && qmri_knob(K_CONV_RESIDENT) != 0) ? 1 : 0;
    out->resident_timeouts = p.res_timeouts;
    out->lsqr_one_launch = (ctx->ks_persist < 0) ? -1 : (ctx->ks_persist > 0 ? 1 : 0);
    out->lsqr_timeouts = ctx->ks_timeouts;
    out->repeated_calls = ctx->admm_repeats;
    out->last_call_wall_ms = ctx->last_call_wall_ms;
    for (int i = 0; i < 4; ++i) out->last_call_stage_ms[i] = ctx->last_call_ms[i];
    for (int i = 0; i < 3; ++i) out->set_denoiser_ms[i] = p.setup_ms[i];
    return QMRI_OK;
}

extern "C" int qmri_denoise(qmri_ctx* ctx, const double* in, int H, int W, int C, int B, double* out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    NetPlan& p = ctx->net;
    if (!p.ready) { qmri_set_error(ctx, "denoiser not set: call qmri_set_denoiser first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, in && out, "in / out must not be NULL");
    if (H != p.H || W != p.W || C != p.desc.in_nc) {
        // MATLAB: images:denoiseImage:incompatibleImageNetwork-style size error from activations()
        qmri_set_error(ctx, "input %d x %d x %d does not match the network input %d x %d x %d", H, W, C, p.H, p.W, p.desc.in_nc);
        return QMRI_ERR_INVALID_ARG;
    }
    QMRI_CHECK_ARG(ctx, B >= 1 && B <= p.maxB, "batch exceeds max_batch of qmri_set_denoiser");
    const size_t HW = (size_t)H * W, nin = HW * C * B, nout = HW * p.desc.out_nc * B;
    // Both architectures are bias-free convolutions + ReLU (+ skips): net(2^k x) = 2^k net(x) exactly in fp32.  Inputs far from unit
    // scale are therefore brought to [0.5, 1) by a power of two on the way in and back on the way out, so that the f16 pieces of
    // the activations stay in their normal range; inputs of ordinary scale -- the [0, 1] images of the
    // ADMM loop -- are left alone.
    float in_scale = 1.f, out_scale = 1.f;
    {
        // (four independent maxima: the loop vectorises; a NaN never raises amax -- such an input goes through unscaled and the range guard sees it)
        double m4[4] = {0.0, 0.0, 0.0, 0.0};
        size_t i = 0;
        for (; i + 4 <= nin; i += 4)
            for (int j = 0; j < 4; ++j) { const double a = std::fabs(in[i + j]); m4[j] = a > m4[j] ? a : m4[j]; }
        for (; i < nin; ++i) { const double a = std::fabs(in[i]); m4[0] = a > m4[0] ? a : m4[0]; }
        const double amax = std::max(std::max(m4[0], m4[1]), std::max(m4[2], m4[3]));
        if (std::isfinite(amax) && amax > 0.0 && (amax < 0.0625 || amax >= 256.0)) {
            int e = 0;
            (void)std::frexp(amax, &e);                             // amax = f * 2^e, f in [0.5, 1)
            const int k = std::min(100, std::max(-100, -e));
            in_scale = std::ldexp(1.f, k); out_scale = std::ldexp(1.f, -k);
        }
    }
    // staging buffer of the per-call drop-in mode (the reference's own PnP_ADMM.m calling param.net 100 times): kept with the plan, grown on demand
    if (p.io_cap < std::max(nin, nout)) {
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p.d_io) { (void)hipFree(p.d_io); p.d_io = nullptr; p.io_cap = 0; }
        QMRI_HIP(ctx, hipMalloc((void**)&p.d_io, std::max(nin, nout) * sizeof(double)));
        p.io_cap = std::max(nin, nout);
    }
    double* const d_io = p.d_io;
    int st = QMRI_OK;
    bool again = false;
    for (int attempt = 0; attempt < 3; ++attempt) {        // (further passes only after a guard changed the plan: see qmri_net_forward_dev)
        again = false;
        do {
            if (hipMemcpyAsync(d_io, in, nin * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { st = QMRI_ERR_HIP; break; }
            if ((st = ew_launch_pack(ctx, B, C, H, W, d_io, 1, p.in32, in_scale)) != QMRI_OK) break;   // im2single: :72-77
            if ((st = net_forward(ctx, B)) != QMRI_OK) break;                                      // activations(...): :88
            if ((st = ew_launch_unpack(ctx, B, p.desc.out_nc, H, W, p.out32, p.in32, p.desc.residual_noise, d_io, 1, out_scale)) != QMRI_OK) break;
            if (hipMemcpyAsync(out, d_io, nout * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { st = QMRI_ERR_HIP; break; }
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) { st = QMRI_ERR_HIP; break; }
            st = net_range_tripped(ctx, again);
        } while (0);
        if (st != QMRI_OK || !again) break;
    }
    if (st == QMRI_OK && again) { qmri_set_error(ctx, "the network's guards asked for a fourth pass (resident-tile hand-off / f16 range): giving up"); st = QMRI_ERR_HIP; }
    if (st == QMRI_ERR_HIP && ctx->err.empty()) qmri_set_error(ctx, "HIP failure in qmri_denoise");
    return st;
}

// ---------------------------------------------------------------------------------------------------
// PnP-ADMM
// ---------------------------------------------------------------------------------------------------
struct StageTimer {
    qmri_ctx* ctx;
    bool on, marks;
    int cur = -1;
    explicit StageTimer(qmri_ctx* c) : ctx(c), on(c->prof_level == 1 || c->prof_level == 2), marks(c->prof_level == 3) {
        c->marks_n = 0;
        for (double& v : c->last_call_ms) v = 0.0;
    }
    void start() {
        if (on) (void)hipEventRecord(ctx->ev[0], ctx->stream);
        if (marks) {
            if (ctx->marks_n + 2 > ctx->marks.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { marks = false; return; }
                ctx->marks.push_back(a); ctx->marks.push_back(b); ctx->mark_kind.push_back(0);
            }
            cur = (int)ctx->marks_n;
            (void)hipEventRecord(ctx->marks[cur], ctx->stream);
        }
    }
    void stop(double& acc) {
        if (on) {
            (void)hipEventRecord(ctx->ev[1], ctx->stream);
            (void)hipEventSynchronize(ctx->ev[1]);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]);
            acc += ms;
            ctx->last_call_ms[kind_of(acc)] += ms;
        }
        if (marks && cur >= 0) {
            (void)hipEventRecord(ctx->marks[cur + 1], ctx->stream);
            ctx->mark_kind[cur / 2] = kind_of(acc);
            ctx->marks_n = (size_t)cur + 2;
            cur = -1;
        }
    }
    int kind_of(const double& acc) const {
        const qmri_profile& p = ctx->prof;
        return (&acc == &p.ms_xupdate) ? 0 : (&acc == &p.ms_denoiser) ? 1 : (&acc == &p.ms_elementwise) ? 2 : 3;
    }
    // after the call's final synchronisation: the marks become stage times (profile and last_call_ms)
    void resolve() {
        if (!marks) return;
        for (size_t i = 0; i + 1 < ctx->marks_n; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->marks[i], ctx->marks[i + 1]) != hipSuccess) continue;
            const int k = ctx->mark_kind[i / 2];
            ctx->last_call_ms[k] += ms;
            (k == 0 ? ctx->prof.ms_xupdate : k == 1 ? ctx->prof.ms_denoiser : k == 2 ? ctx->prof.ms_elementwise : ctx->prof.ms_diag) += ms;
        }
        ctx->marks_n = 0;
    }
};

static int pnp_admm_dev_impl(qmri_ctx* ctx, int nslices, const void* d_y, const qmri_admm_params* prm, const void* d_x0,
                             const void* d_gt, void* d_x_out, double* diag_out, int32_t* lsqr_iters_out);

// (the wall clock of the call, repeats included, for qmri_get_health)
extern "C" int qmri_pnp_admm_dev(qmri_ctx* ctx, int nslices, const void* d_y, const qmri_admm_params* prm, const void* d_x0,
                                 const void* d_gt, void* d_x_out, double* diag_out, int32_t* lsqr_iters_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    const int st = pnp_admm_dev_impl(ctx, nslices, d_y, prm, d_x0, d_gt, d_x_out, diag_out, lsqr_iters_out);
    ctx->last_call_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

static int pnp_admm_dev_impl(qmri_ctx* ctx, int nslices, const void* d_y, const qmri_admm_params* prm, const void* d_x0,
                             const void* d_gt, void* d_x_out, double* diag_out, int32_t* lsqr_iters_out) {
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    NetPlan& net = ctx->net;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    if (!net.ready) { qmri_set_error(ctx, "denoiser not set: call qmri_set_denoiser first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, d_y && prm && d_x_out, "y / params / x_out must not be NULL");
    const int B = nslices;
    QMRI_CHECK_ARG(ctx, B >= 1 && B <= o.maxB && B <= net.maxB, "nslices exceeds max_batch of the operator or the denoiser");
    QMRI_CHECK_ARG(ctx, prm->iters >= 0 && prm->gamma > 0 && prm->cg_maxit >= 0, "iters >= 0, gamma > 0, cg_maxit >= 0 required");
    const int multi = prm->denoiser_type == QMRI_DENOISER_MULTI_LEVEL;
    if (net.H != o.N || net.W != o.M || net.desc.in_nc != o.s + (multi ? 1 : 0) || net.desc.out_nc != o.s) {
        qmri_set_error(ctx, "denoiser (%d x %d, %d -> %d channels) does not fit the operator (%d x %d x %d, %s)", net.H, net.W,
                       net.desc.in_nc, net.desc.out_nc, o.N, o.M, o.s, multi ? "multi_level" : "single_level");
        return QMRI_ERR_INVALID_ARG;
    }
    const OpDev op = qmri_opdev(ctx);
    const size_t plane = (size_t)o.N * o.M, n = plane * o.s, nb = (size_t)B * n * sizeof(double2);
    const double2* y = (const double2*)d_y;
    StageTimer tm(ctx);
    const auto prof_at_entry = ctx->prof;                  // (an attempt the range guard aborts must not stay in the profile)

    QMRI_TRY(dc_launch_sort_y(ctx, op, o.ls, B, y));
    if (d_x0) QMRI_HIP(ctx, hipMemcpyAsync(o.d_x, d_x0, nb, hipMemcpyDeviceToDevice, ctx->stream));        // x = param.X0
    else QMRI_TRY(dc_launch_adj(ctx, op, B, y, o.d_tmp, o.d_x));                    // F.adjoint(Y)
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_vv, o.d_x, nb, hipMemcpyDeviceToDevice, ctx->stream));                  // v = x
    QMRI_HIP(ctx, hipMemsetAsync(o.d_u, 0, nb, ctx->stream));                                                // uold = 0
    if (prm->solver == QMRI_SOLVER_DIRECT) {
        QMRI_TRY(qmri_prepare_direct(ctx, prm->gamma));
        const double2* aty = o.d_x;
        if (d_x0) { QMRI_TRY(dc_launch_adj(ctx, op, B, y, o.d_tmp, o.d_xa)); aty = o.d_xa; }
        QMRI_TRY(dc_launch_fwd(ctx, op, o.ls, DC_SPECTRUM, B, aty, o.d_tmp, o.d_chat, nullptr));
    } else if (prm->solver != QMRI_SOLVER_LSQR) {
        qmri_set_error(ctx, "unknown solver %d", prm->solver);
        return QMRI_ERR_INVALID_ARG;
    }
    if (prm->want_diag && diag_out) {
        if (o.d_diag) { (void)hipFree(o.d_diag); o.d_diag = nullptr; }
        QMRI_HIP(ctx, hipMalloc((void**)&o.d_diag, (size_t)B * std::max(prm->iters, 1) * 2 * sizeof(double)));
    }
    std::vector<int32_t> it_b(B);
    o.xhat_valid = false;                                  // x was just set: its spectrum is not known yet
    const bool diag = prm->want_diag && diag_out;
    bool range_trip = false;
    // LSQR state per (ADMM iteration, slice) in pinned memory: with the one-launch LSQR kernel the host does not wait inside the loop at all
    // (qmri_lsqr_run, "deferred") -- the kernels of all iterations are queued back to back and the counts are read after the final
    // synchronisation; an event or a host round trip per x-update left the GPU idle for ~6 us each
    std::vector<char> deferred_it((size_t)std::max(prm->iters, 1), 0);
    if (prm->solver == QMRI_SOLVER_LSQR && (size_t)prm->iters * B > o.h_ring_cap) {
        if (o.h_ring) { QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(o.h_ring); o.h_ring = nullptr; o.h_ring_cap = 0; }
        const size_t cap = std::max<size_t>((size_t)prm->iters * B, 128);
        QMRI_HIP(ctx, hipHostMalloc((void**)&o.h_ring, cap * sizeof(LsqrState), hipHostMallocDefault));
        o.h_ring_cap = cap;
    }
    // Round 4: the small launches around the network are folded into their neighbours (LSQR solver; knob fuse_ew = 0 restores the separate kernels
    // for A/Bs): un-normalise + dual update + z + the h-pass of z's transform + the forward pass's |output| report = ONE launch (k_dual_fwd_h);
    // the w-pass of z rides in the solve's first kernel (k_ks_init_a<FWDW>); the min / max of real(x + u) come out of the solve's last h-pass.
    const bool fused = qmri_knob(K_FUSE_EW) != 0 && prm->solver == QMRI_SOLVER_LSQR;
    const int hb = dc_hpass_blocks(op);
    bool z_in_tmp = false;                                 // o.d_tmp holds the h-pass of z (and ls.pz hb partial sums per slice)
    struct DeferGuard { NetPlan& n; ~DeferGuard() { n.act_defer = false; n.act_pending_valid = false; } } defer_guard{net};
    net.act_defer = fused;
    for (int it = 0; it < prm->iters; ++it) {
        // Step 1 (PnP_ADMM.m:102): x = argmin ||y - Ax||^2 + r ||x - (v - uold)||^2
        tm.start();
        if (it == 0) QMRI_TRY(dc_launch_prepare_z(ctx, op, o.ls, B, o.d_vv, o.d_u, o.d_z));   // later: fused into the dual update
        if (prm->solver == QMRI_SOLVER_LSQR) {
            bool deferred = false;
            LsqrFuse lf;
            if (fused) { lf.z_hpass_nblk = z_in_tmp ? hb : 0; lf.mm_u = o.d_u; lf.mm = o.d_mm; }
            QMRI_TRY(qmri_lsqr_run(ctx, B, o.d_z, prm->gamma, prm->cg_tol, prm->cg_maxit, o.d_x, it_b.data(), nullptr,
                                   diag ? o.d_pd : nullptr,            // (the data-fidelity partials come with the solve)
                                   o.h_ring + (size_t)it * B, &deferred, &lf));
            deferred_it[it] = deferred ? 1 : 0;
            if (!deferred && lsqr_iters_out) for (int b = 0; b < B; ++b) lsqr_iters_out[(size_t)b * prm->iters + it] = it_b[b];
            // The range guard of earlier forwards is on the host (pinned words written by k_act_check).  After a wait inside qmri_lsqr_run (the
            // two-launch iteration) it is current up to the previous iteration; without one it is whatever has arrived.  A tripped guard ends
            // this attempt at once instead of after all iterations.
            if (it > 0 && net.sp6 == 2 && host_range_tripped(net)) { range_trip = true; tm.stop(ctx->prof.ms_xupdate); break; }
        } else {
            QMRI_TRY(dc_launch_direct(ctx, op, B, o.d_z, o.d_chat, prm->gamma, o.d_tmp, o.d_x));
            if (lsqr_iters_out) for (int b = 0; b < B; ++b) lsqr_iters_out[(size_t)b * prm->iters + it] = 0;
        }
        tm.stop(ctx->prof.ms_xupdate);
        if (prm->want_diag && diag_out) {                                                                    // PnP_ADMM.m:106-109
            tm.start();
            if (prm->solver != QMRI_SOLVER_LSQR) QMRI_TRY(dc_launch_fwd(ctx, op, o.ls, DC_DIAG, B, o.d_x, o.d_tmp, nullptr, o.d_pd));
            QMRI_TRY(ew_launch_diag(ctx, op, o.ls, B, o.d_x, (const double2*)d_gt, o.d_pd, o.d_diag, prm->iters, it));
            tm.stop(ctx->prof.ms_diag);
        }
        // Step 2 (PnP_ADMM.m:115-138): v = real(x+uold) -> [0,1] -> net -> undo
        tm.start();
        QMRI_TRY(ew_launch_minmax_normalise(ctx, B, n, (int)plane, o.N, o.s, multi, prm->noise_std, o.d_x, o.d_u, o.d_mm, o.d_norm,
                                            fused ? hb : o.ls.nblk_z, net.in32, fused /* the partial min / max came with the solve's last h-pass */));
        tm.stop(ctx->prof.ms_elementwise);
        tm.start();
        QMRI_TRY(net_forward(ctx, B));
        // the range guard of this forward is read at the next x-update's synchronisation point (or at the end): k_act_check has written it
        // to the pinned host words; only a network with more layers than words copies the device flag
        if (net.sp6 == 2 && (int)net.layers.size() + 1 > net.h_range_words)
            QMRI_HIP(ctx, hipMemcpyAsync(net.h_range_flag, net.d_range_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        tm.stop(ctx->prof.ms_denoiser);
        // Step 3 (PnP_ADMM.m:138,144): v = I*range + min ; uold = uold + x - v
        tm.start();
        if (fused) {
            const DualArgs da = {net.out32.base1(), net.in32.base1(), net.out32.hp, (int)net.out32.plane(), net.out32.batch_stride(), net.in32.batch_stride(),
                                 net.desc.residual_noise, o.d_norm, o.d_x, o.d_u, o.ls.pz};
            ActCheckArgs ac{};
            if (net.act_pending_valid) { ac = net.act_pending; net.act_pending_valid = false; }
            QMRI_TRY(dc_launch_dual_fwd_h(ctx, op, B, da, ac, o.d_tmp));
            z_in_tmp = true;
        } else {
            QMRI_TRY(ew_launch_unnormalise_dual(ctx, B, n, (int)plane, o.N, net.out32, net.in32, net.desc.residual_noise, o.d_norm, o.d_x, o.d_u,
                                                nullptr /* v itself is never read again: z = v - u goes to the next x-update */, o.d_z, o.ls.pz, o.ls.nblk_z));
        }
        tm.stop(ctx->prof.ms_elementwise);
        ctx->prof.admm_iters += 1;
    }
    if (!range_trip) {
        QMRI_HIP(ctx, hipMemcpyAsync(d_x_out, o.d_x, nb, hipMemcpyDeviceToDevice, ctx->stream));             // returns x, not v
    }
    if (prm->want_diag && diag_out && prm->iters > 0 && !range_trip)
        QMRI_HIP(ctx, hipMemcpyAsync(diag_out, o.d_diag, (size_t)B * prm->iters * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    tm.resolve();                                                  // (profile level 3: the stage marks of this call)
    QMRI_TRY(qmri_prof_chain_finish(ctx));                         // (profile level 2: the LSQR launches since the last forward pass)
    if (prm->solver == QMRI_SOLVER_LSQR && !range_trip) {          // LSQR counts of the iterations whose state was deferred; a timed-out one-launch kernel
        bool timed_out = false;
        for (int it = 0; it < prm->iters; ++it) {
            if (!deferred_it[it]) continue;
            for (int b = 0; b < B; ++b) {
                const LsqrState& h = o.h_ring[(size_t)it * B + b];
                if (h.flag == 77) timed_out = true;
                const int n_it = h.done ? h.iter : prm->cg_maxit;
                if (lsqr_iters_out) lsqr_iters_out[(size_t)b * prm->iters + it] = n_it;
                ctx->prof.lsqr_iters += n_it;
            }
        }
        if (timed_out) {                                           // (never seen) everything after it is garbage: once more with the two-launch iteration
            fprintf(stderr, "libqmri: the one-launch LSQR timed out waiting for a partial sum; repeating the reconstruction with the two-launch iteration\n");
            ctx->ks_persist = 0;
            ctx->ks_timeouts += 1; ctx->admm_repeats += 1;
            ctx->prof = prof_at_entry;
            return pnp_admm_dev_impl(ctx, nslices, d_y, prm, d_x0, d_gt, d_x_out, diag_out, lsqr_iters_out);
        }
    }
    if (prm->iters > 0) {
        // f16 range guard: the network now runs on the bf16 scheme; the inputs are untouched (d_x_out must not alias d_x0), run again
        bool again = false;
        QMRI_TRY(net_range_tripped(ctx, again));
        if (again) {
            ctx->prof = prof_at_entry;                     // the repeated run is the one that counts
            ctx->admm_repeats += 1;
            return pnp_admm_dev_impl(ctx, nslices, d_y, prm, d_x0, d_gt, d_x_out, diag_out, lsqr_iters_out);
        }
    }
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------
// Multi-coil extension (no reference counterpart: README.md:63 -- parity unpinned; mc_kernels.hip): the x-update and the PnP-ADMM loop of
// PnP_ADMM.m:76-146 with A replaced by the SENSE operator of qmri_set_coils.  One slice; host arrays in, host arrays out.
// ---------------------------------------------------------------------------------------------------
int qmri_lsqr_mc_dev(qmri_ctx* ctx, const double2* d_y, const double2* d_z, double r, double tol, int maxit, double2* d_x, int32_t* iters_out, int32_t* flag_out);

struct McStage {                                   // device copies of one multi-coil problem
    double2 *y = nullptr, *z = nullptr, *x = nullptr;
    ~McStage() { if (y) (void)hipFree(y); if (z) (void)hipFree(z); if (x) (void)hipFree(x); }
};

extern "C" int qmri_xupdate_mc(qmri_ctx* ctx, const void* y_mc, const void* z, double r, double tol, int maxit, const void* x0, void* x_out,
                               int32_t* iters_out, int32_t* flag_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    if (!o.ncoil) { qmri_set_error(ctx, "no coil maps set: call qmri_set_coils first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, y_mc && z && x_out && r > 0 && maxit >= 0, "y / z / x_out must not be NULL, r > 0, maxit >= 0");
    const size_t n = (size_t)o.N * o.M * o.s, mtot = (size_t)o.ncoil * o.m;
    McStage st;
    QMRI_HIP(ctx, hipMalloc((void**)&st.y, mtot * sizeof(double2)));
    QMRI_HIP(ctx, hipMalloc((void**)&st.z, n * sizeof(double2)));
    QMRI_HIP(ctx, hipMalloc((void**)&st.x, n * sizeof(double2)));
    QMRI_HIP(ctx, hipMemcpyAsync(st.y, y_mc, mtot * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    QMRI_HIP(ctx, hipMemcpyAsync(st.z, z, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    if (x0) QMRI_HIP(ctx, hipMemcpyAsync(st.x, x0, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    else QMRI_HIP(ctx, hipMemsetAsync(st.x, 0, n * sizeof(double2), ctx->stream));
    QMRI_TRY(qmri_lsqr_mc_dev(ctx, st.y, st.z, r, tol, maxit, st.x, iters_out, flag_out));
    QMRI_HIP(ctx, hipMemcpyAsync(x_out, st.x, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

extern "C" int qmri_pnp_admm_mc(qmri_ctx* ctx, const void* y_mc, const qmri_admm_params* prm, const void* x0, void* x_out, int32_t* lsqr_iters_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    NetPlan& net = ctx->net;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    if (!o.ncoil) { qmri_set_error(ctx, "no coil maps set: call qmri_set_coils first"); return QMRI_ERR_STATE; }
    if (!net.ready) { qmri_set_error(ctx, "denoiser not set: call qmri_set_denoiser first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, y_mc && prm && x_out, "y / params / x_out must not be NULL");
    QMRI_CHECK_ARG(ctx, prm->iters >= 0 && prm->gamma > 0 && prm->cg_maxit >= 0 && prm->solver == QMRI_SOLVER_LSQR, "iters >= 0, gamma > 0, cg_maxit >= 0, LSQR solver required");
    const int multi = prm->denoiser_type == QMRI_DENOISER_MULTI_LEVEL;
    if (net.H != o.N || net.W != o.M || net.desc.in_nc != o.s + (multi ? 1 : 0) || net.desc.out_nc != o.s) {
        qmri_set_error(ctx, "denoiser (%d x %d, %d -> %d channels) does not fit the operator (%d x %d x %d)", net.H, net.W, net.desc.in_nc, net.desc.out_nc, o.N, o.M, o.s);
        return QMRI_ERR_INVALID_ARG;
    }
    const size_t plane = (size_t)o.N * o.M, n = plane * o.s, mtot = (size_t)o.ncoil * o.m;
    McStage st;
    QMRI_HIP(ctx, hipMalloc((void**)&st.y, mtot * sizeof(double2)));
    QMRI_HIP(ctx, hipMalloc((void**)&st.x, n * sizeof(double2)));
    QMRI_HIP(ctx, hipMemcpyAsync(st.y, y_mc, mtot * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    double2* scr = nullptr;                                        // [max_batch][n] coil images of the initial adjoint
    QMRI_HIP(ctx, hipMalloc((void**)&scr, (size_t)o.maxB * n * sizeof(double2)));
    struct Scr { double2* p; ~Scr() { if (p) (void)hipFree(p); } } scr_guard{scr};
    if (x0) QMRI_HIP(ctx, hipMemcpyAsync(st.x, x0, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    else {                                                         // x = F.adjoint(Y)  (PnP_ADMM.m:84)
        for (int j0 = 0; j0 < o.ncoil; j0 += o.maxB) {
            const int cnt = std::min(o.maxB, o.ncoil - j0);
            QMRI_TRY(dc_launch_adj(ctx, qmri_opdev(ctx), cnt, st.y + (size_t)j0 * o.m, o.d_tmp, scr));
            QMRI_TRY(ew_launch_coil_sum(ctx, n, plane, cnt, scr, o.d_coils + (size_t)j0 * plane, st.x, j0 > 0));
        }
    }
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_vv, st.x, n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));          // v = x
    QMRI_HIP(ctx, hipMemsetAsync(o.d_u, 0, n * sizeof(double2), ctx->stream));                                          // uold = 0
    QMRI_TRY(dc_launch_prepare_z(ctx, qmri_opdev(ctx), o.ls, 1, o.d_vv, o.d_u, o.d_z));                                 // z = v - uold
    bool again = false;
    for (int it = 0; it < prm->iters; ++it) {
        int32_t li = 0;
        QMRI_TRY(qmri_lsqr_mc_dev(ctx, st.y, o.d_z, prm->gamma, prm->cg_tol, prm->cg_maxit, st.x, &li, nullptr));      // PnP_ADMM.m:102
        if (lsqr_iters_out) lsqr_iters_out[it] = li;
        QMRI_TRY(ew_launch_minmax_normalise(ctx, 1, n, (int)plane, o.N, o.s, multi, prm->noise_std, st.x, o.d_u, o.d_mm, o.d_norm, o.ls.nblk_z, net.in32, false));
        QMRI_TRY(net_forward(ctx, 1));
        QMRI_TRY(ew_launch_unnormalise_dual(ctx, 1, n, (int)plane, o.N, net.out32, net.in32, net.desc.residual_noise, o.d_norm, st.x, o.d_u, nullptr, o.d_z,
                                            o.ls.pz, o.ls.nblk_z));
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        QMRI_TRY(net_range_tripped(ctx, again));                   // (f16 range / hand-off guards: the network is re-packed or the form switched; repeat from the start)
        if (again) break;
    }
    if (again) { ctx->admm_repeats += 1; return qmri_pnp_admm_mc(ctx, y_mc, prm, x0, x_out, lsqr_iters_out); }
    QMRI_HIP(ctx, hipMemcpyAsync(x_out, st.x, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));                // returns x, not v
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    o.xhat_valid = false;
    return QMRI_OK;
}

extern "C" int qmri_pnp_admm(qmri_ctx* ctx, const void* y, const qmri_admm_params* p, const void* x0, const void* gt,
                             void* x_out, double* diag_out, int32_t* lsqr_iters_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, y && p && x_out, "y / params / x_out must not be NULL");
    const size_t n = (size_t)o.N * o.M * o.s;
    double2* d_gt = nullptr;
    double2* d_x0 = nullptr;
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_ya, y, (size_t)o.m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    if (x0) { d_x0 = o.d_xb; QMRI_HIP(ctx, hipMemcpyAsync(d_x0, x0, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream)); }
    if (gt) {
        QMRI_HIP(ctx, hipMalloc((void**)&d_gt, n * sizeof(double2)));
        if (hipMemcpyAsync(d_gt, gt, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            (void)hipFree(d_gt);
            qmri_set_error(ctx, "copy of gt_tsmi to the device failed");
            return QMRI_ERR_HIP;
        }
    }
    int st = qmri_pnp_admm_dev(ctx, 1, o.d_ya, p, d_x0, d_gt, o.d_xa, diag_out, lsqr_iters_out);
    if (st == QMRI_OK) {
        if (hipMemcpyAsync(x_out, o.d_xa, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            qmri_set_error(ctx, "copy of the result to the host failed");
            st = QMRI_ERR_HIP;
        }
    }
    if (d_gt) (void)hipFree(d_gt);
    return st;
}

// A slice stack from host buffers through ONE context (the MATLAB route for `PnP_ADMM_hip(Y, param)` with a measurement matrix): the slices
// advance slices_per_launch at a time through qmri_pnp_admm_dev.  Plain and synchronous -- copy in, reconstruct, copy out per launch;
// qmri_recon_batch is the pipelined, multi-device form of the same work.
extern "C" int qmri_pnp_admm_batch(qmri_ctx* ctx, int nslices, int slices_per_launch, const void* y, const qmri_admm_params* p, const void* x0,
                                   const void* gt, void* x_out, double* diag_out, int32_t* lsqr_iters_out) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    OpHost& o = ctx->op;
    if (!o.ready) { qmri_set_error(ctx, "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; }
    if (!ctx->net.ready) { qmri_set_error(ctx, "denoiser not set: call qmri_set_denoiser first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, y && p && x_out && nslices >= 1 && slices_per_launch >= 1, "y / params / x_out must not be NULL, nslices and slices_per_launch >= 1");
    const int spl = std::min(slices_per_launch, nslices);
    QMRI_CHECK_ARG(ctx, spl <= o.maxB && spl <= ctx->net.maxB, "slices_per_launch exceeds max_batch of the operator or the denoiser");
    const size_t n = (size_t)o.N * o.M * o.s, m = (size_t)o.m, it = (size_t)std::max(p->iters, 0);
    double2 *dY = nullptr, *dX = nullptr, *dX0 = nullptr, *dGT = nullptr;
    int st = QMRI_OK;
    auto fail = [&](const char* what) { qmri_set_error(ctx, "%s failed in qmri_pnp_admm_batch", what); st = QMRI_ERR_HIP; };
    do {
        if (hipMalloc((void**)&dY, spl * m * sizeof(double2)) != hipSuccess || hipMalloc((void**)&dX, spl * n * sizeof(double2)) != hipSuccess ||
            (x0 && hipMalloc((void**)&dX0, spl * n * sizeof(double2)) != hipSuccess) || (gt && hipMalloc((void**)&dGT, spl * n * sizeof(double2)) != hipSuccess)) {
            qmri_set_error(ctx, "hipMalloc failed in qmri_pnp_admm_batch"); st = QMRI_ERR_NOMEM; break;
        }
        for (int s0 = 0; s0 < nslices && st == QMRI_OK; s0 += spl) {
            const size_t cnt = (size_t)std::min(spl, nslices - s0);
            if (hipMemcpyAsync(dY, (const double2*)y + (size_t)s0 * m, cnt * m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
            if (x0 && hipMemcpyAsync(dX0, (const double2*)x0 + (size_t)s0 * n, cnt * n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
            if (gt && hipMemcpyAsync(dGT, (const double2*)gt + (size_t)s0 * n, cnt * n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
            st = qmri_pnp_admm_dev(ctx, (int)cnt, dY, p, dX0, dGT, dX, diag_out ? diag_out + (size_t)s0 * it * 2 : nullptr,
                                   lsqr_iters_out ? lsqr_iters_out + (size_t)s0 * it : nullptr);
            if (st != QMRI_OK) break;
            if (hipMemcpyAsync((double2*)x_out + (size_t)s0 * n, dX, cnt * n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        }
    } while (0);
    void* ptrs[] = {dY, dX, dX0, dGT};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    return st;
}

// ---------------------------------------------------------------------------------------------------
// dictionary
// ---------------------------------------------------------------------------------------------------
void qmri_free_dict(qmri_ctx* ctx) {
    DictHost& d = ctx->dict;
    if (d.d_pack) (void)hipFree(d.d_pack);
    if (d.d_pack16) (void)hipFree(d.d_pack16);
    if (d.d_gmax) (void)hipFree(d.d_gmax);
    if (d.d_normD) (void)hipFree(d.d_normD);
    if (d.d_lut) (void)hipFree(d.d_lut);
    if (d.d_part) (void)hipFree(d.d_part);
    if (d.d_xp) (void)hipFree(d.d_xp);
    if (d.d_win) (void)hipFree(d.d_win);
    const int filter_on = d.filter_on; const float margin_scale = d.margin_scale;
    d = DictHost();
    d.filter_on = filter_on; d.margin_scale = margin_scale;
}

extern "C" int qmri_set_dictionary(qmri_ctx* ctx, int K, int s, int Q, const float* D, const float* normD, const float* lut) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_CHECK_ARG(ctx, D && normD && lut, "D / normD / lut must not be NULL");
    QMRI_CHECK_ARG(ctx, K > 0 && s > 0 && Q > 0, "K, s, Q must be positive");
    if (s > 1024) { qmri_set_error(ctx, "dictionary match supports s <= 1024 channels (got %d)", s); return QMRI_ERR_UNSUPPORTED; }
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    qmri_free_dict(ctx);
    DictHost& d = ctx->dict;
    d.K = K; d.s = s; d.Q = Q;
    if (s > 16) {
        // wide dictionaries (uncompressed fingerprints, s = T; mrf_dtm_cpu.m:41-50 is T-generic): channel-blocked GEMM, dictw_kernels.hip
        d.wide = 1;
        int st = dictw_pack_dictionary(ctx, D, K, s);
        if (st == QMRI_OK) st = dev_alloc(ctx, &d.d_normD, (size_t)K);
        if (st == QMRI_OK) st = dev_alloc(ctx, &d.d_lut, (size_t)K * Q);
        if (st != QMRI_OK) { qmri_free_dict(ctx); return st; }
        QMRI_HIP(ctx, hipMemcpy(d.d_normD, normD, (size_t)K * sizeof(float), hipMemcpyHostToDevice));
        QMRI_HIP(ctx, hipMemcpy(d.d_lut, lut, (size_t)K * Q * sizeof(float), hipMemcpyHostToDevice));
        QMRI_HIP(ctx, hipDeviceSynchronize());                    // (blocking copies on the NULL stream; this context's stream is not ordered with it)
        d.ready = true;
        return QMRI_OK;
    }
    d.ntiles = (K + 31) / 32;
    const int npair = (s + 1) / 2;
    // [tile][lane][NPL] with NPL = 4 or 8 floats per lane (its A-fragment value of every channel pair, zero padded): a lane fetches its
    // share of a tile with one or two 16-byte requests (dict_kernels.hip)
    const int npl = (npair <= 4) ? 4 : 8;
    std::vector<float> pack((size_t)d.ntiles * 64 * npl, 0.f);
    for (int t = 0; t < d.ntiles; ++t)
        for (int q = 0; q < npair; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int atom = t * 32 + (lane & 31), c = 2 * q + (lane >> 5);
                if (atom < K && c < s) pack[((size_t)t * 64 + lane) * npl + q] = D[(size_t)atom + (size_t)K * c];
            }
    QMRI_TRY(dev_alloc(ctx, &d.d_pack, pack.size()));
    QMRI_TRY(dev_alloc(ctx, &d.d_normD, (size_t)K));
    QMRI_TRY(dev_alloc(ctx, &d.d_lut, (size_t)K * Q));
    QMRI_HIP(ctx, hipMemcpy(d.d_pack, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(d.d_normD, normD, (size_t)K * sizeof(float), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(d.d_lut, lut, (size_t)K * Q * sizeof(float), hipMemcpyHostToDevice));
    // f16 pieces for the filter: a = g D in (-1, 1) with one power of two g, hi = f16(a), lo = f16(a - hi) (the difference is exact in f32)
    {
        float dmax = 0.f; double r2max = 0.0; bool finite = true;
        for (int a = 0; a < K && finite; ++a) {
            double r2 = 0.0;
            for (int c = 0; c < s; ++c) {
                const float v = D[(size_t)a + (size_t)K * c];
                if (!std::isfinite(v)) { finite = false; break; }
                dmax = std::max(dmax, std::fabs(v)); r2 += (double)v * v;
            }
            r2max = std::max(r2max, r2);
        }
        if (finite && dmax > 1e-30f && dmax < 1e30f) {
            int e = 0; (void)std::frexp(dmax, &e);
            const float g = std::ldexp(1.f, -e);                            // g dmax in [0.5, 1)
            std::vector<_Float16> p16((size_t)d.ntiles * 64 * 16, (_Float16)0.f);
            for (int t = 0; t < d.ntiles; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int atom = t * 32 + (lane & 31);
                    _Float16* hi = &p16[((size_t)t * 128 + lane) * 8], *lo = hi + 64 * 8;     // [tile][hi | lo][lane][8]
                    for (int jj = 0; jj < 8; ++jj) {
                        const int c = 8 * (lane >> 5) + jj;
                        if (atom >= K || c >= s) continue;
                        const float a = D[(size_t)atom + (size_t)K * c] * g;
                        hi[jj] = (_Float16)a; lo[jj] = (_Float16)(a - (float)hi[jj]);
                    }
                }
            QMRI_HIP(ctx, hipMalloc((void**)&d.d_pack16, p16.size() * sizeof(_Float16)));
            QMRI_HIP(ctx, hipMemcpy(d.d_pack16, p16.data(), p16.size() * sizeof(_Float16), hipMemcpyHostToDevice));
            d.marg_coef = (float)(std::ldexp(1.0, -14) * r2max * (double)g * (double)g * 1.001);
        }
    }
    QMRI_HIP(ctx, hipDeviceSynchronize());                        // (blocking copies on the NULL stream; this context's stream is not ordered with it)
    d.ready = true;
    return QMRI_OK;
}

extern "C" int qmri_debug_dict_filter(qmri_ctx* ctx, int on, float margin_scale) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_CHECK_ARG(ctx, margin_scale >= 0.f, "margin_scale must be >= 0");
    ctx->dict.filter_on = on ? 1 : 0;
    ctx->dict.margin_scale = margin_scale;
    return QMRI_OK;
}

extern "C" int qmri_dict_match_xfit_dev(qmri_ctx* ctx, const void* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float* d_xfit) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->dict.ready) { qmri_set_error(ctx, "dictionary not set: call qmri_set_dictionary first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, d_X && Npix > 0, "X must not be NULL and Npix > 0");
    return dict_launch(ctx, (const double2*)d_X, Npix, d_qmap, d_pd, d_mt, d_dm, (float2*)d_xfit);
}

extern "C" int qmri_dict_match_dev(qmri_ctx* ctx, const void* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm) {
    return qmri_dict_match_xfit_dev(ctx, d_X, Npix, d_qmap, d_pd, d_mt, d_dm, nullptr);
}

extern "C" int qmri_dict_match_xfit(qmri_ctx* ctx, const void* X, int Npix, float* qmap, float* pd, float* mt, int32_t* dm, float* xfit) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->dict.ready) { qmri_set_error(ctx, "dictionary not set: call qmri_set_dictionary first"); return QMRI_ERR_STATE; }
    QMRI_CHECK_ARG(ctx, X && Npix > 0, "X must not be NULL and Npix > 0");
    const DictHost& d = ctx->dict;
    const size_t nx = (size_t)Npix * d.s;
    double2* dX = nullptr; float* dq = nullptr; float* dp = nullptr; float* dmt = nullptr; int32_t* ddm = nullptr; float2* dxf = nullptr;
    int st = QMRI_OK;
    auto fail = [&](const char* what) { qmri_set_error(ctx, "%s failed in qmri_dict_match", what); st = QMRI_ERR_HIP; };
    do {
        if (hipMalloc((void**)&dX, nx * sizeof(double2)) != hipSuccess) { fail("hipMalloc"); break; }
        if (qmap && hipMalloc((void**)&dq, (size_t)Npix * d.Q * sizeof(float)) != hipSuccess) { fail("hipMalloc"); break; }
        if (pd && hipMalloc((void**)&dp, (size_t)Npix * 2 * sizeof(float)) != hipSuccess) { fail("hipMalloc"); break; }
        if (mt && hipMalloc((void**)&dmt, (size_t)Npix * sizeof(float)) != hipSuccess) { fail("hipMalloc"); break; }
        if (dm && hipMalloc((void**)&ddm, (size_t)Npix * sizeof(int32_t)) != hipSuccess) { fail("hipMalloc"); break; }
        if (xfit && hipMalloc((void**)&dxf, nx * sizeof(float2)) != hipSuccess) { fail("hipMalloc"); break; }
        if (hipMemcpyAsync(dX, X, nx * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { fail("H2D copy"); break; }
        if ((st = dict_launch(ctx, dX, Npix, dq, dp, dmt, ddm, dxf)) != QMRI_OK) break;
        if (qmap && hipMemcpyAsync(qmap, dq, (size_t)Npix * d.Q * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (pd && hipMemcpyAsync(pd, dp, (size_t)Npix * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (mt && hipMemcpyAsync(mt, dmt, (size_t)Npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (dm && hipMemcpyAsync(dm, ddm, (size_t)Npix * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (xfit && hipMemcpyAsync(xfit, dxf, nx * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { fail("D2H copy"); break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { fail("synchronize"); break; }
    } while (0);
    void* ptrs[] = { dX, dq, dp, dmt, ddm, dxf };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    return st;
}

extern "C" int qmri_dict_match(qmri_ctx* ctx, const void* X, int Npix, float* qmap, float* pd, float* mt, int32_t* dm) {
    return qmri_dict_match_xfit(ctx, X, Npix, qmap, pd, mt, dm, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// slice batches over several GPUs: one host thread + one context per device, static round-robin of launches
// (slices are independent, no collective)
// ---------------------------------------------------------------------------------------------------
// Round 4: the worker no longer waits for its copies.  Every launch (slices_per_launch slices) has one of two sets of device and PINNED host
// buffers.  After the reconstruction of launch k (qmri_pnp_admm_dev returns synchronised) the dictionary matches of its slices are queued on the
// compute stream and the results (x, maps) are copied to the pinned set on a COPY stream behind an event; the host then moves the PREVIOUS launch's
// results from its pinned set into the caller's (pageable) arrays while the device works, and goes on to launch k + 1, whose kernels overlap the
// copies of launch k.  Before: pageable hipMemcpy of 8 MB per slice plus a synchronise and two small copies per slice, all in series with the compute.
// shared_device (round 6): another worker of this call uses the same GPU.  The launches that need the device to themselves -- the one-launch LSQR
// iteration (one workgroup per CU, every unit resident at once) and the resident-tile convolution launch -- would then be partially resident
// side by side, both would wait to their time-outs and the reconstruction would be repeated: such a worker starts on the two-launch iteration
// and one launch per layer (same bits, tested).
static int recon_worker(int device, bool shared_device, int widx, int nworkers, int nslices, const qmri_problem* pb, const char* Y, char* X_out,
                        float* qmap_out, float* pd_out, std::string* err) {
    qmri_ctx* ctx = nullptr;
    int st = qmri_create(device, &ctx);
    if (st != QMRI_OK) { *err = qmri_last_error(nullptr); return st; }
    const int spl = std::max(1, pb->slices_per_launch);
    const size_t n = (size_t)pb->N * pb->M * pb->s, npix = (size_t)pb->N * pb->M;
    const int m = pb->frame_ptr[pb->T];
    const int Q = std::max(pb->Q, 1);
    const bool maps = pb->K > 0 && (qmap_out || pd_out);
    const size_t by = (size_t)spl * m * sizeof(double2), bx = (size_t)spl * n * sizeof(double2);
    const size_t bq = (size_t)spl * npix * Q * sizeof(float), bp = (size_t)spl * npix * 2 * sizeof(float);
    struct Set { double2 *dY = nullptr, *dX = nullptr; float *dq = nullptr, *dp = nullptr; char *hY = nullptr, *hX = nullptr; float *hq = nullptr, *hp = nullptr;
                 hipEvent_t matched = nullptr, copied = nullptr; int s0 = -1, cnt = 0; } set[2];
    hipStream_t cs = nullptr;
    auto bail = [&](int code) { *err = qmri_last_error(ctx); return code; };
    auto hipfail = [&](const char* what) { *err = std::string(what) + " failed in qmri_recon_batch"; st = QMRI_ERR_HIP; };
    // launch held by set `S` -> the caller's arrays (its copies have been queued; wait for them, then plain host copies)
    auto drain = [&](Set& S) {
        if (S.s0 < 0) return;
        if (hipEventSynchronize(S.copied) != hipSuccess) { hipfail("hipEventSynchronize"); return; }
        std::memcpy(X_out + (size_t)S.s0 * n * sizeof(double2), S.hX, (size_t)S.cnt * n * sizeof(double2));
        if (maps && qmap_out) std::memcpy(qmap_out + (size_t)S.s0 * npix * pb->Q, S.hq, (size_t)S.cnt * npix * pb->Q * sizeof(float));
        if (maps && pd_out) std::memcpy(pd_out + (size_t)S.s0 * npix * 2, S.hp, (size_t)S.cnt * npix * 2 * sizeof(float));
        S.s0 = -1;
    };
    do {
        if ((st = qmri_set_operator(ctx, pb->N, pb->M, pb->s, pb->T, pb->V, pb->frame_ptr, pb->kidx, spl)) != QMRI_OK) { bail(st); break; }
        if ((st = qmri_set_denoiser(ctx, pb->net, pb->weights, pb->weights_nbytes, pb->N, pb->M, spl)) != QMRI_OK) { bail(st); break; }
        if (pb->K > 0 && (st = qmri_set_dictionary(ctx, pb->K, pb->s, pb->Q, pb->D, pb->normD, pb->lut)) != QMRI_OK) { bail(st); break; }
        if (shared_device) {
            if ((st = qmri_debug_lsqr_persist(ctx, 0)) != QMRI_OK) { bail(st); break; }
            if ((st = qmri_debug_conv_resident(ctx, 0, nullptr)) != QMRI_OK) { bail(st); break; }
        }
        bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
        for (int j = 0; j < 2 && ok; ++j) {
            Set& S = set[j];
            ok = hipMalloc((void**)&S.dY, by) == hipSuccess && hipMalloc((void**)&S.dX, bx) == hipSuccess && hipHostMalloc((void**)&S.hY, by, hipHostMallocDefault) == hipSuccess &&
                 hipHostMalloc((void**)&S.hX, bx, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&S.matched, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&S.copied, hipEventDisableTiming) == hipSuccess;
            if (ok && maps) ok = hipMalloc((void**)&S.dq, bq) == hipSuccess && hipMalloc((void**)&S.dp, bp) == hipSuccess &&
                                 hipHostMalloc((void**)&S.hq, bq, hipHostMallocDefault) == hipSuccess && hipHostMalloc((void**)&S.hp, bp, hipHostMallocDefault) == hipSuccess;
        }
        if (!ok) { *err = "allocation failed in qmri_recon_batch"; st = QMRI_ERR_NOMEM; break; }
        const int nlaunch = (nslices + spl - 1) / spl;
        int k = 0;
        for (int l = widx; l < nlaunch && st == QMRI_OK; l += nworkers, ++k) {
            Set& S = set[k & 1];
            drain(S);                                              // (its previous launch, two launches ago: long since copied)
            if (st != QMRI_OK) break;
            const int s0 = l * spl, cnt = std::min(spl, nslices - s0);
            std::memcpy(S.hY, Y + (size_t)s0 * m * sizeof(double2), (size_t)cnt * m * sizeof(double2));
            if (hipMemcpyAsync(S.dY, S.hY, (size_t)cnt * m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { hipfail("H2D copy"); break; }
            if ((st = qmri_pnp_admm_dev(ctx, cnt, S.dY, &pb->admm, nullptr, nullptr, S.dX, nullptr, nullptr)) != QMRI_OK) { bail(st); break; }
            if (maps) {
                for (int i = 0; i < cnt && st == QMRI_OK; ++i)
                    if ((st = qmri_dict_match_dev(ctx, S.dX + (size_t)i * n, (int)npix, qmap_out ? S.dq + (size_t)i * npix * Q : nullptr,
                                                  pd_out ? S.dp + (size_t)i * npix * 2 : nullptr, nullptr, nullptr)) != QMRI_OK) bail(st);
                if (st != QMRI_OK) break;
            }
            if (hipEventRecord(S.matched, ctx->stream) != hipSuccess || hipStreamWaitEvent(cs, S.matched, 0) != hipSuccess) { hipfail("event"); break; }
            if (hipMemcpyAsync(S.hX, S.dX, (size_t)cnt * n * sizeof(double2), hipMemcpyDeviceToHost, cs) != hipSuccess) { hipfail("D2H copy"); break; }
            if (maps && qmap_out && hipMemcpyAsync(S.hq, S.dq, (size_t)cnt * npix * Q * sizeof(float), hipMemcpyDeviceToHost, cs) != hipSuccess) { hipfail("D2H copy"); break; }
            if (maps && pd_out && hipMemcpyAsync(S.hp, S.dp, (size_t)cnt * npix * 2 * sizeof(float), hipMemcpyDeviceToHost, cs) != hipSuccess) { hipfail("D2H copy"); break; }
            if (hipEventRecord(S.copied, cs) != hipSuccess) { hipfail("event"); break; }
            S.s0 = s0; S.cnt = cnt;
            drain(set[(k & 1) ^ 1]);                               // the previous launch's results, while the device matches and copies this one's
        }
        if (st == QMRI_OK) { drain(set[0]); if (st == QMRI_OK) drain(set[1]); }
    } while (0);
    (void)hipDeviceSynchronize();
    for (Set& S : set) {
        void* dptr[] = { S.dY, S.dX, S.dq, S.dp };
        for (void* p : dptr) if (p) (void)hipFree(p);
        void* hptr[] = { S.hY, S.hX, S.hq, S.hp };
        for (void* p : hptr) if (p) (void)hipHostFree(p);
        if (S.matched) (void)hipEventDestroy(S.matched);
        if (S.copied) (void)hipEventDestroy(S.copied);
    }
    if (cs) (void)hipStreamDestroy(cs);
    qmri_destroy(ctx);
    return st;
}

extern "C" int qmri_recon_batch(int ndev, const int* devs, int nslices, const qmri_problem* prob, const void* Y, void* X_out,
                                float* qmap_out, float* pd_out, char* errbuf, size_t errbuf_len) {
    auto report = [&](const std::string& s) { if (errbuf && errbuf_len) { snprintf(errbuf, errbuf_len, "%s", s.c_str()); } };
    if (ndev <= 0 || !devs || nslices <= 0 || !prob || !Y || !X_out || !prob->V || !prob->frame_ptr || !prob->kidx || !prob->net ||
        !prob->weights) {
        report("qmri_recon_batch: invalid arguments");
        return QMRI_ERR_INVALID_ARG;
    }
    std::vector<std::thread> th;
    std::vector<int> status(ndev, QMRI_OK);
    std::vector<std::string> errs(ndev);
    for (int w = 0; w < ndev; ++w) {
        bool shared = false;
        for (int v = 0; v < ndev; ++v) shared = shared || (v != w && devs[v] == devs[w]);
        th.emplace_back([&, w, shared]() {
            status[w] = recon_worker(devs[w], shared, w, ndev, nslices, prob, (const char*)Y, (char*)X_out, qmap_out, pd_out, &errs[w]);
        });
    }
    for (auto& t : th) t.join();
    for (int w = 0; w < ndev; ++w)
        if (status[w] != QMRI_OK) { report("device " + std::to_string(devs[w]) + ": " + errs[w]); return status[w]; }
    return QMRI_OK;
}

// diagnostic: copy the per-workgroup stamps of the most recent conv launch (see conv_kernels.hip) to the host
extern "C" int qmri_debug_conv_stamps(qmri_ctx* ctx, unsigned long long* out, int nwg) {
    if (ctx && nwg == -6 && ctx->net.d_res_stamps) {                // (the resident-tile launch's stamps: 1024 values, knob res_stamps)
        QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
        QMRI_HIP(ctx, hipMemcpy(out, ctx->net.d_res_stamps, (size_t)1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return QMRI_OK;
    }
    if (!ctx || !ctx->net.d_stamps) return QMRI_ERR_STATE;
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QMRI_HIP(ctx, hipMemcpy(out, ctx->net.d_stamps, (size_t)4096 * 11 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return QMRI_OK;
}

// api_core.cpp -- context, error reporting, mask builders and the forward-operator plugin of libqmri.so.
//
// Replaces (reference file:line): struct F main_recon_tsmis_FFT.m:228-229; setup_subsampling_spiralgrided.m:1-43;
// setup_subsampling_epi.m:1-36; the lsqr x-update call site PnP_ADMM.m:102 with afun PnP_ADMM.m:153-171.
#include "qmri_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>

static thread_local std::string g_create_err;

// ---------------------------------------------------------------------------------------------------
// knobs (qmri_internal.h QmriKnob): A/B and diagnostic switches behind ONE environment variable and one entry point
// ---------------------------------------------------------------------------------------------------
namespace {
struct KnobDef { const char* name; int dflt; };
const KnobDef g_knob_defs[K_COUNT] = {
    {"conv_scheme", 2}, {"conv_f32", 0}, {"conv_wt", 1}, {"conv_xcd", 1}, {"conv_persist", 1}, {"conv_splitk", 1},
    {"conv_midcfg", 3}, {"conv_deepcfg", 3}, {"conv_deepks", 4},
    {"conv_resident", 1}, {"res_head", 1}, {"res_tail", 1}, {"res_down", 1}, {"res_delay", 24}, {"res_rearm", 64},
    {"res_stamps", 0}, {"conv_stamps", 0}, {"conv_stamp_launch", -1}, {"lsqr_stamps", 0},
    {"conv_mt2", 1024}, {"conv_occ", 2},
    {"fuse_ew", 1}, {"lsqr_persist", 1}, {"lsqr_fold", 1}, {"dictw_lsp", 2}, {"verbose", 0},
    {"pack_gpu", 1},
};
std::atomic<int> g_knob_val[K_COUNT];
std::once_flag g_knob_once;

int knob_index(const char* name, size_t len) {
    for (int k = 0; k < K_COUNT; ++k)
        if (strlen(g_knob_defs[k].name) == len && strncmp(g_knob_defs[k].name, name, len) == 0) return k;
    return -1;
}
void knob_init() {
    for (int k = 0; k < K_COUNT; ++k) g_knob_val[k].store(g_knob_defs[k].dflt, std::memory_order_relaxed);
    const char* e = getenv("QMRI_DEBUG");                          // the library's only environment variable
    while (e && *e) {
        const char* end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        const char* eq = (const char*)memchr(e, '=', len);
        const int k = eq ? knob_index(e, (size_t)(eq - e)) : -1;
        if (k >= 0) g_knob_val[k].store(atoi(eq + 1), std::memory_order_relaxed);
        else if (len) fprintf(stderr, "libqmri: QMRI_DEBUG: unknown or malformed entry '%.*s' ignored\n", (int)len, e);
        e = end ? end + 1 : nullptr;
    }
}
}  // namespace

int qmri_knob(QmriKnob k) {
    std::call_once(g_knob_once, knob_init);
    return g_knob_val[k].load(std::memory_order_relaxed);
}

extern "C" int qmri_debug_knob(const char* name, int value) {
    std::call_once(g_knob_once, knob_init);
    const int k = name ? knob_index(name, strlen(name)) : -1;
    if (k < 0) { qmri_set_error(nullptr, "qmri_debug_knob: unknown knob '%s'", name ? name : "(null)"); return QMRI_ERR_INVALID_ARG; }
    g_knob_val[k].store(value, std::memory_order_relaxed);
    return QMRI_OK;
}

void qmri_set_error(qmri_ctx* ctx, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
}

extern "C" int qmri_abi_version(void) { return QMRI_ABI_VERSION; }

extern "C" const char* qmri_last_error(const qmri_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int qmri_create(int device, qmri_ctx** out) {
    if (!out) { qmri_set_error(nullptr, "qmri_create: out is NULL"); return QMRI_ERR_INVALID_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        qmri_set_error(nullptr, "no HIP device available (%s); libqmri has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return QMRI_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        qmri_set_error(nullptr, "device %d out of range (0..%d)", device, ndev - 1);
        return QMRI_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        qmri_set_error(nullptr, "hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return QMRI_ERR_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        qmri_set_error(nullptr, "device %d is %s; libqmri is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return QMRI_ERR_UNSUPPORTED;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        qmri_set_error(nullptr, "hipSetDevice failed: %s", hipGetErrorString(e));
        return QMRI_ERR_HIP;
    }
    qmri_ctx* ctx = new (std::nothrow) qmri_ctx();
    if (!ctx) { qmri_set_error(nullptr, "out of host memory"); return QMRI_ERR_NOMEM; }
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        qmri_set_error(nullptr, "hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return QMRI_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    for (auto& ev : ctx->ev) {
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            qmri_set_error(nullptr, "hipEventCreate failed: %s", hipGetErrorString(e));
            for (auto& x : ctx->ev) if (x) (void)hipEventDestroy(x);
            (void)hipStreamDestroy(ctx->own_stream);
            delete ctx;
            return QMRI_ERR_HIP;
        }
    }
    *out = ctx;
    return QMRI_OK;
}

static void free_dev(void* p) { if (p) (void)hipFree(p); }

void qmri_free_operator(qmri_ctx* ctx) {
    OpHost& o = ctx->op;
    void* ptrs[] = { o.d_Vt, o.d_ent, o.d_perm, o.d_kptr, o.d_tw, o.d_kslot, o.d_ginv, o.d_tmp, o.d_xa, o.d_xb, o.d_ya,
                     o.ls.st, o.ls.pz, o.ls.yk, o.ls.py,
                     (void*)o.ks.unit, (void*)o.ks.es, (void*)o.ks.grp, (void*)o.ks.sgrp,
                     o.ks.pu[0], o.ks.pu[1], o.ks.pv[0], o.ks.pv[1], o.ks.pinit, o.ks.pR, o.ks.cx, o.ks.cv, o.ks.cd, o.ks.cub,
                     o.ks.ut, o.ks.xhat, o.ks.zhat, o.ks.xhat_out, o.ks.stamps,
                     o.d_x, o.d_u, o.d_vv, o.d_z, o.d_chat, o.d_mm, o.d_norm, o.d_diag, o.d_pd, o.d_coils };
    for (void* p : ptrs) free_dev(p);
    if (o.h_state) (void)hipHostFree(o.h_state);
    if (o.h_ring) (void)hipHostFree(o.h_ring);
    free_dev(ctx->d_ks_gran); ctx->d_ks_gran = nullptr; ctx->ks_persist_cap = -1;    // (sized for this operator's work units)
    o = OpHost();
}

void qmri_free_net(qmri_ctx* ctx);
void qmri_free_dict(qmri_ctx* ctx);

extern "C" int qmri_destroy(qmri_ctx* ctx) {
    if (!ctx) return QMRI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    qmri_free_operator(ctx);
    qmri_free_net(ctx);
    qmri_free_dict(ctx);
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_state) (void)hipEventDestroy(ctx->ev_state);
    for (hipEvent_t e : ctx->chain) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->marks) if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return QMRI_OK;
}

extern "C" int qmri_set_stream(qmri_ctx* ctx, void* hip_stream) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return QMRI_OK;
}

extern "C" int qmri_synchronize(qmri_ctx* ctx) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------
// mask builders (host, integer index math)
// ---------------------------------------------------------------------------------------------------
extern "C" int qmri_build_spiral(qmri_ctx* ctx, int N, int S, int T, int32_t* frame_ptr, int32_t* kidx, int cap, int* m_out) {
    QMRI_CHECK_ARG(ctx, N > 0 && S > 1 && T > 0 && frame_ptr && (kidx || cap == 0) && m_out, "qmri_build_spiral arguments");
    // setup_subsampling_spiralgrided.m:7-34.  8-turn exponential spiral sampled at S points, rotated 7.5 deg per
    // frame, rounded onto the N x N grid (MATLAB round = half away from zero), clamped, fftshift-ed; the sample
    // order inside a frame is find()'s ascending column-major order.
    const double pi = 3.14159265358979323846;
    const double delta = pi / 180.0 * 7.5;
    std::vector<double> theta(S), rad(S);
    double lo = HUGE_VAL, hi = -HUGE_VAL;
    for (int j = 0; j < S; ++j) {
        const double t = (j == S - 1) ? 2.0 * pi : (double)j * (2.0 * pi) / (double)(S - 1);   // linspace(0,2*pi,S)
        theta[j] = 8.0 * t;
        rad[j] = std::pow(1.05, theta[j]);
        lo = std::min(lo, rad[j]);
        hi = std::max(hi, rad[j]);
    }
    for (double& r : rad) r = (r - lo) / (hi - lo);
    std::vector<uint8_t> grid((size_t)N * N);
    const int half = N / 2;
    long m = 0;
    for (int f = 0; f < T; ++f) {
        frame_ptr[f] = (int32_t)m;
        std::fill(grid.begin(), grid.end(), 0);
        const double rot = (double)f * delta;
        for (int j = 0; j < S; ++j) {
            double gx = std::round(rad[j] * std::cos(theta[j] + rot) * N / 2.0) + N / 2.0 + 1.0;
            double gy = std::round(rad[j] * std::sin(theta[j] + rot) * N / 2.0) + N / 2.0 + 1.0;
            gx = std::min(gx, (double)N);
            gy = std::min(gy, (double)N);
            const int r = ((int)gx - 1 + half) % N, c = ((int)gy - 1 + half) % N;     // fftshift
            grid[(size_t)c * N + r] = 1;
        }
        for (int k = 0; k < N * N; ++k)
            if (grid[k]) {
                if (m < cap) kidx[m] = k;
                ++m;
            }
    }
    frame_ptr[T] = (int32_t)m;
    *m_out = (int)m;
    if (m > cap) { qmri_set_error(ctx, "kidx capacity %d too small, need %ld", cap, m); return QMRI_ERR_INVALID_ARG; }
    return QMRI_OK;
}

extern "C" int qmri_build_epi(qmri_ctx* ctx, int N, int M, double percentage, int T, int32_t* frame_ptr, int32_t* kidx,
                              int cap, int* m_out) {
    QMRI_CHECK_ARG(ctx, N > 0 && M > 0 && T > 0 && percentage > 0 && frame_ptr && (kidx || cap == 0) && m_out,
                   "qmri_build_epi arguments");
    // setup_subsampling_epi.m:20-33.  Comb of floor(N/step) k-rows, step = round(1/percentage), shifted down by one
    // row (cyclically) before every frame including the first; whole rows are sampled; no fftshift.
    const int step = (int)std::round(1.0 / percentage);
    const int nlines = N / step;
    std::vector<int> rows;
    for (int r = 0; r < step * nlines; r += step) rows.push_back(r);
    long m = 0;
    for (int f = 0; f < T; ++f) {
        frame_ptr[f] = (int32_t)m;
        for (int& r : rows) r = (r + 1) % N;
        std::vector<int> sorted(rows);
        std::sort(sorted.begin(), sorted.end());
        for (int c = 0; c < M; ++c)
            for (int r : sorted) {
                if (m < cap) kidx[m] = r + N * c;
                ++m;
            }
    }
    frame_ptr[T] = (int32_t)m;
    *m_out = (int)m;
    if (m > cap) { qmri_set_error(ctx, "kidx capacity %d too small, need %ld", cap, m); return QMRI_ERR_INVALID_ARG; }
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------
// operator
// ---------------------------------------------------------------------------------------------------
OpDev qmri_opdev(const qmri_ctx* ctx) {
    const OpHost& o = ctx->op;
    OpDev d;
    d.N = o.N; d.M = o.M; d.s = o.s; d.T = o.T; d.m = o.m;
    d.Vt = o.d_Vt; d.ent = o.d_ent; d.perm = o.d_perm; d.kptr = o.d_kptr; d.tw = o.d_tw;
    d.kslot = o.d_kslot; d.ginv = o.d_ginv;
    return d;
}

template <typename T> static int dev_alloc(qmri_ctx* ctx, T** p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { qmri_set_error(ctx, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e)); return QMRI_ERR_NOMEM; }
    return QMRI_OK;
}

extern "C" int qmri_set_operator(qmri_ctx* ctx, int N, int M, int s, int T, const double* V, const int32_t* frame_ptr,
                                 const int32_t* kidx, int max_batch) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    QMRI_HIP(ctx, hipSetDevice(ctx->device));
    QMRI_CHECK_ARG(ctx, V && frame_ptr && kidx, "V / frame_ptr / kidx must not be NULL");
    QMRI_CHECK_ARG(ctx, N > 0 && M > 0 && s > 0 && T > 0 && max_batch > 0, "N, M, s, T, max_batch must be positive");
    if (N != M || !dc_size_supported(N)) {
        qmri_set_error(ctx, "grid %d x %d unsupported: the FFT kernels implement square grids of 32, 64, 128, 224 "
                            "(the reference's spiral mask assumes N == M, setup_subsampling_spiralgrided.m:28-31)", N, M);
        return QMRI_ERR_UNSUPPORTED;
    }
    if (s > 10 || T > 65535 || M > 65535) { qmri_set_error(ctx, "s <= 10 and T, M <= 65535 required (got s=%d T=%d)", s, T); return QMRI_ERR_UNSUPPORTED; }
    QMRI_CHECK_ARG(ctx, frame_ptr[0] == 0, "frame_ptr[0] must be 0");
    const int m = frame_ptr[T];
    QMRI_CHECK_ARG(ctx, m > 0, "empty measurement set");
    for (int t = 0; t < T; ++t) QMRI_CHECK_ARG(ctx, frame_ptr[t + 1] >= frame_ptr[t], "frame_ptr must be non-decreasing");
    for (int i = 0; i < m; ++i) QMRI_CHECK_ARG(ctx, kidx[i] >= 0 && kidx[i] < N * M, "kidx out of range");

    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    qmri_free_operator(ctx);
    OpHost& o = ctx->op;
    o.N = N; o.M = M; o.s = s; o.T = T; o.m = m; o.maxB = max_batch;
    o.V.assign(V, V + (size_t)T * s);
    o.frame_ptr.assign(frame_ptr, frame_ptr + T + 1);
    o.kidx.assign(kidx, kidx + m);

    // k-sorted sample list: primary kh = k % N (k-space row), secondary kw = k / N, tertiary frame
    const int NM = N * M;
    o.kptr_h.assign(NM + 1, 0);
    std::vector<int32_t> frame_of(m);
    for (int t = 0; t < T; ++t)
        for (int i = frame_ptr[t]; i < frame_ptr[t + 1]; ++i) frame_of[i] = t;
    auto kprime = [&](int k) { return (k % N) * M + (k / N); };
    for (int i = 0; i < m; ++i) o.kptr_h[kprime(kidx[i]) + 1]++;
    for (int k = 0; k < NM; ++k) o.kptr_h[k + 1] += o.kptr_h[k];
    std::vector<int32_t> fill(o.kptr_h.begin(), o.kptr_h.end() - 1);
    o.ent_h.resize(m);
    o.perm_h.resize(m);
    for (int i = 0; i < m; ++i) {                    // ascending i == ascending frame inside each k
        const int kp = kprime(kidx[i]);
        const int e = fill[kp]++;
        o.ent_h[e].kw = (uint16_t)(kidx[i] / N);
        o.ent_h[e].t = (uint16_t)frame_of[i];
        o.perm_h[e] = i;
    }
    std::vector<int32_t> kslot(NM, -1);
    o.nsampled = 0;
    for (int kp = 0; kp < NM; ++kp)
        if (o.kptr_h[kp + 1] > o.kptr_h[kp]) kslot[kp] = o.nsampled++;
    std::vector<double> Vt((size_t)T * s);
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < s; ++c) Vt[(size_t)t * s + c] = V[t + (size_t)T * c];
    std::vector<double2> tw(N);
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < N; ++j) { const double a = 2.0 * pi * j / N; tw[j] = make_double2(std::cos(a), -std::sin(a)); }

    const size_t n = (size_t)N * M * s, B = (size_t)max_batch;
    QMRI_TRY(dev_alloc(ctx, &o.d_Vt, Vt.size()));
    QMRI_TRY(dev_alloc(ctx, &o.d_ent, (size_t)m));
    QMRI_TRY(dev_alloc(ctx, &o.d_perm, (size_t)m));
    QMRI_TRY(dev_alloc(ctx, &o.d_kptr, (size_t)NM + 1));
    QMRI_TRY(dev_alloc(ctx, &o.d_tw, (size_t)N));
    QMRI_TRY(dev_alloc(ctx, &o.d_kslot, (size_t)NM));
    QMRI_TRY(dev_alloc(ctx, &o.d_ginv, (size_t)o.nsampled * s * s));
    QMRI_HIP(ctx, hipMemcpy(o.d_Vt, Vt.data(), Vt.size() * sizeof(double), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(o.d_ent, o.ent_h.data(), (size_t)m * sizeof(KEntry), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(o.d_perm, o.perm_h.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(o.d_kptr, o.kptr_h.data(), ((size_t)NM + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(o.d_tw, tw.data(), (size_t)N * sizeof(double2), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipMemcpy(o.d_kslot, kslot.data(), (size_t)NM * sizeof(int32_t), hipMemcpyHostToDevice));

    LsqrDev& ls = o.ls;
    // ---- k-space LSQR plan (kslsqr_kernels.hip): the sampled k locations ("slots", k' order) are cut into work units of
    // similar sample count, each unit's samples into scatter groups of <= gcap samples of one slot (KS_CAPS: 32 for dense masks, 4 for sparse ones)
    KsDev& ks = o.ks;
    {
        const int ns = o.nsampled;
        std::vector<int32_t> sptr(ns + 1), sgrp(ns + 1), bslot, gptr;
        std::vector<KSample> es(m);
        std::vector<KsGroup> grp;
        {
            int j = 0;
            for (int kp = 0; kp < NM; ++kp) if (kslot[kp] >= 0) sptr[j++] = o.kptr_h[kp];
            sptr[ns] = m;
        }
        // Units of similar sample count, about one per CU.  The one-launch iteration (k_ks_persist) is paced by its slowest workgroup, and two
        // workgroups that share a CU are the slow ones: where the caps allow, the cut is repeated with fewer, larger units until at most
        // 250 result (263 at the headline operator with the first cut: 7 CUs held two).
        // Round 5: four unit shapes (KS_CAPS).  A mask that samples every k a few times (EPI: 784 units of the first shape) or few k very often
        // (cut0: 604) gets the shape under which <= 256 units result, so that a slice runs the one-launch iteration too -- a slice batch as
        // well, a slice or two per launch (ks_launch_persist).  With the one-launch iteration switched off, plans for slice batches keep the
        // small shapes: their grids (units x slices) never fit the chip at once, and small units fill it more evenly.
        auto cut = [&](const KsCapsHost& cp) -> int {
            int want = 256;
            for (int attempt = 0; attempt < 8; ++attempt, want -= 8) {
                bslot.clear(); gptr.clear(); grp.clear();
                const int target = std::max(1, (m + want - 1) / want);
                int j = 0;
                while (j < ns) {
                    const int first = j, e0 = sptr[j];
                    int ngr = 0;
                    bslot.push_back(first);
                    gptr.push_back((int32_t)grp.size());
                    while (j < ns) {
                        const int cnt = sptr[j + 1] - sptr[j], gj = (cnt + cp.gcap - 1) / cp.gcap;
                        if (cnt > cp.ecap || cnt > 65535) return -cnt;
                        const int have = sptr[j] - e0;
                        if (j > first && (j - first >= cp.scap || have + cnt > cp.ecap || ngr + gj > cp.gcapb || have + cnt / 2 > target)) break;
                        sgrp[j] = (int32_t)grp.size();
                        for (int e = sptr[j]; e < sptr[j + 1]; e += cp.gcap) {
                            KsGroup g;
                            g.ls = (uint16_t)(j - first); g.b = (uint16_t)(e - e0); g.e = (uint16_t)(std::min(e + cp.gcap, sptr[j + 1]) - e0); g.pad = 0;
                            grp.push_back(g);
                        }
                        for (int e = sptr[j]; e < sptr[j + 1]; ++e) { es[e].ls = (uint16_t)(j - first); es[e].t = o.ent_h[e].t; }
                        ngr += gj;
                        ++j;
                    }
                }
                if ((int)bslot.size() <= 250 || (int)bslot.size() > 320) break;     // (> 320: an operator with many more units than CUs; nothing to gain)
            }
            return (int)bslot.size();
        };
        ks.vcap = ((T * s + 10 + 15) / 16) * 16;       // (+10: the channel loops of the kernels are unrolled to 10)
        // sparse masks (fewer than 8 samples per sampled k on average: EPI) give every scatter group of <= 4 samples to one lane; dense ones
        // (the spiral: 11 at cut3, 56 at cut0) share groups of <= 32 samples among 8 lanes (KS_CAPS)
        const bool sparse = ns > 0 && (double)m / ns < 8.0;
        int base = sparse ? 3 : 0;
        ks.caps = base;
        int nunits = cut(KS_CAPS[base]);
        if (sparse && nunits < 0) { base = 0; ks.caps = 0; nunits = cut(KS_CAPS[0]); }      // (sparse on average with one very busy location: the dense shape)
        const bool one_launch = ctx->ks_persist != 0 && (ctx->ks_persist > 0 || qmri_knob(K_LSQR_PERSIST));
        if ((max_batch == 1 || one_launch) && s == 10 && (nunits > 320 || nunits < 0)) {
            int ncu = 0;                                   // (the larger shapes run one workgroup per CU: every unit needs a CU of its own)
            QMRI_HIP(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
            const int order[2] = {sparse ? 1 : 2, sparse ? 2 : 1};
            for (int c : order) {
                bool fits = false;
                QMRI_TRY(ks_lds_fits(ctx, N, s, M, ks.vcap, c, &fits));
                if (!fits) continue;
                const int nu = cut(KS_CAPS[c]);
                if (nu > 0 && nu <= ncu) { ks.caps = c; nunits = nu; break; }      // (cut0: 256 units of <= 2560 samples -- k = 0 alone is sampled in all 1000 frames)
            }
            if (ks.caps == base) nunits = cut(KS_CAPS[base]);
        }
        if (nunits < 0) {
            // the limit of the shapes that were actually TRIED (the larger shapes only with one slice per launch or the one-launch iteration)
            int tried = std::max(KS_CAPS[base].ecap, KS_CAPS[0].ecap);
            if ((max_batch == 1 || one_launch) && s == 10) tried = std::max(tried, std::max(KS_CAPS[1].ecap, KS_CAPS[2].ecap));
            qmri_set_error(ctx, "a k location is sampled %d times; at most %d are supported%s", -nunits, tried,
                           tried < KS_CAPS[2].ecap ? " with several slices per launch and the two-launch LSQR iteration (one slice per launch: 2560)" : "");
            return QMRI_ERR_UNSUPPORTED;
        }
        bslot.push_back(ns);
        gptr.push_back((int32_t)grp.size());
        sgrp[ns] = (int32_t)grp.size();
        ks.ns = ns; ks.G = (int)bslot.size() - 1; ks.b0 = 0;
        bool v_fits = false;
        QMRI_TRY(ks_lds_fits(ctx, N, s, M, ks.vcap, ks.caps, &v_fits));
        if (!v_fits) {     // T = 1000, s = 10 (cut0, main_recon_tsmis_FFT.m:41-44) needs 80 KB of the CU's 160 KB; T*s <~ 14 900 fits
            qmri_set_error(ctx, "V (T=%d x s=%d) does not fit the on-chip budget of the LSQR kernels", T, s);
            return QMRI_ERR_UNSUPPORTED;
        }
        std::vector<KsUnit> units(ks.G);
        for (int g = 0; g < ks.G; ++g) units[g] = KsUnit{bslot[g], bslot[g + 1], sptr[bslot[g]], sptr[bslot[g + 1]], gptr[g], gptr[g + 1], 0, 0};
        int32_t* p32; KSample* pes; KsGroup* pg; KsUnit* pu_;
#define KS_UPLOAD(ptr, field, vec, T_)                                                                 \
        QMRI_TRY(dev_alloc(ctx, &ptr, (vec).size()));                                                  \
        QMRI_HIP(ctx, hipMemcpy(ptr, (vec).data(), (vec).size() * sizeof(T_), hipMemcpyHostToDevice)); \
        ks.field = ptr;
        KS_UPLOAD(p32, sgrp, sgrp, int32_t)
        KS_UPLOAD(pes, es, es, KSample)
        KS_UPLOAD(pg, grp, grp, KsGroup)
        KS_UPLOAD(pu_, unit, units, KsUnit)
#undef KS_UPLOAD
        ctx->ks_lds_attr[0] = ctx->ks_lds_attr[1] = false;
    }
    ls.nblk_z = 256;
    QMRI_TRY(dev_alloc(ctx, &o.d_tmp, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_xa, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_xb, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_ya, B * (size_t)m));
    QMRI_TRY(dev_alloc(ctx, &ls.st, B));
    QMRI_TRY(dev_alloc(ctx, &ls.pz, B * ls.nblk_z));
    QMRI_TRY(dev_alloc(ctx, &ls.py, B * (size_t)DC_SORT_BLOCKS));
    for (int par = 0; par < 2; ++par) {
        QMRI_TRY(dev_alloc(ctx, &ks.pu[par], B * 2 * ks.G));
        QMRI_TRY(dev_alloc(ctx, &ks.pv[par], B * ks.G));
    }
    QMRI_TRY(dev_alloc(ctx, &ks.pinit, B * 2 * N));
    QMRI_TRY(dev_alloc(ctx, &ks.pR, B * N));
    QMRI_TRY(dev_alloc(ctx, &ks.cx, B * ks.ns * s));
    QMRI_TRY(dev_alloc(ctx, &ks.cv, B * ks.ns * s));
    QMRI_TRY(dev_alloc(ctx, &ks.cd, B * ks.ns * s));
    QMRI_TRY(dev_alloc(ctx, &ks.cub, B * ks.ns * s));
    QMRI_TRY(dev_alloc(ctx, &ks.ut, B * (size_t)m));
    QMRI_TRY(dev_alloc(ctx, &ks.xhat, B * n));
    QMRI_TRY(dev_alloc(ctx, &ks.zhat, B * n));
    QMRI_TRY(dev_alloc(ctx, &ks.xhat_out, B * n));
    ks.stamps = nullptr;
    if (qmri_knob(K_LSQR_STAMPS) > 0) { QMRI_TRY(dev_alloc(ctx, &ks.stamps, (size_t)2 * 512 * 16)); QMRI_HIP(ctx, hipMemset(ks.stamps, 0, 2 * 512 * 16 * 8)); }
    QMRI_TRY(dev_alloc(ctx, &ls.yk, B * (size_t)m));
    QMRI_TRY(dev_alloc(ctx, &o.d_x, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_u, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_vv, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_z, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_chat, B * n));
    QMRI_TRY(dev_alloc(ctx, &o.d_mm, B * ls.nblk_z * 2));
    QMRI_TRY(dev_alloc(ctx, &o.d_norm, B * 2));
    QMRI_TRY(dev_alloc(ctx, &o.d_pd, B * ((size_t)N + 2 * ls.nblk_z)));
    QMRI_HIP(ctx, hipMemset(ls.st, 0, B * sizeof(LsqrState)));
    QMRI_HIP(ctx, hipHostMalloc((void**)&o.h_state, B * sizeof(LsqrState), hipHostMallocDefault));
    ks.hst = nullptr; ks.st = ls.st; ks.pz = ls.pz; ks.nblk_z = ls.nblk_z; ks.yk = ls.yk; ks.pdiag = nullptr;
    o.xhat_valid = false;
    o.ginv_r = -1.0;
    // Round 6: every table above went to the device by blocking copies on the NULL stream, and the kernels that read them run on this context's
    // NON-BLOCKING stream, which is not ordered with it.  hipMemcpy returns once the host buffer has been consumed; beside another process on the
    // device the transfer itself was seen to land AFTER kernels of a later call had started (tools/probe_under_contention.py: a set-up probe reading
    // weights that had not arrived).  One device-wide synchronisation per plan closes that for every later launch.
    QMRI_HIP(ctx, hipDeviceSynchronize());
    o.ready = true;
    return QMRI_OK;
}

extern "C" int qmri_operator_m(const qmri_ctx* ctx, int* m_out) {
    if (!ctx || !m_out) return QMRI_ERR_INVALID_ARG;
    if (!ctx->op.ready) return QMRI_ERR_STATE;
    *m_out = ctx->op.m;
    return QMRI_OK;
}

#define REQUIRE_OP(ctx)                                                                   \
    do {                                                                                  \
        if (!(ctx)) return QMRI_ERR_INVALID_ARG;                                          \
        QMRI_HIP((ctx), hipSetDevice((ctx)->device));                                     \
        if (!(ctx)->op.ready) { qmri_set_error((ctx), "operator not set: call qmri_set_operator first"); return QMRI_ERR_STATE; } \
    } while (0)

extern "C" int qmri_forward_dev(qmri_ctx* ctx, const void* d_x, void* d_y, int batch) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, d_x && d_y && batch >= 1 && batch <= ctx->op.maxB, "qmri_forward_dev arguments / batch > max_batch");
    return dc_launch_fwd(ctx, qmri_opdev(ctx), ctx->op.ls, DC_PLAIN, batch, (const double2*)d_x, ctx->op.d_tmp,
                         (double2*)d_y, nullptr);
}

extern "C" int qmri_adjoint_dev(qmri_ctx* ctx, const void* d_y, void* d_x, int batch) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, d_x && d_y && batch >= 1 && batch <= ctx->op.maxB, "qmri_adjoint_dev arguments / batch > max_batch");
    return dc_launch_adj(ctx, qmri_opdev(ctx), batch, (const double2*)d_y, ctx->op.d_tmp, (double2*)d_x);
}

extern "C" int qmri_forward(qmri_ctx* ctx, const void* x, int x_is_complex, void* y) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    OpHost& o = ctx->op;
    const size_t n = (size_t)o.N * o.M * o.s;
    if (x_is_complex) {
        QMRI_HIP(ctx, hipMemcpyAsync(o.d_xa, x, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    } else {
        QMRI_HIP(ctx, hipMemcpyAsync(o.d_xb, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        QMRI_TRY(ew_launch_real_to_complex(ctx, n, (const double*)o.d_xb, o.d_xa));
    }
    QMRI_TRY(qmri_forward_dev(ctx, o.d_xa, o.d_ya, 1));
    QMRI_HIP(ctx, hipMemcpyAsync(y, o.d_ya, (size_t)o.m * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

extern "C" int qmri_adjoint(qmri_ctx* ctx, const void* y, void* x) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    OpHost& o = ctx->op;
    const size_t n = (size_t)o.N * o.M * o.s;
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_ya, y, (size_t)o.m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    QMRI_TRY(qmri_adjoint_dev(ctx, o.d_ya, o.d_xa, 1));
    QMRI_HIP(ctx, hipMemcpyAsync(x, o.d_xa, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------
// Multi-coil extension (BASELINE.json configs[4]: "complex-valued multi-coil forward op").  NO counterpart in the reference -- it simulates a
// single coil (README.md:63) -- so there is nothing to be a drop-in for and nothing that pins it: parity unpinned, checked by adjointness,
// closed forms and a restatement in the oracle (oracle.py Operator.forward_mc / adjoint_mc).  A_mc x = [A (C_1 .* x); ...; A (C_nc .* x)] with A the
// single-coil operator of qmri_set_operator and C_j the sensitivity maps; A_mc^H y = sum_j conj(C_j) .* A^H y_j.  The coils of a call go through
// the batched transforms max_batch at a time.
// ---------------------------------------------------------------------------------------------------
extern "C" int qmri_set_coils(qmri_ctx* ctx, int ncoil, const void* maps) {
    REQUIRE_OP(ctx);
    OpHost& o = ctx->op;
    QMRI_CHECK_ARG(ctx, ncoil >= 0 && ncoil <= 1024 && (ncoil == 0 || maps), "0 <= ncoil <= 1024, maps must not be NULL");
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (o.d_coils) { (void)hipFree(o.d_coils); o.d_coils = nullptr; }
    o.ncoil = 0;
    if (ncoil == 0) return QMRI_OK;
    const size_t count = (size_t)ncoil * o.N * o.M;
    QMRI_HIP(ctx, hipMalloc((void**)&o.d_coils, count * sizeof(double2)));
    QMRI_HIP(ctx, hipMemcpy(o.d_coils, maps, count * sizeof(double2), hipMemcpyHostToDevice));
    QMRI_HIP(ctx, hipDeviceSynchronize());                        // (as in qmri_set_operator: the copy must have landed before this context's stream reads it)
    o.ncoil = ncoil;
    return QMRI_OK;
}

extern "C" int qmri_forward_mc(qmri_ctx* ctx, const void* x, int x_is_complex, void* y) {
    REQUIRE_OP(ctx);
    OpHost& o = ctx->op;
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    if (!o.ncoil) { qmri_set_error(ctx, "no coil maps set: call qmri_set_coils first"); return QMRI_ERR_STATE; }
    const size_t n = (size_t)o.N * o.M * o.s, plane = (size_t)o.N * o.M;
    if (x_is_complex) {
        QMRI_HIP(ctx, hipMemcpyAsync(o.d_xa, x, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    } else {
        QMRI_HIP(ctx, hipMemcpyAsync(o.d_xb, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        QMRI_TRY(ew_launch_real_to_complex(ctx, n, (const double*)o.d_xb, o.d_xa));
    }
    for (int j0 = 0; j0 < o.ncoil; j0 += o.maxB) {
        const int cnt = std::min(o.maxB, o.ncoil - j0);
        QMRI_TRY(ew_launch_coil_mul(ctx, n, plane, cnt, o.d_xa, o.d_coils + (size_t)j0 * plane, o.d_x));          // (o.d_x: [max_batch][n], free outside a reconstruction)
        QMRI_TRY(dc_launch_fwd(ctx, qmri_opdev(ctx), o.ls, DC_PLAIN, cnt, o.d_x, o.d_tmp, o.d_ya, nullptr));
        QMRI_HIP(ctx, hipMemcpyAsync((double2*)y + (size_t)j0 * o.m, o.d_ya, (size_t)cnt * o.m * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    }
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

extern "C" int qmri_adjoint_mc(qmri_ctx* ctx, const void* y, void* x) {
    REQUIRE_OP(ctx);
    OpHost& o = ctx->op;
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    if (!o.ncoil) { qmri_set_error(ctx, "no coil maps set: call qmri_set_coils first"); return QMRI_ERR_STATE; }
    const size_t n = (size_t)o.N * o.M * o.s, plane = (size_t)o.N * o.M;
    for (int j0 = 0; j0 < o.ncoil; j0 += o.maxB) {
        const int cnt = std::min(o.maxB, o.ncoil - j0);
        QMRI_HIP(ctx, hipMemcpyAsync(o.d_ya, (const double2*)y + (size_t)j0 * o.m, (size_t)cnt * o.m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        QMRI_TRY(dc_launch_adj(ctx, qmri_opdev(ctx), cnt, o.d_ya, o.d_tmp, o.d_x));
        QMRI_TRY(ew_launch_coil_sum(ctx, n, plane, cnt, o.d_x, o.d_coils + (size_t)j0 * plane, o.d_xa, j0 > 0));
    }
    QMRI_HIP(ctx, hipMemcpyAsync(x, o.d_xa, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return QMRI_OK;
}

// single-precision boundary (MATLAB `single` arrays): widened on the host, computed in double, rounded once on the way out
extern "C" int qmri_forward_f32(qmri_ctx* ctx, const float* x, int x_is_complex, float* y) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    const OpHost& o = ctx->op;
    const size_t n = (size_t)o.N * o.M * o.s * (x_is_complex ? 2 : 1);
    std::vector<double> xd(x, x + n), yd((size_t)2 * o.m);
    QMRI_TRY(qmri_forward(ctx, xd.data(), x_is_complex, yd.data()));
    for (size_t i = 0; i < yd.size(); ++i) y[i] = (float)yd[i];
    return QMRI_OK;
}

extern "C" int qmri_adjoint_f32(qmri_ctx* ctx, const float* y, float* x) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, x && y, "x / y must not be NULL");
    const OpHost& o = ctx->op;
    const size_t n2 = (size_t)2 * o.N * o.M * o.s;
    std::vector<double> yd(y, y + (size_t)2 * o.m), xd(n2);
    QMRI_TRY(qmri_adjoint(ctx, yd.data(), xd.data()));
    for (size_t i = 0; i < n2; ++i) x[i] = (float)xd[i];
    return QMRI_OK;
}

// ---------------------------------------------------------------------------------------------------
// x-update drivers
// ---------------------------------------------------------------------------------------------------
// (G_k + r I)^-1 for every sampled k, G_k = sum_{t: k in Omega_t} V(t,:)' V(t,:)   
int qmri_prepare_direct(qmri_ctx* ctx, double r) {
    OpHost& o = ctx->op;
    if (o.ginv_r == r) return QMRI_OK;
    const int s = o.s, NM = o.N * o.M, T = o.T;
    std::vector<double> ginv((size_t)o.nsampled * s * s);
    std::vector<double> G(s * s), Li(s * s);
    int slot = 0;
    for (int kp = 0; kp < NM; ++kp) {
        const int e0 = o.kptr_h[kp], e1 = o.kptr_h[kp + 1];
        if (e1 == e0) continue;
        std::fill(G.begin(), G.end(), 0.0);
        for (int e = e0; e < e1; ++e) {
            const int t = o.ent_h[e].t;
            for (int a = 0; a < s; ++a)
                for (int b = 0; b < s; ++b) G[a * s + b] += o.V[t + (size_t)T * a] * o.V[t + (size_t)T * b];
        }
        for (int a = 0; a < s; ++a) G[a * s + a] += r;
        // Cholesky G = L L^T, then inverse = L^-T L^-1
        for (int j = 0; j < s; ++j) {
            double d = G[j * s + j];
            for (int k = 0; k < j; ++k) d -= G[j * s + k] * G[j * s + k];
            const double ljj = std::sqrt(d);
            G[j * s + j] = ljj;
            for (int i = j + 1; i < s; ++i) {
                double v = G[i * s + j];
                for (int k = 0; k < j; ++k) v -= G[i * s + k] * G[j * s + k];
                G[i * s + j] = v / ljj;
            }
        }
        std::fill(Li.begin(), Li.end(), 0.0);
        for (int c = 0; c < s; ++c) {                 // Li = L^-1 (lower), column by column
            for (int i = c; i < s; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) v -= G[i * s + k] * Li[k * s + c];
                Li[i * s + c] = v / G[i * s + i];
            }
        }
        double* out = ginv.data() + (size_t)slot * s * s;
        for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) {
                double v = 0.0;
                for (int k = std::max(a, b); k < s; ++k) v += Li[k * s + a] * Li[k * s + b];
                out[a * s + b] = v;
            }
        ++slot;
    }
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_ginv, ginv.data(), ginv.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    o.ginv_r = r;
    return QMRI_OK;
}

// LSQR on B slices, iterated in k-space (kslsqr_kernels.hip).  Requires ls.yk / ny2 (dc_launch_sort_y) and ls.pz
// (dc_launch_prepare_z) to be current.  d_x holds x0 on entry and the solution on return; when o.xhat_valid the spectrum of
// x0 is taken from the previous solve instead of being recomputed.  pdiag (or null) receives [B][N] partials of ||y - A x||^2.
// hslot / deferred (round 3): a caller that does not need the iteration count at once passes B pinned LsqrState entries of its own and a flag;
// when the one-launch kernel runs, the call returns WITHOUT waiting (*deferred = true, iters_out / flag_out untouched): the kernel leaves
// iter / done / flag in hslot, to be read after the caller's next synchronisation (state 77 there = the kernel timed out: the caller repeats
// its work with ctx->ks_persist = 0).  The ADMM loop uses this so that the host never waits inside a reconstruction.
// fuse (round 4, nullable): what the ADMM loop has already done in neighbouring launches, or wants done in this solve's last one --
// z_hpass_nblk > 0: o.d_tmp holds the h-pass of z's transform and ls.pz that many partial sums of |z|^2 per slice (k_dual_fwd_h), d_z is not read;
// mm_u / mm: the final h-pass also leaves the partial min / max of real(x + mm_u) per workgroup in mm (k_adj_h).
int qmri_lsqr_run(qmri_ctx* ctx, int B, const double2* d_z, double r, double tol, int maxit, double2* d_x,
                  int32_t* iters_out, int32_t* flag_out, double* pdiag, LsqrState* hslot, bool* deferred, const LsqrFuse* fuse) {
    OpHost& o = ctx->op;
    const OpDev op = qmri_opdev(ctx);
    KsDev ks = o.ks;
    ks.sr = std::sqrt(r); ks.tol = tol; ks.maxit = maxit; ks.ii = 0; ks.pdiag = pdiag;
    LsqrState* const hs = hslot ? hslot : o.h_state;
    ks.hst = hs;
    if (deferred) *deferred = false;
    const bool z_fused = fuse && fuse->z_hpass_nblk > 0;
    const double2* const mm_u = fuse ? fuse->mm_u : nullptr;
    double* const mm = fuse ? fuse->mm : nullptr;
    if (z_fused && !o.xhat_valid) { qmri_set_error(ctx, "qmri_lsqr_run: a transformed z needs the spectrum of x (internal)"); return QMRI_ERR_STATE; }
    if (!o.xhat_valid) QMRI_TRY(dc_launch_fwd(ctx, op, o.ls, DC_SPECTRUM, B, d_x, o.d_tmp, ks.xhat, nullptr));
    if (z_fused) ks.nblk_z = fuse->z_hpass_nblk;
    else QMRI_TRY(dc_launch_fwd(ctx, op, o.ls, DC_FWD_H_ONLY, B, d_z, o.d_tmp, nullptr, nullptr));
    // Round 5: when the one-launch kernel will run, it takes the first Golub-Kahan step (beta0, v = B'u / beta0, d = 0) itself -- k_ks_b<INIT>
    // is not launched, v and d never exist in memory (knob lsqr_fold = 0: the separate launch, same bits)
    if (ctx->ks_persist < 0) ctx->ks_persist = qmri_knob(K_LSQR_PERSIST) ? 1 : 0;
    int per_launch = 0;
    if (ctx->ks_persist > 0 && maxit >= 1) QMRI_TRY(ks_persist_plan(ctx, op, ks, B, &per_launch));
    const bool fold = per_launch > 0 && qmri_knob(K_LSQR_FOLD) != 0;
    QMRI_TRY(ks_launch_init(ctx, op, ks, B, o.d_tmp, !fold));           // w-pass of z (-> ks.zhat) + first Golub-Kahan vectors
    // The predicted number of iterations is launched, then -- speculatively -- the kernels that turn the solution back into
    // an image.  The host waits only for the copy of the LSQR state (an event between the two), so the device keeps working
    // while the host wakes up and enqueues the next stage.  If a slice was not done yet (rare: counts fall from one x-update
    // to the next), two more iterations at a time follow and the final kernels run again from the untouched inputs.
    if (!ctx->ev_state) QMRI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_state, hipEventDisableTiming));
    // Round 3: all iterations in ONE launch when the grid is resident at once (k_ks_persist: the iteration's state stays on chip, the
    // two sums per iteration are in-kernel hand-offs instead of kernel boundaries); the two-launch iteration otherwise (EPI masks,
    // cut0, slice batches) and after a time-out of the persistent kernel (never seen; the inputs are untouched then).
    bool persisted = false;
    if (per_launch > 0) {
        if (!ctx->d_ks_gran) {
            const size_t nb = ks_gran_bytes(ks.G, o.maxB);
            QMRI_HIP(ctx, hipMalloc(&ctx->d_ks_gran, nb));
            QMRI_HIP(ctx, hipMemsetAsync(ctx->d_ks_gran, 0, nb, ctx->stream));      // (tag 0 is never used)
        }
        const unsigned tag0 = ctx->ks_tag;
        QMRI_TRY(ks_launch_persist(ctx, op, ks, B, ctx->d_ks_gran, tag0, fold, &persisted));
        if (persisted) {
            ctx->ks_tag += 2u * (unsigned)(maxit + 2);
            QMRI_TRY(ks_launch_final(ctx, op, ks, B, o.d_tmp));
            QMRI_TRY(dc_launch_adj_h(ctx, op, B, o.d_tmp, d_x, mm_u, mm));
            if (deferred) {                                        // the caller reads hslot after its own synchronisation
                *deferred = true;
                std::swap(o.ks.xhat, o.ks.xhat_out);
                o.xhat_valid = true;
                return QMRI_OK;
            }
            // Not deferred (qmri_xupdate): wait for the stream -- the kernels above are a few tens of microseconds, any HIP error ends the wait,
            // and k_ks_final_w has then overridden the state if a workgroup of the one-launch kernel gave up.  (The ADMM loop never waits here:
            // it reads its pinned slots after its own final synchronisation.)
            QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int b = 0; b < B; ++b)
                if ((unsigned)__atomic_load_n(&hs[b].pad, __ATOMIC_ACQUIRE) != tag0) {
                    qmri_set_error(ctx, "the one-launch LSQR kernel ended without reporting its state");
                    return QMRI_ERR_HIP;
                }
            bool timed_out = false;
            for (int b = 0; b < B; ++b) timed_out = timed_out || hs[b].flag == 77;
            if (timed_out) {                                       // repeat with the two-launch iteration, from the untouched inputs
                fprintf(stderr, "libqmri: the one-launch LSQR timed out waiting for a partial sum; using the two-launch iteration from now on\n");
                ctx->ks_persist = 0;
                ctx->ks_timeouts += 1;
                persisted = false;
                QMRI_TRY(ks_launch_init(ctx, op, ks, B));
            }
        }
    }
    // (guard: the first Golub-Kahan step was left to a one-launch kernel that then did not run -- the plan is the same in both places, so this
    //  cannot happen; if it ever does, the two-launch iteration must not start from an unset state)
    if (fold && !persisted && !(ctx->ks_persist == 0)) QMRI_TRY(ks_launch_init(ctx, op, ks, B));
    int launched = 0;
    int chunk = std::min(std::max(ctx->lsqr_pred, 1), std::max(maxit, 1));
    bool all_done = persisted;
    // (do ... while: with maxit == 0 the final kernels still run once and return x0 -- the assembled spectrum and the diagnostics' partial
    //  sums must exist whatever the iteration count)
    if (!persisted) do {
        const int nthis = std::min(chunk, maxit - launched);
        for (int k = 0; k < nthis; ++k) {
            ks.ii = launched + k + 1;
            QMRI_TRY(ks_launch_iter(ctx, op, ks, B));
        }
        launched += std::max(nthis, 0);
        // (no copy of the state: k_ks_b writes iter / done / flag of every slice to the pinned host array itself, ks.hst)
        QMRI_HIP(ctx, hipEventRecord(ctx->ev_state, ctx->stream));
        QMRI_TRY(ks_launch_final(ctx, op, ks, B, o.d_tmp));              // reads ks.xhat (x0), writes ks.xhat_out
        QMRI_TRY(dc_launch_adj_h(ctx, op, B, o.d_tmp, d_x, mm_u, mm));
        QMRI_HIP(ctx, hipEventSynchronize(ctx->ev_state));
        all_done = true;
        for (int b = 0; b < B; ++b) all_done = all_done && hs[b].done;
        chunk = 2;
    } while (launched < maxit && !all_done);
    std::swap(o.ks.xhat, o.ks.xhat_out);                                  // the assembled spectrum is the next solve's xhat0
    o.xhat_valid = true;
    int worst = 0;
    for (int b = 0; b < B; ++b) {
        const int it = hs[b].done ? hs[b].iter : maxit;
        if (iters_out) iters_out[b] = it;
        if (flag_out) flag_out[b] = hs[b].done ? hs[b].flag : 1;
        worst = std::max(worst, it);
        ctx->prof.lsqr_iters += it;
    }
    ctx->lsqr_pred = worst + 1;      // convergence is detected one iteration after the last x update; counts fall from one x-update to the next
    return QMRI_OK;
}

extern "C" int qmri_xupdate(qmri_ctx* ctx, const void* y, const void* z, double r, double tol, int maxit, int solver,
                            void* x, int32_t* iters_out, int32_t* flag_out) {
    REQUIRE_OP(ctx);
    QMRI_CHECK_ARG(ctx, y && z && x && r > 0 && maxit >= 0, "qmri_xupdate arguments");
    OpHost& o = ctx->op;
    const OpDev op = qmri_opdev(ctx);
    const size_t n = (size_t)o.N * o.M * o.s;
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_ya, y, (size_t)o.m * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_z, z, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    QMRI_HIP(ctx, hipMemcpyAsync(o.d_x, x, n * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    if (solver == QMRI_SOLVER_LSQR) {
        QMRI_TRY(dc_launch_sort_y(ctx, op, o.ls, 1, o.d_ya));
        // ||z||^2 partials: reuse prepare_z with u = 0
        QMRI_HIP(ctx, hipMemsetAsync(o.d_u, 0, n * sizeof(double2), ctx->stream));
        QMRI_TRY(dc_launch_prepare_z(ctx, op, o.ls, 1, o.d_z, o.d_u, o.d_vv));
        o.xhat_valid = false;
        QMRI_TRY(qmri_lsqr_run(ctx, 1, o.d_vv, r, tol, maxit, o.d_x, iters_out, flag_out, nullptr, nullptr, nullptr, nullptr));
    } else if (solver == QMRI_SOLVER_DIRECT) {
        QMRI_TRY(qmri_prepare_direct(ctx, r));
        QMRI_TRY(dc_launch_adj(ctx, op, 1, o.d_ya, o.d_tmp, o.d_xa));
        QMRI_TRY(dc_launch_fwd(ctx, op, o.ls, DC_SPECTRUM, 1, o.d_xa, o.d_tmp, o.d_chat, nullptr));
        QMRI_TRY(dc_launch_direct(ctx, qmri_opdev(ctx), 1, o.d_z, o.d_chat, r, o.d_tmp, o.d_x));
        if (iters_out) *iters_out = 0;
        if (flag_out) *flag_out = 0;
    } else {
        qmri_set_error(ctx, "unknown solver %d", solver);
        return QMRI_ERR_INVALID_ARG;
    }
    QMRI_HIP(ctx, hipMemcpyAsync(x, o.d_x, n * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return qmri_prof_chain_finish(ctx);
}

// test / A-B hook: 1 = all LSQR iterations in one launch where the grid is resident (default), 0 = the two-launch iteration
extern "C" int qmri_debug_lsqr_persist(qmri_ctx* ctx, int on) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    ctx->ks_persist = (on == 2) ? 2 : (on ? 1 : 0);               // (2: test hook -- one partial sum is withheld, the time-out path must take over)
    if (ctx->d_ks_gran && ctx->op.ready) {                         // a fresh start: forget an earlier time-out (the sticky word behind the granules)
        QMRI_HIP(ctx, hipSetDevice(ctx->device));
        QMRI_HIP(ctx, hipMemsetAsync((char*)ctx->d_ks_gran + ks_gran_bytes(ctx->op.ks.G, ctx->op.maxB) - 64, 0, 64, ctx->stream));
    }
    return QMRI_OK;
}

// diagnostic: phase stamps of the most recent LSQR launches (see lsqr_kernels.hip); out holds 2*512*16 values
extern "C" int qmri_debug_lsqr_stamps(qmri_ctx* ctx, unsigned long long* out) {
    if (!ctx || !ctx->op.ks.stamps) return QMRI_ERR_STATE;
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    QMRI_HIP(ctx, hipMemcpy(out, ctx->op.ks.stamps, (size_t)2 * 512 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return QMRI_OK;
}

// profile level 2: the next (start, stop) event pair of the current chain.  The events take the timestamps of a kernel's own dispatch packet
// (hipExtLaunchKernelGGL) -- the duration rocprofv3 --kernel-trace reports -- or, with start on one launch and stop on a later one, the span
// from the first kernel's start to the last one's end (a split-K layer: convolution + reduce).  `kind` says which accumulator of qmri_profile
// the duration goes to, `flop` is the pair's fp32-equivalent algorithmic work (convolutions).
int qmri_prof_pair(qmri_ctx* ctx, hipEvent_t* start, hipEvent_t* stop, int kind, double flop) {
    *start = *stop = nullptr;
    if (ctx->prof_level < 2) return QMRI_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return QMRI_OK;
    while (ctx->chain.size() < ctx->chain_n + 2) {
        hipEvent_t e = nullptr;
        QMRI_HIP(ctx, hipEventCreate(&e));
        ctx->chain.push_back(e);
    }
    *start = ctx->chain[ctx->chain_n];
    *stop = ctx->chain[ctx->chain_n + 1];
    const size_t i = ctx->chain_n / 2;
    if (ctx->chain_kind.size() < i + 1) { ctx->chain_kind.resize(i + 1, PROF_CONV3); ctx->chain_flop.resize(i + 1, 0.0); }
    ctx->chain_kind[i] = kind;
    ctx->chain_flop[i] = flop;
    ctx->chain_n += 2;
    return QMRI_OK;
}

// synchronises and adds the chain's durations to the profile; count >= 0: only the first `count` pairs (the others are dropped)
int qmri_prof_chain_finish(qmri_ctx* ctx, long count) {
    if (ctx->prof_level < 2 || ctx->chain_n == 0) return QMRI_OK;
    QMRI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < ctx->chain_n; i += 2) {
        if (count >= 0 && (long)(i / 2) >= count) break;
        float ms = 0.f;
        QMRI_HIP(ctx, hipEventElapsedTime(&ms, ctx->chain[i], ctx->chain[i + 1]));
        const double flop = ctx->chain_flop[i / 2];
        switch (ctx->chain_kind[i / 2]) {
            case PROF_TV: ctx->prof.ms_tv_iter += ms; ctx->prof.n_tv_iter += 1; break;
            case PROF_LSQR: ctx->prof.ms_lsqr_kernels += ms; ctx->prof.n_lsqr_launches += 1; break;
            case PROF_CONV2: ctx->prof.ms_conv2x2 += ms; ctx->prof.n_conv2x2 += 1; ctx->prof.flop_conv2x2 += flop; break;
            default: ctx->prof.ms_conv3x3 += ms; ctx->prof.n_conv3x3 += 1; ctx->prof.flop_conv3x3 += flop; break;
        }
    }
    ctx->chain_n = 0;
    return QMRI_OK;
}

extern "C" int qmri_profile_enable(qmri_ctx* ctx, int level) {
    if (!ctx) return QMRI_ERR_INVALID_ARG;
    ctx->prof_level = level;
    return QMRI_OK;
}

extern "C" int qmri_profile_get(qmri_ctx* ctx, qmri_profile* out, int reset) {
    if (!ctx || !out) return QMRI_ERR_INVALID_ARG;
    *out = ctx->prof;
    if (reset) ctx->prof = qmri_profile();
    return QMRI_OK;
}

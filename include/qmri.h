/*
 * qmri.h -- C ABI of libqmri.so: the MI355X (gfx950) PnP-ADMM MR-Fingerprinting reconstruction engine.
 *
 * This is the drop-in boundary for the hot path of ketanfatania/QMRI-PnP-Recon-POC.  The reference has no
 * FFI of its own (pure MATLAB); its plugin surface for this path is a set of MATLAB values, and every entry
 * point below states which of them it replaces (file:line relative to the reference root):
 *
 *   struct F with F.forward / F.adjoint          main_recon_tsmis_FFT.m:228-229   -> qmri_set_operator, qmri_forward, qmri_adjoint
 *   P = setup_subsampling_spiralgrided(N,M,S,V)   setup_subsampling_spiralgrided.m:1-43 -> qmri_build_spiral
 *   P = setup_subsampling_epi(N,M,pct,V)          setup_subsampling_epi.m:1-36     -> qmri_build_epi
 *   param.net = @(x) denoiseImage_PnP_ADMM(...)   main_recon_tsmis_FFT.m:164, denoiseImage_PnP_ADMM.m:1-117 -> qmri_set_denoiser, qmri_denoise
 *   Net = importONNXNetwork(denoiser_path, ...)   main_recon_tsmis_FFT.m:138 (weights only)      -> qmri_onnx_read_unetres
 *   x = PnP_ADMM(y, param)                        PnP_ADMM.m:1                      -> qmri_pnp_admm
 *   out = mrf_dtm_cpu(dict, data, par)            mrf_dtm_cpu.m:1                   -> qmri_set_dictionary, qmri_dict_match
 *   x = FISTA_deep(data, param) (LRTV option)     FISTA_deep.m:1, main_recon_tsmis_FFT.m:273-282 -> qmri_lrtv, qmri_prox_tv, qmri_norm_tv
 *
 * Conventions (frozen):
 *   - every function returns 0 on success or a negative qmri_status; the message is qmri_last_error(ctx).
 *     Nothing throws across the boundary.
 *   - arrays are column-major; complex numbers are interleaved (re,im) doubles -- MATLAB R2018a+
 *     mxComplexDouble layout.  A MATLAB array X(h,w,c) is the C array [c][w][h].
 *   - k-space indices crossing the ABI are 0-based column-major k = row + N*col; the measurement vector is
 *     ordered frame-major, ascending k inside a frame (the row order of P in setup_subsampling_*.m:34-37).
 *   - dm (dictionary index) is 1-based, as mrf_dtm_cpu.m:92 returns it.
 *   - host-pointer entry points copy in/out around the call; *_dev entry points take device pointers that are
 *     already resident in HBM (used by the benchmark and by callers that chain stages on the GPU).
 *   - a context is owned by one host thread at a time (MATLAB calls from one thread); it is not locked.
 *   - the library fails loudly (QMRI_ERR_HIP) when no gfx950 device is usable; there is no CPU fallback.
 */
#ifndef QMRI_H
#define QMRI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMRI_ABI_VERSION 1

typedef enum {
    QMRI_OK = 0,
    QMRI_ERR_INVALID_ARG = -1,   /* bad pointer / size / enum (MATLAB: validateattributes errors) */
    QMRI_ERR_STATE = -2,         /* operator / denoiser / dictionary not set yet */
    QMRI_ERR_HIP = -3,           /* HIP runtime error, or no usable GPU */
    QMRI_ERR_UNSUPPORTED = -4,   /* size or architecture outside what the kernels implement */
    QMRI_ERR_NOMEM = -5
} qmri_status;

typedef struct qmri_ctx qmri_ctx;

/* ---- context -------------------------------------------------------------------------------------- */
int qmri_abi_version(void);
/* One context = one device, one HIP stream, all device buffers and workspaces. */
int qmri_create(int device, qmri_ctx** out);
int qmri_destroy(qmri_ctx* ctx);
/* Message of the last failing call on this context (ctx == NULL: last failing qmri_create on this thread). */
const char* qmri_last_error(const qmri_ctx* ctx);
/* Launch on the caller's hipStream_t instead of the context's own stream (NULL restores it). */
int qmri_set_stream(qmri_ctx* ctx, void* hip_stream);
int qmri_synchronize(qmri_ctx* ctx);

/* ---- forward-operator plugin: struct F, main_recon_tsmis_FFT.m:228-229 ------------------------------ */
/* Mask builders (host, integer): replace setup_subsampling_spiralgrided.m:7-34 / setup_subsampling_epi.m:20-33.
 * frame_ptr has T+1 entries, kidx holds up to cap entries; *m_out receives the total sample count.
 * Returns QMRI_ERR_INVALID_ARG with *m_out set if cap is too small.  ctx may be NULL. */
int qmri_build_spiral(qmri_ctx* ctx, int N, int S, int T, int32_t* frame_ptr, int32_t* kidx, int cap, int* m_out);
int qmri_build_epi(qmri_ctx* ctx, int N, int M, double percentage, int T, int32_t* frame_ptr, int32_t* kidx,
                   int cap, int* m_out);
/* Defines P (setup_subsampling_*.m:36-42): V is T x s column-major real (main_recon_tsmis_FFT.m:129).
 * max_batch = number of slices the context can hold at once (>= 1). */
int qmri_set_operator(qmri_ctx* ctx, int N, int M, int s, int T, const double* V, const int32_t* frame_ptr,
                      const int32_t* kidx, int max_batch);
int qmri_operator_m(const qmri_ctx* ctx, int* m_out);
/* y = F.forward(x): x is N*M*s (complex if x_is_complex else real doubles), y is m complex. */
int qmri_forward(qmri_ctx* ctx, const void* x, int x_is_complex, void* y);
/* x = F.adjoint(y): y m complex -> x N*M*s complex. */
int qmri_adjoint(qmri_ctx* ctx, const void* y, void* x);
/* The same two maps for MATLAB `single` arrays (interleaved complex floats at the boundary; x real if !x_is_complex).  The arithmetic
 * stays complex double as in the reference's F (fft2 of a single array would be single in MATLAB: these entry points are at least as
 * accurate); results are rounded to single once, on the way out. */
int qmri_forward_f32(qmri_ctx* ctx, const float* x, int x_is_complex, float* y);
int qmri_adjoint_f32(qmri_ctx* ctx, const float* y, float* x);
/* device-resident variants, `batch` slices stored back to back */
int qmri_forward_dev(qmri_ctx* ctx, const void* d_x, void* d_y, int batch);
int qmri_adjoint_dev(qmri_ctx* ctx, const void* d_y, void* d_x, int batch);
/* Multi-coil extension of F (BASELINE.json configs[4]: "complex-valued multi-coil forward op").  The reference simulates ONE coil (README.md:63): no
 * interface there to replace, parity unpinned.  maps: N x M x ncoil complex doubles (coil sensitivities C_j; ncoil = 0 clears them; set after
 * qmri_set_operator).  y = A_mc x: m x ncoil complex, y(:, j) = F.forward(C_j .* x);  x = A_mc^H y = sum_j conj(C_j) .* F.adjoint(y(:, j)). */
int qmri_set_coils(qmri_ctx* ctx, int ncoil, const void* maps);
int qmri_forward_mc(qmri_ctx* ctx, const void* x, int x_is_complex, void* y);
int qmri_adjoint_mc(qmri_ctx* ctx, const void* y, void* x);
/* ... and the reconstruction on top of it (round 6; the same label: an extension, no reference counterpart, parity unpinned).  The x-update of
 * PnP_ADMM.m:102,153-171 with A replaced by A_mc -- x = lsqr(@afun, [y_mc; sqrt(r) z], tol, maxit, [], [], x0), afun: [A_mc; sqrt(r) I] -- as an
 * image-domain LSQR (coil maps act in image space, so the k-space iteration of the single-coil path does not apply), recurrences and stop rules as the
 * single-coil restatement of MATLAB's lsqr; and the PnP-ADMM loop of PnP_ADMM.m:76-146 around it (x0 NULL: x = A_mc^H y as :84; returns x as :148).
 * y_mc: m x ncoil complex doubles; z, x0, x_out: N x M x s complex doubles; lsqr_iters_out (nullable): prm->iters entries.  One slice, LSQR solver only. */
int qmri_xupdate_mc(qmri_ctx* ctx, const void* y_mc, const void* z, double r, double tol, int maxit, const void* x0, void* x_out,
                    int32_t* iters_out, int32_t* flag_out);
/* The x-update alone: x = lsqr(@afun,[y; sqrt(r) z], tol, maxit, [], [], x)  (PnP_ADMM.m:102,153-171), or the
 * closed-form minimiser when solver == QMRI_SOLVER_DIRECT.  Host buffers; x is in/out (warm start). */
int qmri_xupdate(qmri_ctx* ctx, const void* y, const void* z, double r, double tol, int maxit, int solver,
                 void* x, int32_t* iters_out, int32_t* flag_out);

/* ---- denoiser plugin: param.net, main_recon_tsmis_FFT.m:138-171 ----------------------------------- */
enum { QMRI_ARCH_UNETRES = 0, QMRI_ARCH_SEQ_CONV = 1 };
typedef struct {
    int32_t arch;            /* QMRI_ARCH_UNETRES: network_unet.py:68-117;  QMRI_ARCH_SEQ_CONV: conv3x3(+ReLU) stack */
    int32_t in_nc;           /* 10 single_level, 11 multi_level (main_test.py:245-252) */
    int32_t out_nc;          /* 10 */
    int32_t nc[4];           /* {64,128,256,512}; SEQ_CONV uses nc[0] as width */
    int32_t nb;              /* ResBlocks per stage (4); SEQ_CONV: number of conv layers */
    int32_t residual_noise;  /* denoiseImage_PnP_ADMM.m:99-104: 1 = return input - CNN(input) */
} qmri_net_desc;
/* weights: flat fp32 in state_dict() order, Conv2d OIHW / ConvTranspose2d IOHW (what export_to_onnx,
 * PyTorch_Denoiser/utils.py:444-485, serialises).  nbytes must equal 4 * qmri_net_nparams(desc). */
size_t qmri_net_nparams(const qmri_net_desc* desc);
int qmri_set_denoiser(qmri_ctx* ctx, const qmri_net_desc* desc, const float* weights, size_t nbytes,
                      int H, int W, int max_batch);
/* Weight ingestion from the ONNX file the reference loads with `Net = importONNXNetwork(denoiser_path, ...)`
 * (main_recon_tsmis_FFT.m:79-83,138), i.e. what export_to_onnx (PyTorch_Denoiser/utils.py:468-481: opset 9, weights as
 * graph initializers) writes.  Reads the Conv / ConvTranspose weights in graph order, checks that they form a UNetRes,
 * fills desc_out (arch, in_nc, out_nc, nc, nb; residual_noise = 0) and *nfloats_out, and -- when weights != NULL --
 * copies the blob qmri_set_denoiser takes (capacity_floats >= *nfloats_out).  Call once with weights == NULL to size
 * the buffer.  Host-only (no GPU, no ONNX / protobuf library); errors via qmri_last_error(NULL). */
int qmri_onnx_read_unetres(const char* path, qmri_net_desc* desc_out, float* weights, size_t capacity_floats,
                           size_t* nfloats_out);
/* out = denoiseImage_PnP_ADMM(in, net, true, residual_noise): in H x W x C x B doubles -> out H x W x out_nc x B. */
int qmri_denoise(qmri_ctx* ctx, const double* in, int H, int W, int C, int B, double* out);
/* raw network forward on device fp32 tensors [B][C][W][H] (no casts); the dominant kernel chain.  Precondition of the default
 * (f16-split) arithmetic: inputs at ordinary scale, as the [0, 1] images of PnP_ADMM.m:121 are; the call synchronises, reads the
 * range guard and -- like qmri_denoise / qmri_pnp_admm -- repeats itself on the bf16 scheme if an activation left the f16 range.
 * The repeated pass reads d_in again: d_in and d_out must not overlap (QMRI_ERR_INVALID_ARG otherwise). */
int qmri_net_forward_dev(qmri_ctx* ctx, const float* d_in, int B, float* d_out);
/* Which arithmetic the convolutions run on: *scheme_out = 2 (f16 pieces, 3 MFMA products per fp32 product) or 3 (bf16 pieces, 6
 * products: no range limits, twice the matrix time); *fallbacks_out = how often a run-time guard has moved the network from 2
 * to 3 since qmri_set_denoiser (a call that trips the guard is repeated transparently: a 2x slower call is visible here).
 * Either pointer may be NULL. */
int qmri_denoiser_scheme(const qmri_ctx* ctx, int* scheme_out, int* fallbacks_out);

/* ---- PnP-ADMM: x = PnP_ADMM(y, param), PnP_ADMM.m:1 ----------------------------------------------- */
enum { QMRI_SOLVER_LSQR = 0, QMRI_SOLVER_DIRECT = 1 };
enum { QMRI_DENOISER_SINGLE_LEVEL = 0, QMRI_DENOISER_MULTI_LEVEL = 1 };
typedef struct {
    double gamma;            /* param.gamma = sigma_squared/eta = 0.05   main_recon_tsmis_FFT.m:285-287 */
    int32_t iters;           /* param.iter = 100                          :288 */
    double cg_tol;           /* param.cg_tol = 1e-4                       :289 */
    int32_t cg_maxit;        /* 100 (literal in PnP_ADMM.m:102) */
    int32_t solver;          /* QMRI_SOLVER_LSQR reproduces the reference; DIRECT is the exact minimiser */
    int32_t denoiser_type;   /* param.denoiser_type                       :167 */
    double noise_std;        /* build_noise_map(0.01,...)                 :76,:170 */
    int32_t want_diag;       /* the two per-iteration diagnostics of PnP_ADMM.m:106-109 */
} qmri_admm_params;
/* y: m complex.  x0: N*M*s complex or NULL (=> F.adjoint(y), main_recon_tsmis_FFT.m:292).  gt: N*M*s complex or
 * NULL (param.gt_tsmi, only for the second diagnostic).  x_out: N*M*s complex (the LAST lsqr solution, as the
 * reference returns).  diag_out: iters*2 doubles or NULL.  lsqr_iters_out: iters int32 or NULL. */
int qmri_pnp_admm(qmri_ctx* ctx, const void* y, const qmri_admm_params* p, const void* x0, const void* gt,
                  void* x_out, double* diag_out, int32_t* lsqr_iters_out);
/* nslices independent slices, device-resident y / x0 / gt / x_out (slice-major); diag/lsqr outputs are host. */
int qmri_pnp_admm_dev(qmri_ctx* ctx, int nslices, const void* d_y, const qmri_admm_params* p, const void* d_x0,
                      const void* d_gt, void* d_x_out, double* diag_out, int32_t* lsqr_iters_out);

/* A slice stack from HOST buffers through one context: y nslices x m, x0 / gt nslices x N*M*s or NULL, x_out nslices x N*M*s (slice-major =
 * the columns of a MATLAB m x S matrix / the 4th dimension of an N x M x s x S array), advanced slices_per_launch at a time
 * (<= max_batch of qmri_set_operator and qmri_set_denoiser).  diag_out: nslices x iters x 2 or NULL; lsqr_iters_out: nslices x iters or NULL.
 * What `PnP_ADMM_hip(Y, param)` calls for a measurement matrix; qmri_recon_batch is the pipelined multi-GPU form. */
int qmri_pnp_admm_batch(qmri_ctx* ctx, int nslices, int slices_per_launch, const void* y, const qmri_admm_params* p, const void* x0,
                        const void* gt, void* x_out, double* diag_out, int32_t* lsqr_iters_out);
/* Multi-coil extension of the loop (see qmri_xupdate_mc above: no reference counterpart, parity unpinned). */
int qmri_pnp_admm_mc(qmri_ctx* ctx, const void* y_mc, const qmri_admm_params* prm, const void* x0, void* x_out, int32_t* lsqr_iters_out);

/* ---- LRTV option: x = FISTA_deep(data, param), main_recon_tsmis_FFT.m:273-282 -------------------------- */
/* FISTA with backtracking on 0.5 |y - F.forward(x)|^2 + K |x|_TV (FISTA_deep.m:31-104); the TV prox is unlocbox's
 * prox_tv (prox_tv.m:99-203) on the stacked image [real(x); imag(x)] of 2N rows x M*s columns (FISTA_deep.m:66,75). */
typedef struct {
    double  K;           /* param.K = 4e-5 (:275); 0 skips the prox (FISTA_deep.m:74) */
    int32_t iters;       /* param.iter = 200 (:276) */
    double  step;        /* param.step; <= 0: numel(X0)/numel(Y) (:277) */
    double  tol;         /* param.tol = 1e-4: stop when |obj - obj_prev| / obj < tol (FISTA_deep.m:103) */
    int32_t backtrack;   /* param.backtrack = 1 (:279) */
    double  prox_tol;    /* prox_tv param.tol; <= 0: 10e-4 (prox_tv.m:99) */
    int32_t prox_maxit;  /* prox_tv param.maxit; <= 0: 200 (prox_tv.m:101) */
} qmri_lrtv_params;
typedef struct {
    int32_t iters;             /* FISTA iterations performed */
    int32_t halvings;          /* 'reducing stepsize...' events */
    double  step;              /* final step size */
    double  obj;               /* last objective 0.5 |y - Fx|^2 + K |x|_TV */
    int32_t prox_calls;
    int32_t prox_iters_total;  /* inner iterations over all prox_tv calls */
} qmri_lrtv_info;
/* y: m complex doubles (ABI order); x_out: N x M x s complex doubles; info may be NULL.  Needs qmri_set_operator. */
int qmri_lrtv(qmri_ctx* ctx, const void* y, const qmri_lrtv_params* p, void* x_out, qmri_lrtv_info* info);
/* [sol, info] = prox_tv(b, gamma, param) on a real column-major R x C image (prox_tv.m:1); host buffers.
 * iters_out / obj_out (info.iter, the last objective) may be NULL. */
int qmri_prox_tv(qmri_ctx* ctx, const double* b, int R, int C, double gamma, double tol, int maxit, double* sol,
                 int32_t* iters_out, double* obj_out);
/* y = norm_tv(I) (unlocbox/utils/norm_tv.m:45-55), same layout. */
int qmri_norm_tv(qmri_ctx* ctx, const double* I, int R, int C, double* out);

/* ---- dictionary match: out = mrf_dtm_cpu(dict, data, par), mrf_dtm_cpu.m:1 --------------------------- */
/* D: K x s column-major unit-norm atoms, normD: K, lut: K x Q column-major (dict.D / .normD / .lut, :8-12).  s <= 1024: the compressed
 * atoms of the shipped script (s = 10; s <= 16 keeps a pixel tile in registers) or uncompressed fingerprints (s = T; mrf_dtm_cpu.m:41-50 takes T
 * from size(data.X)), matched by a channel-blocked GEMM. */
int qmri_set_dictionary(qmri_ctx* ctx, int K, int s, int Q, const float* D, const float* normD, const float* lut);
/* X: Npix x s complex double column-major (data.X reshaped, :50).  qmap: Npix x Q (NaN->0, :136-141);
 * pd: Npix complex single interleaved (:144-148); mt: Npix or NULL (:150-154); dm: Npix 1-based or NULL (:156-160).
 * The outputs are those of the single-precision products ip = D x^H (:91) and max(abs(ip)) with the first index winning (:92), bit for bit;
 * which 32-atom tiles need those products is decided by a filter on f16 pieces with a proven margin (qmri_debug_dict_filter switches it
 * off; dictionaries with non-finite entries are matched without it). */
int qmri_dict_match(qmri_ctx* ctx, const void* X, int Npix, float* qmap, float* pd, float* mt, int32_t* dm);
int qmri_dict_match_dev(qmri_ctx* ctx, const void* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt,
                        int32_t* d_dm);
/* The same with out.Xfit (par.f.Xout, mrf_dtm_cpu.m:95,129-134): xfit (nullable) = Npix x s complex single interleaved, column-major,
 * Xfit(p,:) = ip(dm(p)) .* D(dm(p),:) -- the matched atom scaled by the unnormalised inner product, before the division by normD (:96). */
int qmri_dict_match_xfit(qmri_ctx* ctx, const void* X, int Npix, float* qmap, float* pd, float* mt, int32_t* dm, float* xfit);
int qmri_dict_match_xfit_dev(qmri_ctx* ctx, const void* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt,
                             int32_t* d_dm, float* d_xfit);

/* ---- TSMI synthesis from quantitative maps: main_synthesize_tsmis.m:54,82-100 (mode 'real') ------------ */
/* I = knnsearch(KDTreeSearcher(dict.lut), qm(:,1:2)); X = real(dict.D(I,:)) .* dict.normD(I) .* abs(qm(:,3)); X .* sign(X(:,:,1)).
 * qmap: Npix x 3 doubles column-major (T1, T2, PD, in the units of dict.lut); X_out: Npix x s singles column-major;
 * idx_out (nullable): the 1-based nearest entry.  Uses the dictionary of qmri_set_dictionary (Q >= 2). */
int qmri_synthesize_tsmi(qmri_ctx* ctx, const double* qmap, int Npix, float* X_out, int32_t* idx_out);
/* mode 'complex' of the same script (main_synthesize_tsmis.m:27,100-103): X = real(dict.D(I,:)) .* dict.normD(I) .* qm(:,3) with a complex
 * PD, no abs and no sign alignment, stored as cat(3, real(X), imag(X)).  qmap: Npix x 3 (T1, T2, real(PD)); pd_imag: Npix or NULL
 * (imaginary part of PD); X_out: Npix x 2s singles column-major (the s real channels, then the s imaginary ones). */
int qmri_synthesize_tsmi_complex(qmri_ctx* ctx, const double* qmap, const double* pd_imag, int Npix, float* X_out, int32_t* idx_out);

/* ---- slice batches over several GPUs of one node (slices are independent; no collective) ------------ */
typedef struct {
    int32_t N, M, s, T;
    const double* V;                 /* T x s */
    const int32_t* frame_ptr;        /* T+1 */
    const int32_t* kidx;             /* m */
    const qmri_net_desc* net;
    const float* weights;
    size_t weights_nbytes;
    int32_t K, Q;                    /* dictionary (K == 0: skip the match) */
    const float* D;
    const float* normD;
    const float* lut;
    qmri_admm_params admm;
    int32_t slices_per_launch;       /* slices batched through the denoiser on one GPU (>= 1) */
} qmri_problem;
/* Y: nslices x m complex (host).  X_out: nslices x N*M*s complex.  qmap_out: nslices x Npix x Q or NULL,
 * pd_out: nslices x Npix complex single or NULL.  One host thread + one context per device in devs[]. */
int qmri_recon_batch(int ndev, const int* devs, int nslices, const qmri_problem* prob, const void* Y,
                     void* X_out, float* qmap_out, float* pd_out, char* errbuf, size_t errbuf_len);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------------ */
typedef struct {
    double ms_xupdate, ms_denoiser, ms_elementwise, ms_diag, ms_match;   /* hipEvent time per stage */
    double ms_conv3x3;          /* level 2: summed duration of the 3x3 convolution launches, each taken from its own dispatch timestamps: one
                                 * launch per layer (a split-K layer: from the convolution's start to its reduce kernel's end), or one
                                 * resident-tile launch of a whole run of layers (with whatever else rides in it) as ONE unit */
    int64_t n_conv3x3;          /* number of those units */
    int64_t lsqr_iters;         /* LSQR iterations executed */
    int64_t admm_iters;
    double ms_tv_iter;          /* LRTV: summed duration of the prox_tv iteration kernel's launches (level 2) */
    int64_t n_tv_iter;
    double flop_conv3x3;        /* fp32-equivalent algorithmic work of the units in ms_conv3x3: 2 Cout Cin taps Hout Wout B per layer inside them */
    double ms_conv2x2;          /* level 2: the 2x2 / stride-2 (transposed) convolutions launched on their own */
    int64_t n_conv2x2;
    double flop_conv2x2;
    double ms_lsqr_kernels;     /* level 2: the LSQR iteration kernels alone (k_ks_a start -> k_ks_b end per iteration, or a whole k_ks_persist launch) */
    int64_t n_lsqr_launches;
    double ms_net_forward;      /* level 2: whole forward passes of the network, first launch to last (stream events) */
    int64_t n_net_forward;
} qmri_profile;
int qmri_profile_enable(qmri_ctx* ctx, int level);   /* 0 off, 1 per stage (synchronises at every stage boundary), 2 also per conv3x3 launch,
                                                        3 stage MARKS: event records at the stage boundaries that are read only after the call's own final
                                                        synchronisation -- the stage split of a timed run without a wait inside it (batches; an event record
                                                        between two dependent kernels costs a few microseconds, so not for the one-slice headline) */
int qmri_profile_get(qmri_ctx* ctx, qmri_profile* out, int reset);

/* Health of a context (no counterpart in the reference, which reconstructs one slice per run -- main_recon_tsmis_FFT.m:37-38 -- and has no
 * alternative paths): which of the library's self-checking fast paths are armed, how often one of them gave up and the work was repeated on the
 * slower path, and what the most recent qmri_pnp_admm_dev call cost.  A reconstruction that is slow for one of THESE reasons says so here; the
 * results are the same either way (every fallback is tested for bits).  Counters run from qmri_create; the denoiser's from qmri_set_denoiser. */
typedef struct {
    int denoiser_scheme;        /* 2 = f16 x 3 products, 3 = bf16 x 6 products, 0 = no denoiser set */
    int denoiser_fallbacks;     /* switches f16 -> bf16 by the range / low-magnitude guards since qmri_set_denoiser */
    int resident_armed;         /* 1 = the resident-tile launch of the full-resolution ResBlocks is in use (one slice per launch only) */
    int resident_timeouts;      /* ring hand-offs that timed out since qmri_set_denoiser (each: the forward pass was repeated, one launch per layer) */
    int lsqr_one_launch;        /* 1 = the one-launch LSQR iteration is armed, 0 = switched off (by a time-out or by the caller), -1 = not decided yet */
    int lsqr_timeouts;          /* one-launch LSQR kernels that gave up waiting for a partial sum (each: the solve or the reconstruction was repeated) */
    int repeated_calls;         /* qmri_pnp_admm* calls that ran their reconstruction twice (any of the reasons above) */
    int reserved;
    double last_call_wall_ms;   /* host wall clock of the most recent qmri_pnp_admm_dev call, entry to return */
    double last_call_stage_ms[4]; /* its x-update / denoiser / elementwise / diagnostics stages on the device (profile level 1 or 3 only, else zeros) */
    double set_denoiser_ms[3];  /* the most recent qmri_set_denoiser on the host clock: weight splitting, packing and upload of all layers / tensors and
                                   buffers / the two-pass calibration probe (the reference loads its network once per run: main_recon_tsmis_FFT.m:138-152) */
} qmri_health;
int qmri_get_health(const qmri_ctx* ctx, qmri_health* out);

/* Diagnostics (no counterpart in the reference): in-kernel 100 MHz phase stamps, recorded only when the library was
 * started with the knobs lsqr_stamps / conv_stamps set (qmri_debug_knob, QMRI_DEBUG; otherwise QMRI_ERR_STATE).  `out` receives 2*512*16 and 4096*11
 * 64-bit values; layouts are those read by tools/lsqr_stamps.py and tools/conv6_stamps.py. */
int qmri_debug_lsqr_stamps(qmri_ctx* ctx, unsigned long long* out);
int qmri_debug_conv_stamps(qmri_ctx* ctx, unsigned long long* out, int reserved);
/* Test / A-B hook: the LSQR x-update (PnP_ADMM.m:102) runs all its iterations in ONE launch where the operator's work units are resident at
 * once (default; same bits as the two-launch iteration); on = 0 selects the two-launch iteration (also the knob lsqr_persist = 0); on = 2 makes
 * the one-launch kernel lose a partial sum on purpose: its waits time out, the library reports it on stderr and repeats the solve with the
 * two-launch iteration (the recovery path, tested). */
int qmri_debug_lsqr_persist(qmri_ctx* ctx, int on);
/* Test / A-B hook for the dictionary match (mrf_dtm_cpu.m:91-92): on = 1 (default) puts the f16 filter in front of the exact single-precision
 * products (the result is the same, bit for bit: the filter only decides which 32-atom tiles need the exact products), on = 0 computes
 * every product exactly.  margin_scale (default 1) multiplies the filter's margin: the tests shrink it to measure how far the proven
 * margin is from the first wrong answer. */
int qmri_debug_dict_filter(qmri_ctx* ctx, int on, float margin_scale);
/* Test / A-B hook for the denoiser's convolutions: on = 1 (default) runs the ResBlocks of the full-resolution level of a one-slice forward pass as
 * ONE launch with LDS-resident tiles (same bits as one launch per layer); on = 0 selects one launch per layer (also the knob conv_resident = 0); on = 2
 * makes one tile withhold its hand-off on purpose: its neighbours' waits time out, the library reports it on stderr, repeats the call with one
 * launch per layer and keeps the resident form off (the recovery path, tested).  timeouts_out (or NULL): hand-off time-outs seen so far. */
int qmri_debug_conv_resident(qmri_ctx* ctx, int on, int* timeouts_out);
/* The one entry point of the process-wide A/B and diagnostic switches ("knobs": tile configurations, the fused launches, in-kernel stamps of the
 * diagnostic builds ...; names and defaults: csrc/api_core.cpp g_knob_defs).  The same switches can be set at start-up through the library's
 * only environment variable, QMRI_DEBUG="name=value,name=value".  Every default is the product's behaviour; an unknown name is
 * QMRI_ERR_INVALID_ARG (message: qmri_last_error(NULL)).  Knobs are read when a plan is made or a launch is issued: set them before
 * qmri_set_operator / qmri_set_denoiser. */
int qmri_debug_knob(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* QMRI_H */

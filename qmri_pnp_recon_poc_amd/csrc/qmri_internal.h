// qmri_internal.h -- shared declarations of libqmri.so (host side + kernel launch prototypes).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/qmri.h"

// ---------------------------------------------------------------------------------------------------
// error handling: nothing throws across the C ABI
// ---------------------------------------------------------------------------------------------------
struct qmri_ctx;
void qmri_set_error(qmri_ctx* ctx, const char* fmt, ...);

#define QMRI_HIP(ctx, expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            qmri_set_error((ctx), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return QMRI_ERR_HIP;                                                                  \
        }                                                                                         \
    } while (0)

#define QMRI_CHECK_ARG(ctx, cond, msg)                                                            \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            qmri_set_error((ctx), "invalid argument: %s", (msg));                                 \
            return QMRI_ERR_INVALID_ARG;                                                          \
        }                                                                                         \
    } while (0)

#define QMRI_TRY(expr)                                                                            \
    do {                                                                                          \
        int _s = (expr);                                                                          \
        if (_s != QMRI_OK) return _s;                                                             \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// A/B and diagnostic switches ("knobs").  ONE environment variable, QMRI_DEBUG="name=value,name=value" (read once per process), and
// one entry point, qmri_debug_knob(name, value) (include/qmri.h), set them; every default is the product's behaviour.  The table of
// names and defaults is in api_core.cpp (g_knob_defs, same order as this enum).
// ---------------------------------------------------------------------------------------------------
enum QmriKnob {
    K_CONV_SCHEME,        // 2: f16 x 3 products (default), 3: bf16 x 6 products from the start
    K_CONV_F32,           // 1: every layer on the f32-MFMA kernels (conv_kernels.hip)
    K_CONV_WT,            // write-through (sc1) output stores of the conv kernels
    K_CONV_XCD,           // XCD-aware tile order
    K_CONV_PERSIST,       // k_conv6p for slice batches
    K_CONV_SPLITK,        // split-K at the deep levels
    K_CONV_MIDCFG, K_CONV_DEEPCFG, K_CONV_DEEPKS,   // tile configuration of the 56 x 56 / 28 x 28 level, largest K split of the latter
    K_CONV_RESIDENT,      // k_conv6r: the full-resolution level's layers as one launch with resident tiles
    K_RES_HEAD, K_RES_TAIL, K_RES_DOWN,             // ... with the head / tail / down-sampling convolution inside
    K_RES_DELAY,          // ... pause before the ring fetch, in units of 64 clocks
    K_RES_REARM,          // ... clean one-launch-per-layer passes after a hand-off time-out before the resident form is tried again
    K_RES_STAMPS, K_CONV_STAMPS, K_CONV_STAMP_LAUNCH, K_LSQR_STAMPS,   // diagnostic builds' in-kernel stamps
    K_CONV_MT2, K_CONV_OCC,                         // f32-MFMA fallback kernels: tile / occupancy choices
    K_FUSE_EW,            // the fused launches between the network and the solve
    K_LSQR_PERSIST,       // k_ks_persist: all LSQR iterations of a solve in one launch
    K_LSQR_FOLD,          // ... and the first Golub-Kahan step (k_ks_b<INIT>) inside that launch
    K_DICTW_LSP,
    K_VERBOSE,            // calibration / guard decisions on stderr
    K_PACK_GPU,           // qmri_set_denoiser: weights split and ordered on the device (1, default) or on the host (0: the round-1 packers, same bits)
    K_COUNT
};
int qmri_knob(QmriKnob k);

// ---------------------------------------------------------------------------------------------------
// forward operator (struct F): device-side description
// ---------------------------------------------------------------------------------------------------
struct KEntry {          // one sample of the k-sorted measurement list
    uint16_t kw;         // k-space column (second FFT dimension)
    uint16_t t;          // frame
};


struct OpDev {
    int N, M, s, T, m;
    const double* Vt;        // [T][s]  V(t,c), frame-major
    const KEntry* ent;       // [m]     k-sorted: (kh, kw, t) ascending
    const int32_t* perm;     // [m]     k-sorted position -> frame-major measurement index (ABI order)
    const int32_t* kptr;     // [N*M+1] CSR over k' = kh*M + kw into ent
    const double2* tw;       // [N]     exp(-2*pi*i*j/N)
    const int32_t* kslot;    // [N*M]   DIRECT solver: slot of k' among sampled locations or -1
    const double* ginv;      // [nsampled][s*s] (G_k + r I)^-1, symmetric, row-major
};

// ---------------------------------------------------------------------------------------------------
// k-space LSQR (kslsqr_kernels.hip): work partition over the sampled k locations ("slots", k' order)
// ---------------------------------------------------------------------------------------------------
struct KSample { uint16_t ls, t; };              // slot of a sample relative to its block's first slot; frame
struct KsGroup { uint16_t ls, b, e, pad; };      // <= gcap samples of one slot; b, e relative to the block's first sample
// The shape of a work unit: slots, samples, samples per scatter group, lanes that share a group, scatter groups per block (= ecap / gcap + scap).
// Picked when the operator is planned (api_core.cpp); kslsqr_kernels.hip instantiates its kernels for each (KsCaps):
//   0  dense masks (a sampled k location carries >= 8 samples on average: the spiral), slice batches and single slices whose units fit the chip
//   1  sparse masks (EPI: every k location, 2.7 samples each), ONE slice: 256 slots per unit so that <= 250 units result (one-launch iteration)
//   2  dense, very many samples per k (cut0: 56), ONE slice: 2560 samples per unit, <= 256 units (one-launch iteration)
//   3  sparse masks, slice batches (and whatever 1 does not fit): small units
// Sparse shapes give every scatter group (<= 4 samples of one slot) to ONE lane; the dense ones share a group of <= 32 samples among 8 lanes.
struct KsCapsHost { int scap, ecap, gcap, sl, gcapb; };
constexpr KsCapsHost KS_CAPS[4] = {{64, 1024, 32, 8, 96}, {256, 768, 4, 1, 448}, {64, 2560, 32, 8, 144}, {64, 256, 4, 1, 128}};
constexpr int KS_NCAPS = 4;
struct KsUnit { int32_t s0, s1, e0, e1, g0, g1, pad0, pad1; };   // a work unit: its slots, samples and groups (one 32-byte scalar load)
struct LsqrState;
struct KsDev {
    int ns, G;                                   // sampled k locations; blocks of the iteration kernels
    int caps;                                    // shape of the work units: index into KS_CAPS
    const KsUnit* unit;                          // [G]    slot / sample / group ranges of each block
    const KSample* es;                           // [m]
    const KsGroup* grp;                          // scatter groups, block after block
    const int32_t* sgrp;                         // [ns+1] first group of each slot
    LsqrState* st;                               // [B]
    LsqrState* hst;                              // [B] pinned host copy: the kernel that changes iter / done / flag writes them here too (no copy launch per x-update)
    double* pu[2];                               // [B][2G] partial |u|^2 (image part, then samples), by iteration parity
    double* pv[2];                               // [B][G]  partial |v|^2
    double* pinit;                               // [B][2N] partial |u|^2 of the initial residual, per k-row
    double* pR;                                  // [B][N]  partial R
    const double* pz; int nblk_z;                // [B][nblk_z] partial |z|^2
    double2* cx; double2* cv; double2* cd; double2* cub;   // [B][ns*s] x, v (not normalised), d, u(m+1:end) on the sampled k
    double2* ut; const double2* yk;              // [B][m]
    double2* xhat; double2* zhat;                // [B][n] unitary spectra, layout [c][kh][kw]
    double2* xhat_out;                           // [B][n] assembled solution spectrum (becomes xhat of the next solve)
    double* pdiag;                               // [B][N] partial ||y - A x||^2 or null
    double sr, tol;
    int maxit, ii, vcap;
    int b0;                                      // k_ks_persist: first slice of this launch (a batch goes through it a few slices at a time)
    unsigned long long* stamps;
};

// LSQR (PnP_ADMM.m:102) device state, one per slice
struct LsqrScalars {
    double c, s, phibar, normr, norma, factor;   // recurrences; normar = alpha * factor
    double thet, rho, phi, beta, alpha;          // values A2 needs for the current iteration
    double ua, ub, uc, ue;                       // k-space solver: v, u(m+1:end), d, x - x0 on the never-sampled k, as multiples of (zhat - xhat0)
};
struct LsqrState {
    LsqrScalars sc[2];       // ping-pong by iteration parity
    double n2b, tolb;
    double ny2;              // ||y||^2
    double R;                // k-space solver: sum over never-sampled k of |zhat - xhat0|^2
    double ue_final;         // k-space solver: ue after the last x update
    int32_t iter, done, flag, pad;
};

enum { DC_PLAIN = 0, DC_FWD_H_ONLY = 1, DC_DIAG = 3, DC_DIRECT = 4, DC_SPECTRUM = 5 };
constexpr int DC_SORT_BLOCKS = 64;   // workgroups of k_sort_y per slice (partial sums of |y|^2 added in block order)

struct LsqrDev {
    LsqrState* st;           // [B]
    double* pz;              // [B][nblk_z] partial sums of |z|^2
    double2* yk;             // [B][m]  y in k-sorted order
    double* py;              // [B][DC_SORT_BLOCKS] partial sums of |y|^2 (k_sort_y)
    int nblk_z;
};

// ---------------------------------------------------------------------------------------------------
// denoiser: one packed layer of the conv engine
// ---------------------------------------------------------------------------------------------------
enum ConvKind { CONV_3X3 = 0, CONV_DOWN = 1, CONV_UP = 2, CONV_3X3N = 3 };   // 3X3N: 3x3 with 32-channel chunks (Cin <= 64)
struct ConvLayer {
    ConvKind kind;
    int Cin, Cout;           // logical channels (UP: Cout = real output channels)
    int cin_pad, n_ct;       // padded input channels, number of 32-row output tiles in the packed weights
    int MT;                  // 32-row MFMA tiles per wave
    float* wp;               // packed weights (device)
    size_t wp_floats;
    void* d_tab;             // tile table {cout tile, ow0, oh0, batch} for (tab_B, tab_MT)
    int tab_B, tab_MT;
    void* wp6 = nullptr;     // weights split into f16 pairs (sp6 == 2) or bf16 triples (sp6 == 3), MFMA A-fragment order (conv6_kernels.hip)
    int sp6 = 2;             // pieces per fp32 operand of the matrix-core path
    float w6_descale = 1.f;  // f16 scheme: 2^-k of the power-of-two the packed weights carry (conv6_kernels.hip)
    int nchunk6 = 0, n_ct6 = 0;   // 16-channel chunks (2x2 layers: K steps), 64-row output tiles of wp6
    int nsteps6s = 0;             // 2x2 layers: K steps that carry weights (nchunk6 is padded to a multiple of 3)
    int index = -1;               // position in NetPlan::layers (row of the |output| report, conv6_kernels.hip ACT_LOW)
    size_t w_off = 0;             // first weight of this layer in the caller's flat blob (floats)
};

// activation tensor in HBM: [B][Cal][W+2][hp] fp32, h fastest, permanent zero halo, channels >= C are zero.
// A row holds h0 - 1 unused floats, the top halo, the H interior values (starting at float h0), the bottom halo and padding
// up to the pitch hp; h0 and hp are multiples of 32, so every interior row starts on a 128-byte line.  Kernels address the
// tensor through base1(), relative to which the interior starts at +1 as in a plain [W+2][H+2] layout with pitch hp.
struct PTensor {
    float* p = nullptr;
    int C = 0, Cal = 0, H = 0, W = 0;
    int hp = 0, h0 = 1;
    // BLOCKED: element (c, w, h) of the padded image lives at ((c/8 * plane + w * hp + h) * 8 + c%8 instead of c * plane + w * hp + h -- the
    // interior format of the matrix-core conv kernels (conv6_kernels.hip: 8 channels of a pixel = 32 contiguous bytes = two 16-byte
    // requests).  Same allocation either way (Cal % 8 == 0); a property of the current forward pass, set by net_forward_padded.
    bool blk = false;
    float* base1() const { return p + (h0 - 1); }
    float* fbase() const { return p + (size_t)(h0 - 1) * (blk ? 8 : 1); }      // halo origin of row 0 in the tensor's current format
    size_t plane() const { return (size_t)hp * (W + 2); }
    size_t batch_stride() const { return (size_t)Cal * plane(); }
};

// per-layer reduction of the |output| reports of a forward pass (conv6_act.h); nlayers = 0: nothing to do
struct ActCheckArgs { const float* slots; int* count; float* ref; int record; unsigned* range_flag; unsigned* host_words; int nlayers; };

struct NetPlan {
    qmri_net_desc desc{};
    int H = 0, W = 0, maxB = 0;
    std::vector<ConvLayer> layers;
    // activation buffers (device, padded planes): per UNet level the skip tensor x and two work tensors a, t
    PTensor x[4], a[4], t[4];
    PTensor in32;            // normalised network input (in_nc channels, allocated up to the head's padded Cin)
    PTensor out32;           // network output (out_nc channels)
    std::vector<float*> allocs;
    unsigned* d_counter = nullptr;   // tile-queue counter of the persistent conv kernels
    unsigned counter_base = 0;       // host mirror of its value after the launches issued so far
    bool counter_by_memset = false;  // a launch under a stream capture: the queue is reset before every launch instead
    bool force_f32 = false;             // calibration (qmri_set_denoiser): run the f32-MFMA kernels whatever the scheme
    unsigned* d_range_flag = nullptr;   // f16 scheme: raised by a conv kernel whose output leaves the f16-splittable range
    unsigned* h_range_flag = nullptr;   // pinned host words written by k_act_check at the end of every forward pass: [0] overflow bit, [1 + layer] low-magnitude bit
    int h_range_words = 0;              // ... how many (1 + layers must fit, else the ADMM loop copies the device flag instead)
    int fallbacks = 0;                  // times a run-time guard moved the network from the f16 to the bf16 scheme since qmri_set_denoiser
    float* d_act_slots = nullptr;       // f16 scheme: largest |output| per (reporting launch, wave) of the current forward pass (conv6_kernels.hip, ACT_LOW)
    int* d_act_count = nullptr;         // ... valid slots per reporting launch
    float* d_act_ref = nullptr;         // ... the magnitude of every layer under the set-up probe (calibrated reference)
    int act_cap = 0;                    // rows of the three arrays (= layers)
    bool act_on = false, act_record = false;   // a reporting forward pass is under way; it is the calibration probe
    // the PnP-ADMM loop lets the kernel after the forward pass finish the per-layer |output| report (conv6_act.h) instead of launching k_act_check
    bool act_defer = false, act_pending_valid = false;
    ActCheckArgs act_pending{};
    int sp6 = 2;                     // scheme the layers are packed for
    bool blk_ok = false;             // the network's interior tensors may be BLOCKED (PTensor::blk): every layer runs on the conv6 kernels, channels % 8 == 0
    int interior_fmt = -1;           // format the interior tensors were last written in (-1: untouched zeros, 0 planar, 1 blocked): a change re-zeroes them (halo)
    std::vector<float> w_host;       // the caller's weights (kept to re-pack the layers for the other scheme; knob pack_gpu = 0 only)
    float* d_wflat = nullptr;        // ... on the device, in the caller's order (knob pack_gpu = 1, default: the packers run there)
    const float* w_begin = nullptr;  // (inside qmri_set_denoiser only: start of the caller's blob, for the layers' offsets)
    std::vector<float> w_max;        // ... and every layer's largest |w| (from the device): the f16 scheme's per-layer scale
    float* d_c6part = nullptr;       // split-K partial outputs of k_conv6 (conv6_kernels.hip), grown on demand
    size_t c6part_floats = 0;
    void* d_stamps = nullptr;        // diagnostic: per-workgroup timing stamps of the last conv launch (knob conv_stamps = 1)
    // resident-tile ResBlock runs (conv6_kernels.hip k_conv6r): the exchange buffer of the tiles' edge pixels (two layer parities x tiles x 23.5 KB), the
    // running tag of its granules (never reset), and the switch a timed-out hand-off turns off for the life of the plan
    unsigned char* d_res_xbuf = nullptr;
    int res_tiles = 0;
    unsigned res_epoch = 0;
    bool res_off = false;
    bool res_forced_off = false;     // ... by qmri_debug_conv_resident(ctx, 0, ..): never re-armed
    double setup_ms[3] = {0, 0, 0};  // qmri_set_denoiser: weight packing + upload / tensors and buffers / calibration probe (qmri_get_health)
    int res_timeouts = 0;            // hand-off time-outs since qmri_set_denoiser (three: the form stays off)
    int res_clean = 0;               // clean one-launch-per-layer passes since the last one (K_RES_REARM of them re-arm the form)
    int res_drop = 0;                // test hook: bit 0 = tile 0 withholds its hand-off
    double* d_io = nullptr; size_t io_cap = 0;   // qmri_denoise: device staging of the caller's doubles (elements), kept between calls
    void* d_res_stamps = nullptr;    // diagnostic: phase stamps of the last k_conv6r launch (knob res_stamps), 4 x 2 x R_MAXL (10) x 8 of 1024 values
    bool ready = false;
};

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
struct OpHost {
    bool ready = false;
    int N = 0, M = 0, s = 0, T = 0, m = 0, maxB = 0, nsampled = 0;
    double* d_Vt = nullptr; KEntry* d_ent = nullptr; int32_t* d_perm = nullptr; int32_t* d_kptr = nullptr;
    double2* d_tw = nullptr; int32_t* d_kslot = nullptr; double* d_ginv = nullptr;
    double2* d_coils = nullptr; int ncoil = 0;   // multi-coil extension: [ncoil][N*M] sensitivity maps (qmri_set_coils)
    KsDev ks{};                         // k-space LSQR plan + state (device pointers owned here)
    bool xhat_valid = false;            // ks.xhat holds the spectrum of d_x
    double ginv_r = -1.0;
    std::vector<double> V;              // T x s column-major (host copy)
    std::vector<int32_t> frame_ptr, kidx, kptr_h, perm_h;
    std::vector<KEntry> ent_h;
    // workspaces
    double2* d_tmp = nullptr;           // [B][s][N][M]
    double2* d_xa = nullptr; double2* d_xb = nullptr;   // [B][n] staging for host-pointer entry points
    double2* d_ya = nullptr;            // [B][m]
    LsqrDev ls{};
    // ADMM state
    double2* d_x = nullptr; double2* d_u = nullptr; double2* d_vv = nullptr; double2* d_z = nullptr;
    double2* d_chat = nullptr;          // DIRECT: unitary FFT2 of A'y
    double* d_mm = nullptr;             // [B][nblk_z][2] min/max partials
    double* d_norm = nullptr;           // [B][2] lo, range
    double* d_diag = nullptr;           // [B][iters][2]
    double* d_pd = nullptr;             // diag partials
    int32_t* h_flags = nullptr;         // pinned host mirror of done flags
    LsqrState* h_state = nullptr;       // pinned
    LsqrState* h_ring = nullptr;        // pinned [ADMM iteration][slice]: LSQR state of a reconstruction whose host never waits (api_net.cpp)
    size_t h_ring_cap = 0;
};

struct DictHost {
    bool ready = false;
    int K = 0, s = 0, Q = 0, ntiles = 0;
    // s <= 16 (dict_kernels.hip): d_pack = [ntiles][lane][4 | 8] MFMA A-fragments.
    // s  > 16 (wide = 1, dictw_kernels.hip): d_pack = [ntiles (padded to 128-atom tiles)][G8][lane][4], G8 = groups of 8 channels (s padded to 16)
    int wide = 0, G8 = 0;
    float* d_pack = nullptr;
    float* d_xp = nullptr; size_t xp_cap = 0;             // wide: the pixels as single-precision B-fragments, [tile32][G8][re | -im][lane][4]  (bytes)
    float4* d_win = nullptr; size_t win_cap = 0;          // per pixel (re ip, im ip, atom, |ip|) of the winner, kept when Xfit is asked for (bytes)
    int slots_w = 0;                                      // workgroups of k_dictw_match the device holds at once
    float* d_normD = nullptr; float* d_lut = nullptr;
    float4* d_part = nullptr; size_t part_cap = 0;        // (|ip|, atom index, re, im) per (atom part, pixel) when the atoms are split over workgroups (bytes)
    int slots = 0, slots_f = 0;                           // workgroups of k_dict_match / k_dict_match_f the device holds at once (occupancy query, first launch)
    // f16 filter in front of the exact products (dict_kernels.hip): hi / lo pieces of g D as A-fragments, [ntiles][hi | lo][64 lanes] of 16 bytes;
    // nullptr when D holds a non-finite entry (no filter then).  marg_coef = 2^-14 (g R)^2, R = largest row 2-norm of D.
    uint4* d_pack16 = nullptr;
    int* d_gmax = nullptr; size_t gmax_cap = 0;           // per pixel: largest filtered |ip|^2 seen by any wave (float bits), -1 at launch (bytes)
    float marg_coef = 0.f;
    int filter_on = 1; float margin_scale = 1.f;          // qmri_debug_dict_filter
};

struct qmri_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    OpHost op;
    NetPlan net;
    DictHost dict;
    int prof_level = 0;
    qmri_profile prof{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_state = nullptr;      // LSQR state copied to the host
    // profile level 2: the 3x3 conv kernels are launched with hipExtLaunchKernelGGL and a (start, stop) event pair each,
    // which takes the timestamps of the kernel's own dispatch packet -- the duration rocprofv3 --kernel-trace reports --
    // without putting extra packets between dependent kernels (event records in the stream add 3-5 us per kernel)
    std::vector<hipEvent_t> chain;      // pairs: [2i] start, [2i+1] stop
    size_t chain_n = 0;                 // events handed out in the current forward
    std::vector<int> chain_kind;        // which accumulator of qmri_profile a pair's duration goes to (PROF_*)
    std::vector<double> chain_flop;     // fp32-equivalent algorithmic flop of the launch(es) a pair brackets (convolutions)
    bool conv6_attr[4][2] = {};         // dynamic LDS size of k_conv6<CFG, SP> allowed
    bool conv6p_attr[2][3] = {{false, false, false}, {false, false, false}};   // ... of k_conv6p<CFG, NRES>
    bool conv6r_attr = false;           // ... of k_conv6r
    bool ks_lds_attr[2] = {false, false};   // large dynamic LDS allowed for the k-space LSQR kernels
    int lsqr_pred = 20;                 // predicted LSQR iteration count for launch chunking
    int ks_persist = -1;                // k_ks_persist (all LSQR iterations in one launch): -1 = knob lsqr_persist (default on), 0 / 1 set by qmri_debug_lsqr_persist
    int ks_persist_cap = -1;            // ... workgroups of it the device holds at once (occupancy query, first use)
    void* d_ks_gran = nullptr;          // ... its tagged partial sums (granules)
    unsigned ks_tag = 16;               // ... tag of the next solve's first granule (tags are unique across solves)
    // round 6 (qmri_get_health): what a slow or repeated reconstruction was doing.  Counters since qmri_create.
    int ks_timeouts = 0;                // one-launch LSQR kernels that gave up waiting for a partial sum (the solve / reconstruction was repeated)
    int admm_repeats = 0;               // qmri_pnp_admm_dev calls that ran their reconstruction a second time (LSQR time-out, f16 range guard, hand-off time-out)
    // profile level 3: stage MARKS -- event records at the stage boundaries of the PnP-ADMM loop, never waited for inside the loop; resolved after
    // the call's own final synchronisation into prof.ms_* and last_call_ms (level 1 synchronises at every boundary and is for stage splits only)
    std::vector<hipEvent_t> marks;      // [2i] start, [2i+1] stop of a stage interval
    std::vector<int> mark_kind;         // 0 x-update, 1 denoiser, 2 elementwise, 3 diagnostics
    size_t marks_n = 0;
    double last_call_ms[4] = {0, 0, 0, 0};   // stage times of the most recent qmri_pnp_admm_dev call (levels 1 and 3)
    double last_call_wall_ms = 0;            // ... its host wall clock, entry to return (always)
    int conv_ncu = 0;                   // f32-MFMA conv kernels (conv_kernels.hip): CU count and resident workgroups per CU by (kind, MT)
    int conv_occ[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
};

OpDev qmri_opdev(const qmri_ctx* ctx);

// ---------------------------------------------------------------------------------------------------
// kernel launchers (dc_kernels.hip)
// ---------------------------------------------------------------------------------------------------
bool dc_size_supported(int N);
// forward:  src [B][n] -> (mode-dependent) ; tmp workspace [B][n]
int dc_launch_fwd(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int mode, int B, const double2* src, double2* tmp,
                  double2* y_out, double* pdiag);
int dc_launch_adj(qmri_ctx* ctx, const OpDev& op, int B, const double2* y_in, double2* tmp, double2* dst);
int dc_launch_adj_h(qmri_ctx* ctx, const OpDev& op, int B, const double2* tmp, double2* dst, const double2* u = nullptr,
                    double* mm = nullptr);   // inverse h-pass only (+ partial min / max of real(dst + u) per workgroup)
// the step between the denoiser and the next x-update in one launch (dc_kernels.hip, k_dual_fwd_h)
struct DualArgs {
    const float* out32; const float* in32; int php, pplane; size_t out_bs, in_bs; int residual_noise;
    const double* norm; const double2* x; double2* u; double* pz;
};
int dc_hpass_blocks(const OpDev& op);
// what the ADMM loop fuses into the launches around a solve (qmri_lsqr_run)
struct LsqrFuse { int z_hpass_nblk = 0; const double2* mm_u = nullptr; double* mm = nullptr; };
int dc_launch_dual_fwd_h(qmri_ctx* ctx, const OpDev& op, int B, const DualArgs& d, const ActCheckArgs& ac, double2* tmp);
// k-space LSQR (kslsqr_kernels.hip)
int ks_launch_init(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, const double2* hpass_tmp = nullptr, bool first_step = true);
int ks_launch_iter(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B);
int ks_launch_final(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, double2* tmp);
int ks_persist_plan(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, int* per_launch);                             // slices per k_ks_persist launch (0: not applicable)
int ks_launch_persist(qmri_ctx* ctx, const OpDev& op, const KsDev& ks, int B, void* gran, unsigned tag0, bool init_here, bool* ran);   // all iterations in one launch
size_t ks_gran_bytes(int G, int B);   // tmp <- conj-domain inverse w-pass of xhat
int ks_lds_fits(qmri_ctx* ctx, int N, int s, int M, int vcap, int caps, bool* ok);   // V (vcap doubles) fits the LDS of every k-space LSQR kernel with unit shape `caps`
int dc_launch_direct(qmri_ctx* ctx, const OpDev& op, int B, const double2* z, const double2* chat, double r,
                     double2* tmp, double2* x_out);
// y (ABI order) -> k-sorted order, plus ||y||^2 into state
int dc_launch_sort_y(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* y);
// z = v - u, partial ||z||^2
int dc_launch_prepare_z(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* v, const double2* u,
                        double2* z);
// elementwise ADMM stages (PnP_ADMM.m:115-121,138,144)
int ew_launch_minmax_normalise(qmri_ctx* ctx, int B, size_t n, int plane, int H, int s, int multi_level, double noise_std,
                               const double2* x, const double2* u, double* mm, double* norm, int nblk, const PTensor& in32, bool mm_ready = false);
int ew_launch_unnormalise_dual(qmri_ctx* ctx, int B, size_t n, int plane, int H, const PTensor& out32, const PTensor& in32,
                               int residual_noise, const double* norm, const double2* x, double2* u, double2* v, double2* z,
                               double* pz, int nblk_z);   // also z = v - u and the partials of ||z||^2 for the next x-update
int ew_launch_diag(qmri_ctx* ctx, const OpDev& op, const LsqrDev& ls, int B, const double2* x, const double2* gt,
                   double* pd, double* diag_slot, int iters_total, int it);
int ew_launch_pack(qmri_ctx* ctx, int B, int C, int H, int W, const void* src, int src_is_double, const PTensor& dst, float scale = 1.f);
int ew_launch_absmax(qmri_ctx* ctx, const float* x, const float* y, size_t n, unsigned* d_out);   // calibration: max |x - y| (y may be NULL) as a bit pattern
int ew_launch_unpack(qmri_ctx* ctx, int B, int C, int H, int W, const PTensor& out32, const PTensor& in32, int residual_noise,
                     void* dst, int dst_is_double, float scale = 1.f);
int ew_launch_real_to_complex(qmri_ctx* ctx, size_t count, const double* in, double2* out);
// multi-coil extension (ew_kernels.hip): coil maps times image / conjugate coil combination, `cnt` coils of a chunk at a time
int ew_launch_coil_mul(qmri_ctx* ctx, size_t n, size_t plane, int cnt, const double2* x, const double2* maps, double2* out);
int ew_launch_coil_sum(qmri_ctx* ctx, size_t n, size_t plane, int cnt, const double2* xj, const double2* maps, double2* x, int accumulate);

// conv engine (conv_kernels.hip)
int conv_launch(qmri_ctx* ctx, ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                const PTensor* add2, int relu_out);
int conv_cin_pad(ConvKind kind, int Cin);
// bf16 x 6 path of the 3x3 layers (conv6_kernels.hip)
enum { PROF_CONV3 = 0, PROF_CONV2 = 1, PROF_LSQR = 2, PROF_TV = 3 };
int qmri_prof_pair(qmri_ctx* ctx, hipEvent_t* start, hipEvent_t* stop, int kind = PROF_CONV3, double flop = 0.0);   // profile level 2: next event pair (else nullptrs)
int qmri_prof_chain_finish(qmri_ctx* ctx, long count = -1);   // synchronises, adds the pairs' durations to the profile; count >= 0: only the first `count` pairs
bool conv6_enabled();
int conv6_act_begin(qmri_ctx* ctx, int nlayers);           // f16 scheme: start / finish the per-layer |output| report of a forward pass
int conv6_act_end(qmri_ctx* ctx);
int conv6_default_sp();                                    // 2 = f16 x 3 products, 3 = bf16 x 6 products (knob conv_scheme = 3)
bool conv6_weights_fit_f16(const float* w, size_t n);
void conv6_plan_pack(ConvLayer& L, const float* w, std::vector<uint16_t>& packed);
int conv6_pack_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w, float wmax);       // the same on the device from the flat blob there (allocates L.wp6)
int conv6s_pack_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w, float wmax);
int conv_pack_weights_dev(qmri_ctx* ctx, ConvLayer& L, const float* d_w);           // ... the f32 A-fragments of the fallback kernels (allocates L.wp)
int conv6_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out, const PTensor* add1,
                 const PTensor* add2, int relu_out);
// a run of 3x3 layers of the full-resolution level as ONE launch with LDS-resident tiles (conv6_kernels.hip k_conv6r): nres = 2 nb ResBlock layers
// 64 -> 64 on src -> cur (+ skip at the last one), optionally with the network's head in front (head_in -> src)
struct Conv6rRun {
    const ConvLayer* head = nullptr; const PTensor* head_in = nullptr;
    const ConvLayer* res = nullptr; int nres = 0;
    const PTensor* src = nullptr; const PTensor* cur = nullptr; const PTensor* skip = nullptr;
    const ConvLayer* tail = nullptr; const PTensor* tail_out = nullptr;   // optional last layer 64 -> out_nc writing the planar network output
    const ConvLayer* down = nullptr; const PTensor* down_out = nullptr;   // optional last layer: the level's 2x2 / stride-2 convolution 64 -> 128 into the next level's tensor (cur is then not written)
};
int conv6r_try(qmri_ctx* ctx, const Conv6rRun& run, int B, bool* done);   // *done = false: not eligible, nothing launched
size_t conv6r_xbuf_bytes(int tiles);
void conv6s_plan_pack(ConvLayer& L, const float* w, std::vector<uint16_t>& packed);   // 2x2 / stride-2 layers
bool conv6s_usable(const ConvLayer& L, const PTensor& in, const PTensor& out);
int conv6s_launch(qmri_ctx* ctx, const ConvLayer& L, int B, const PTensor& in, const PTensor& out);
size_t conv_pack_weights(const ConvLayer& L, const float* w_src, std::vector<float>& packed);
void conv_plan_layer(ConvLayer& L, ConvKind kind, int Cin, int Cout);

// dictionary match (dict_kernels.hip)
int dict_launch(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float2* d_xfit);
int dict_scratch(qmri_ctx* ctx, void** buf, size_t* cap, size_t need_bytes);
int dict_launch_merge(qmri_ctx* ctx, const float4* part, int P, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win);
// wide dictionaries (16 < s <= 1024; dictw_kernels.hip)
int dictw_pack_dictionary(qmri_ctx* ctx, const float* D_host, int K, int s);    // fills ctx->dict.d_pack / G8 / ntiles
int dictw_launch(qmri_ctx* ctx, const double2* d_X, int Npix, float* d_qmap, float* d_pd, float* d_mt, int32_t* d_dm, float4* win);

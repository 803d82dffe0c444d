/*
 * orc_net.c -- the denoiser plugin.  Test infrastructure only.
 *
 * Architecture restated from PyTorch_Denoiser/zhang_dpir_testing_code/network_unet.py:68-117 (UNetRes) and
 * basicblock.py:61-98 (conv), :211-223 (ResBlock: x + conv(relu(conv(x)))), :413-419 (convtranspose 2x2 s2),
 * :437-443 (strideconv 2x2 s2); instantiation main_test.py:245-252 (nc=[64,128,256,512], nb=4, bias-free).
 * Wrapper semantics restated from main_files/utils/denoiseImage_PnP_ADMM.m:72-77 (double->single cast),
 * :88 (output of the last conv), :99-104 (residual_noise switch), :111-115 (cast back).
 *
 * PINNED: tests/test_oracle_net.py checks this file against golden vectors produced by the reference's own
 * UNetRes module (tools/gen_golden.py).
 *
 * Tensor layout: [B][C][W][H] fp32 with h (MATLAB row) fastest -- the memory order of a MATLAB H x W x C array.
 * Weight tensors keep PyTorch layout: Conv2d OIHW (kh pairs with h), ConvTranspose2d IOHW.
 */
#include "orc_internal.h"

struct orc_net {
    int arch, in_nc, out_nc, nc[4], nb;
    float* w;
    size_t nw;
};

size_t orc_net_nparams(int arch, int in_nc, int out_nc, const int* nc, int nb) {
    size_t n = 0;
    if (arch == 0) {
        n += (size_t)nc[0] * in_nc * 9;
        for (int l = 0; l < 3; ++l) {
            n += (size_t)2 * nb * nc[l] * nc[l] * 9;
            n += (size_t)nc[l + 1] * nc[l] * 4;
        }
        n += (size_t)2 * nb * nc[3] * nc[3] * 9;
        for (int l = 3; l > 0; --l) {
            n += (size_t)nc[l] * nc[l - 1] * 4;
            n += (size_t)2 * nb * nc[l - 1] * nc[l - 1] * 9;
        }
        n += (size_t)out_nc * nc[0] * 9;
    } else {
        int width = nc[0];
        if (nb == 1) return (size_t)out_nc * in_nc * 9;
        n += (size_t)width * in_nc * 9;
        n += (size_t)(nb - 2) * width * width * 9;
        n += (size_t)out_nc * width * 9;
    }
    return n;
}

orc_net* orc_net_create(int arch, int in_nc, int out_nc, const int* nc4, int nb, const float* weights,
                        size_t nfloats) {
    if (nfloats != orc_net_nparams(arch, in_nc, out_nc, nc4, nb)) return NULL;
    orc_net* net = (orc_net*)orc_xmalloc(sizeof(orc_net));
    net->arch = arch; net->in_nc = in_nc; net->out_nc = out_nc; net->nb = nb;
    for (int i = 0; i < 4; ++i) net->nc[i] = nc4[i];
    net->nw = nfloats;
    net->w = (float*)orc_xmalloc(sizeof(float) * nfloats);
    memcpy(net->w, weights, sizeof(float) * nfloats);
    return net;
}

void orc_net_destroy(orc_net* net) {
    if (!net) return;
    free(net->w); free(net);
}

/* out[o][w][h] (+)= sum_c sum_kh sum_kw Wt[o][c][kh][kw] * act(in[c][w+kw-1][h+kh-1]),  zero padding 1.
 * relu_in: apply ReLU to the input while reading; add: optional tensor added to the result (residual). */
__attribute__((optimize("fp-contract=fast")))
static void conv3x3(const float* in, int Cin, int H, int W, const float* wt, int Cout, int relu_in,
                    const float* add, float* out) {
    const int Hp = H + 2, Wp = W + 2;
    const size_t pplane = (size_t)Hp * Wp, plane = (size_t)H * W;
    float* pad = (float*)calloc(pplane * Cin + 64, sizeof(float));
    if (!pad) abort();
#pragma omp parallel for schedule(static)
    for (int c = 0; c < Cin; ++c)
        for (int w = 0; w < W; ++w) {
            const float* src = in + c * plane + (size_t)w * H;
            float* dst = pad + c * pplane + (size_t)(w + 1) * Hp + 1;
            if (relu_in) for (int h = 0; h < H; ++h) dst[h] = src[h] > 0.f ? src[h] : 0.f;
            else memcpy(dst, src, sizeof(float) * H);
        }
    /* register-blocked direct convolution: OB output channels x HB rows per block, fp32 FMA chain over
       (c, kw, kh).  Weights are repacked per output block so the OB values of one tap are contiguous. */
    enum { OB = 4, HB = 24 };
    typedef float v8 __attribute__((vector_size(32)));
    typedef float v8u __attribute__((vector_size(32), aligned(4)));
    const int nob = (Cout + OB - 1) / OB;
    float* wpk = (float*)calloc((size_t)nob * Cin * 9 * OB, sizeof(float));
    if (!wpk) abort();
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw)
                    wpk[(((size_t)(o / OB) * Cin + c) * 9 + kw * 3 + kh) * OB + (o % OB)] =
                        wt[(((size_t)o * Cin + c) * 3 + kh) * 3 + kw];
#pragma omp parallel for collapse(2) schedule(static)
    for (int ob = 0; ob < nob; ++ob)
        for (int w = 0; w < W; ++w) {
            const int o0 = ob * OB;
            const int no = (Cout - o0 < OB) ? Cout - o0 : OB;
            const float* wb = wpk + (size_t)ob * Cin * 9 * OB;
            for (int hh = 0; hh < H; hh += HB) {
                /* a short last block is recomputed as a full block ending at H (stores are idempotent) */
                const int h0 = (H - hh < HB && H >= HB) ? H - HB : hh;
                const int nh = (H - h0 < HB) ? H - h0 : HB;
                float accs[OB][HB];
                if (nh == HB) {
                    v8 a00 = {0}, a01 = {0}, a02 = {0}, a10 = {0}, a11 = {0}, a12 = {0};
                    v8 a20 = {0}, a21 = {0}, a22 = {0}, a30 = {0}, a31 = {0}, a32 = {0};
                    for (int c = 0; c < Cin; ++c) {
                        const float* wc = wb + (size_t)c * 9 * OB;
                        for (int kw = 0; kw < 3; ++kw) {
                            const float* row = pad + c * pplane + (size_t)(w + kw) * Hp + h0;
                            for (int kh = 0; kh < 3; ++kh) {
                                const float* src = row + kh;
                                const v8 s0 = *(const v8u*)(src), s1 = *(const v8u*)(src + 8), s2 = *(const v8u*)(src + 16);
                                const float* wq = wc + (kw * 3 + kh) * OB;
                                v8 b;
                                b = (v8){wq[0], wq[0], wq[0], wq[0], wq[0], wq[0], wq[0], wq[0]};
                                a00 += b * s0; a01 += b * s1; a02 += b * s2;
                                b = (v8){wq[1], wq[1], wq[1], wq[1], wq[1], wq[1], wq[1], wq[1]};
                                a10 += b * s0; a11 += b * s1; a12 += b * s2;
                                b = (v8){wq[2], wq[2], wq[2], wq[2], wq[2], wq[2], wq[2], wq[2]};
                                a20 += b * s0; a21 += b * s1; a22 += b * s2;
                                b = (v8){wq[3], wq[3], wq[3], wq[3], wq[3], wq[3], wq[3], wq[3]};
                                a30 += b * s0; a31 += b * s1; a32 += b * s2;
                            }
                        }
                    }
                    *(v8u*)&accs[0][0] = a00; *(v8u*)&accs[0][8] = a01; *(v8u*)&accs[0][16] = a02;
                    *(v8u*)&accs[1][0] = a10; *(v8u*)&accs[1][8] = a11; *(v8u*)&accs[1][16] = a12;
                    *(v8u*)&accs[2][0] = a20; *(v8u*)&accs[2][8] = a21; *(v8u*)&accs[2][16] = a22;
                    *(v8u*)&accs[3][0] = a30; *(v8u*)&accs[3][8] = a31; *(v8u*)&accs[3][16] = a32;
                } else {
                    for (int a = 0; a < OB; ++a)
                        for (int i = 0; i < HB; ++i) accs[a][i] = 0.f;
                    for (int c = 0; c < Cin; ++c)
                        for (int kw = 0; kw < 3; ++kw) {
                            const float* row = pad + c * pplane + (size_t)(w + kw) * Hp + h0;
                            for (int kh = 0; kh < 3; ++kh) {
                                const float* src = row + kh;
                                const float* wq = wb + ((size_t)c * 9 + kw * 3 + kh) * OB;
                                for (int a = 0; a < OB; ++a)
                                    for (int i = 0; i < nh; ++i) accs[a][i] += wq[a] * src[i];
                            }
                        }
                }
                for (int a = 0; a < no; ++a) {
                    float* dst = out + (size_t)(o0 + a) * plane + (size_t)w * H + h0;
                    if (add) {
                        const float* ad = add + (size_t)(o0 + a) * plane + (size_t)w * H + h0;
                        for (int i = 0; i < nh; ++i) dst[i] = ad[i] + accs[a][i];
                    } else {
                        for (int i = 0; i < nh; ++i) dst[i] = accs[a][i];
                    }
                }
            }
        }
    free(wpk);
    free(pad);
}

/* strideconv: out[o][j][i] = sum_c sum_kh sum_kw Wt[o][c][kh][kw] * in[c][2j+kw][2i+kh]   (basicblock.py:437-443) */
static void conv2x2s2(const float* in, int Cin, int H, int W, const float* wt, int Cout, float* out) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t plane = (size_t)H * W, oplane = (size_t)Ho * Wo;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
        float* dst = out + o * oplane;
        for (size_t i = 0; i < oplane; ++i) dst[i] = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float* wv = wt + ((size_t)o * Cin + c) * 4;   /* [kh][kw] */
            const float* src = in + c * plane;
            for (int j = 0; j < Wo; ++j)
                for (int i = 0; i < Ho; ++i) {
                    const float* p0 = src + (size_t)(2 * j) * H + 2 * i;       /* kw = 0 */
                    const float* p1 = p0 + H;                                  /* kw = 1 */
                    dst[(size_t)j * Ho + i] += wv[0] * p0[0] + wv[2] * p0[1] + wv[1] * p1[0] + wv[3] * p1[1];
                }
        }
    }
}

/* convtranspose: out[o][2j+kw][2i+kh] = sum_c in[c][j][i] * Wt[c][o][kh][kw]   (basicblock.py:413-419) */
static void convT2x2s2(const float* in, int Cin, int H, int W, const float* wt, int Cout, float* out) {
    const int Ho = H * 2;
    const size_t plane = (size_t)H * W, oplane = plane * 4;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < Cout; ++o) {
        float* dst = out + o * oplane;
        for (size_t i = 0; i < oplane; ++i) dst[i] = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float* wv = wt + ((size_t)c * Cout + o) * 4;   /* [kh][kw] */
            const float* src = in + c * plane;
            for (int j = 0; j < W; ++j)
                for (int i = 0; i < H; ++i) {
                    const float v = src[(size_t)j * H + i];
                    float* q0 = dst + (size_t)(2 * j) * Ho + 2 * i;            /* kw = 0 */
                    float* q1 = q0 + Ho;                                       /* kw = 1 */
                    q0[0] += v * wv[0]; q0[1] += v * wv[2];
                    q1[0] += v * wv[1]; q1[1] += v * wv[3];
                }
        }
    }
}

static void add_inplace(float* a, const float* b, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) a[i] += b[i];
}

/* nb ResBlocks on C channels, in place: a <- a + conv(relu(conv(a))).  Returns advanced weight pointer. */
static const float* resblocks(float* a, int C, int H, int W, int nb, const float* w, float* t1, float* t2) {
    const size_t wsz = (size_t)C * C * 9, n = (size_t)C * H * W;
    for (int b = 0; b < nb; ++b) {
        conv3x3(a, C, H, W, w, C, 0, NULL, t1); w += wsz;
        conv3x3(t1, C, H, W, w, C, 1, a, t2); w += wsz;
        memcpy(a, t2, sizeof(float) * n);
    }
    return w;
}

static void unetres_forward_one(const orc_net* net, const float* x0, int H, int W, float* out) {
    const int* nc = net->nc;
    const int nb = net->nb;
    const float* w = net->w;
    const size_t p1 = (size_t)H * W, p2 = p1 / 4, p3 = p2 / 4, p4 = p3 / 4;
    const size_t big = (size_t)nc[0] * p1;
    float* x1 = (float*)orc_xmalloc(sizeof(float) * nc[0] * p1);
    float* x2 = (float*)orc_xmalloc(sizeof(float) * nc[1] * p2);
    float* x3 = (float*)orc_xmalloc(sizeof(float) * nc[2] * p3);
    float* x4 = (float*)orc_xmalloc(sizeof(float) * nc[3] * p4);
    float* a = (float*)orc_xmalloc(sizeof(float) * big);
    float* t1 = (float*)orc_xmalloc(sizeof(float) * big);
    float* t2 = (float*)orc_xmalloc(sizeof(float) * big);
    /* x1 = m_head(x0)   network_unet.py:107 */
    conv3x3(x0, net->in_nc, H, W, w, nc[0], 0, NULL, x1); w += (size_t)nc[0] * net->in_nc * 9;
    /* x2 = m_down1(x1)  :108 */
    memcpy(a, x1, sizeof(float) * nc[0] * p1);
    w = resblocks(a, nc[0], H, W, nb, w, t1, t2);
    conv2x2s2(a, nc[0], H, W, w, nc[1], x2); w += (size_t)nc[1] * nc[0] * 4;
    /* x3 = m_down2(x2)  :109 */
    memcpy(a, x2, sizeof(float) * nc[1] * p2);
    w = resblocks(a, nc[1], H / 2, W / 2, nb, w, t1, t2);
    conv2x2s2(a, nc[1], H / 2, W / 2, w, nc[2], x3); w += (size_t)nc[2] * nc[1] * 4;
    /* x4 = m_down3(x3)  :110 */
    memcpy(a, x3, sizeof(float) * nc[2] * p3);
    w = resblocks(a, nc[2], H / 4, W / 4, nb, w, t1, t2);
    conv2x2s2(a, nc[2], H / 4, W / 4, w, nc[3], x4); w += (size_t)nc[3] * nc[2] * 4;
    /* x = m_body(x4)    :111 */
    memcpy(a, x4, sizeof(float) * nc[3] * p4);
    w = resblocks(a, nc[3], H / 8, W / 8, nb, w, t1, t2);
    /* x = m_up3(x + x4) :112 */
    add_inplace(a, x4, (size_t)nc[3] * p4);
    convT2x2s2(a, nc[3], H / 8, W / 8, w, nc[2], t1); w += (size_t)nc[3] * nc[2] * 4;
    memcpy(a, t1, sizeof(float) * nc[2] * p3);
    w = resblocks(a, nc[2], H / 4, W / 4, nb, w, t1, t2);
    /* x = m_up2(x + x3) :113 */
    add_inplace(a, x3, (size_t)nc[2] * p3);
    convT2x2s2(a, nc[2], H / 4, W / 4, w, nc[1], t1); w += (size_t)nc[2] * nc[1] * 4;
    memcpy(a, t1, sizeof(float) * nc[1] * p2);
    w = resblocks(a, nc[1], H / 2, W / 2, nb, w, t1, t2);
    /* x = m_up1(x + x2) :114 */
    add_inplace(a, x2, (size_t)nc[1] * p2);
    convT2x2s2(a, nc[1], H / 2, W / 2, w, nc[0], t1); w += (size_t)nc[1] * nc[0] * 4;
    memcpy(a, t1, sizeof(float) * nc[0] * p1);
    w = resblocks(a, nc[0], H, W, nb, w, t1, t2);
    /* x = m_tail(x + x1) :115 */
    add_inplace(a, x1, (size_t)nc[0] * p1);
    conv3x3(a, nc[0], H, W, w, net->out_nc, 0, NULL, out);
    free(x1); free(x2); free(x3); free(x4); free(a); free(t1); free(t2);
}

/* arch 1: conv3x3 -> ReLU -> ... -> conv3x3 (no ReLU after the last); DnCNN-style, parity unpinned. */
static void seqconv_forward_one(const orc_net* net, const float* x0, int H, int W, float* out) {
    const int width = net->nc[0], nb = net->nb;
    const size_t p = (size_t)H * W;
    const float* w = net->w;
    if (nb == 1) { conv3x3(x0, net->in_nc, H, W, w, net->out_nc, 0, NULL, out); return; }
    float* a = (float*)orc_xmalloc(sizeof(float) * width * p);
    float* b = (float*)orc_xmalloc(sizeof(float) * width * p);
    conv3x3(x0, net->in_nc, H, W, w, width, 0, NULL, a); w += (size_t)width * net->in_nc * 9;
    for (int l = 1; l < nb - 1; ++l) {
        conv3x3(a, width, H, W, w, width, 1, NULL, b); w += (size_t)width * width * 9;
        float* t = a; a = b; b = t;
    }
    conv3x3(a, width, H, W, w, net->out_nc, 1, NULL, out);
    free(a); free(b);
}

void orc_net_forward(const orc_net* net, const float* in, int H, int W, int B, float* out) {
    const size_t p = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
        if (net->arch == 0) unetres_forward_one(net, in + (size_t)b * net->in_nc * p, H, W, out + (size_t)b * net->out_nc * p);
        else seqconv_forward_one(net, in + (size_t)b * net->in_nc * p, H, W, out + (size_t)b * net->out_nc * p);
    }
}

void orc_denoise(const orc_net* net, const double* in, int H, int W, int C, int B, int residual_noise,
                 double* out) {
    const size_t p = (size_t)H * W;
    float* fin = (float*)orc_xmalloc(sizeof(float) * p * C * B);
    float* fout = (float*)orc_xmalloc(sizeof(float) * p * net->out_nc * B);
    for (size_t i = 0; i < p * C * B; ++i) fin[i] = (float)in[i];          /* im2single of a double: cast  :72-77 */
    orc_net_forward(net, fin, H, W, B, fout);                              /* activations(...) :88 */
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < net->out_nc; ++c)
            for (size_t i = 0; i < p; ++i) {
                float res = fout[((size_t)b * net->out_nc + c) * p + i];
                float I = residual_noise ? fin[((size_t)b * C + c) * p + i] - res : res;   /* :99-104 */
                out[((size_t)b * net->out_nc + c) * p + i] = (double)I;                   /* :111-115 */
            }
    free(fin); free(fout);
}

#!/usr/bin/env python3
"""can Winograd F(2x2, 3x3) carry the denoiser at the parity tolerance?

The 64-layer UNetRes (network_unet.py:68-117 restated with torch.nn.functional on the CPU) on synth.random_weights(seed=1, gain=0.7) -- the weights under
which every level matters -- and the 224 x 224 golden input, with every 3x3 layer computed four ways:
  f64 direct       the reference value (float64 conv2d)
  f32 direct       what the reference's arithmetic gives (float32 conv2d; the oracle's orc_net.c is this up to summation order)
  f32 winograd     F(2x2, 3x3): U = G g G^T (weights, float32), V = B^T d B (float32), M = sum_c U V per position (float32 GEMM), Y = A^T M A (float32)
  f16x3 winograd   the same with the GEMM as the matrix-core path would run it: U and V split into f16 pieces (hi, 2^11 (x - hi)), three products
                   hi hi + 2^-11 (hi lo + lo hi), fp32 accumulation -- the arithmetic of conv6_kernels.hip applied to the transformed operands
                   (weights scaled per layer by the power of two that puts max |U| into [1, 2), as conv6_weight_scale does)
Every variant runs the WHOLE network in its own arithmetic (errors propagate through all 64 layers as they would on the GPU).
Output: relative L2 of each against f64 direct and of the winograd variants against f32 direct, one JSON line.  Stop rule of the verdict:
rel-L2 of f32 winograd vs the fp32 reference above 1e-5 -> not worth a kernel; the parity tolerance of tests/test_gpu_net.py is 2e-5."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import synth  # noqa: E402

IN_NC, OUT_NC, NC, NB = 10, 10, (64, 128, 256, 512), 4
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split(blob, dtype):
    ws, off = [], 0
    for name, shp in synth.unetres_weight_shapes(IN_NC, OUT_NC, NC, NB):
        n = int(np.prod(shp))
        ws.append(torch.from_numpy(blob[off:off + n].reshape(shp).copy()).to(dtype))
        off += n
    return ws


def f16_pieces(x):
    """hi = f16(x), lo' = f16((x - hi) * 2^11), as split_pair_h (conv6_device.h); returned as float32 tensors holding f16 values."""
    hi = x.to(torch.float16).to(torch.float32)
    lo = ((x - hi) * 2048.0).to(torch.float16).to(torch.float32)
    return hi, lo


def winograd_conv3(x, w, mode):
    """x [1, C, H, W] float32, w [Co, C, 3, 3] float32 -> [1, Co, H, W]; H, W even; zero padding 1."""
    C, H, W = x.shape[1], x.shape[2], x.shape[3]
    Co = w.shape[0]
    g, bt, at = G.float(), BT.float(), AT.float()
    U = torch.einsum("ij,ocjk,lk->iloc", g, w, g).reshape(16, Co, C)                  # [pos][Co][C]
    xp = F.pad(x[0], (1, 1, 1, 1))                                                     # [C][H+2][W+2]
    d = xp.unfold(1, 4, 2).unfold(2, 4, 2)                                             # [C][H/2][W/2][4][4]
    V = torch.einsum("ij,cabjk,lk->ilcab", bt, d, bt).reshape(16, C, (H // 2) * (W // 2))   # [pos][C][tiles]  (adds / subtracts only)
    if mode == "f32":
        M = torch.bmm(U, V)
    else:                                                                              # f16 x 3 products, per-layer power-of-two weight scale
        mx = float(U.abs().max())
        k = 0 if mx == 0 else 1 - int(np.floor(np.log2(mx))) - 1                      # max |U| 2^k in [1, 2)
        sc = float(2.0 ** k)
        uh, ul = f16_pieces(U * sc)
        vh, vl = f16_pieces(V)
        M = (torch.bmm(uh, vh) + (torch.bmm(uh, vl) + torch.bmm(ul, vh)) * (1.0 / 2048.0)) * (1.0 / sc)
    M = M.reshape(4, 4, Co, H // 2, W // 2)
    Y = torch.einsum("pi,ijoab,qj->oapbq", at, M, at).reshape(Co, H, W)               # 2 x 2 outputs per tile, interleaved back
    return Y[None]


def forward(x, w, conv3):
    it = iter(w)

    def c3(t):
        return conv3(t, next(it))

    def resblocks(t):
        for _ in range(NB):
            t = t + c3(F.relu(c3(t)))
        return t

    x1 = c3(x)
    x2 = F.conv2d(resblocks(x1), next(it), stride=2)
    x3 = F.conv2d(resblocks(x2), next(it), stride=2)
    x4 = F.conv2d(resblocks(x3), next(it), stride=2)
    t = resblocks(x4)
    t = resblocks(F.conv_transpose2d(t + x4, next(it), stride=2))
    t = resblocks(F.conv_transpose2d(t + x3, next(it), stride=2))
    t = resblocks(F.conv_transpose2d(t + x2, next(it), stride=2))
    return c3(t + x1)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    blob = synth.random_weights(in_nc=IN_NC, out_nc=OUT_NC, nc=NC, nb=NB, seed=1, gain=0.7)
    x = torch.from_numpy(synth.golden224_input(10)[None].copy())
    direct = lambda t, w: F.conv2d(t, w, padding=1)                                    # noqa: E731
    with torch.no_grad():
        y64 = forward(x.double(), split(blob, torch.float64), direct)
        y32 = forward(x, split(blob, torch.float32), direct)
        yw32 = forward(x, split(blob, torch.float32), lambda t, w: winograd_conv3(t, w, "f32"))
        yw16 = forward(x, split(blob, torch.float32), lambda t, w: winograd_conv3(t, w, "f16x3"))
        # one layer alone (the 224 x 224 x 64 shape), same input for every variant: the error a single layer adds
        t = torch.rand(1, 64, 224, 224)
        w1 = split(blob, torch.float32)[1]
        l64 = F.conv2d(t.double(), w1.double(), padding=1)
        layer = {"f32_direct": rel(F.conv2d(t, w1, padding=1), l64), "f32_winograd": rel(winograd_conv3(t, w1, "f32"), l64),
                 "f16x3_winograd": rel(winograd_conv3(t, w1, "f16x3"), l64)}
    out = {"what": "UNetRes 10 -> 10 at 224 x 224, random_weights(seed=1, gain=0.7), golden224_input; every 3x3 layer of the whole network in the named arithmetic",
           "rel_l2_vs_f64_direct": {"f32_direct": rel(y32, y64), "f32_winograd": rel(yw32, y64), "f16x3_winograd": rel(yw16, y64)},
           "rel_l2_vs_f32_direct": {"f32_winograd": rel(yw32, y32), "f16x3_winograd": rel(yw16, y32)},
           "one_224x224x64_layer_rel_l2_vs_f64": layer,
           "stop_rule": "f32 winograd vs the fp32 reference > 1e-5 -> record and stop (VERDICT r04 item 3); parity tolerance of the GPU tests 2e-5",
           "max_abs_output": float(y64.abs().max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A labelled vendor reference point for the denoiser (SURVEY.md section 7 allows vendor libraries in the harness, never on the product
path): the 64-layer UNetRes forward -- same shapes, same topology (network_unet.py:68-117 restated: head, 3 x [4 ResBlocks + 2x2 s2 conv],
4 ResBlocks, 3 x [2x2 s2 transposed conv + 4 ResBlocks], tail, additive skips, no bias) -- as a chain of torch.nn.functional.conv2d /
conv_transpose2d calls (MIOpen / hipBLASLt behind them) on this box, next to libqmri's forward on the same random weights and input.

  fp32       the arithmetic the reference runs (float32or64_mode = '32'); the comparison that counts
  tf32-like  not available on this stack; f16 is shown as CONTEXT only (it is not fp32-accurate: ~1e-3 relative)
One JSON line.  The reference publishes no timings: this is the only same-node bar there is."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

N, IN_NC, OUT_NC, NC, NB = 224, 10, 10, (64, 128, 256, 512), 4
DENOISER_FLOP = 213_253_619_712


def split(blob):
    ws, off = [], 0
    for name, shp in synth.unetres_weight_shapes(IN_NC, OUT_NC, NC, NB):
        n = int(np.prod(shp))
        ws.append(torch.from_numpy(blob[off:off + n].reshape(shp).copy()))
        off += n
    assert off == blob.size
    return ws


def forward(x, w):
    """x [B, in_nc, H, W]; w: the 64 tensors in state-dict order (Conv2d OIHW, ConvTranspose2d IOHW)."""
    it = iter(w)

    def conv3(t):
        return F.conv2d(t, next(it), padding=1)

    def resblocks(t):
        for _ in range(NB):
            t = t + conv3(F.relu(conv3(t)))
        return t

    x1 = conv3(x)
    x2 = F.conv2d(resblocks(x1), next(it), stride=2)
    x3 = F.conv2d(resblocks(x2), next(it), stride=2)
    x4 = F.conv2d(resblocks(x3), next(it), stride=2)
    t = resblocks(x4)
    t = resblocks(F.conv_transpose2d(t + x4, next(it), stride=2))
    t = resblocks(F.conv_transpose2d(t + x3, next(it), stride=2))
    t = resblocks(F.conv_transpose2d(t + x2, next(it), stride=2))
    return conv3(t + x1)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    torch.cuda.init()
    torch.backends.cudnn.benchmark = True                      # let MIOpen pick its best kernels (find runs in the untimed warm-up)
    blob = synth.random_weights(in_nc=IN_NC, out_nc=OUT_NC, nc=NC, nb=NB, seed=1, gain=0.7)
    w32 = [t.cuda() for t in split(blob)]
    w16 = [t.half() for t in w32]
    out = {"what": "UNetRes (DRUNet) forward 10 -> 10 channels at 224 x 224, 213.25 GFLOP per slice, random weights (seed 1, gain 0.7)",
           "device": torch.cuda.get_device_name(0), "torch": torch.__version__}
    eng = E.Engine(0)
    eng.set_denoiser(blob, N, N, max_batch=15)
    rng = np.random.default_rng(0)
    for B in (1, 15):
        x = rng.random((B, IN_NC, N, N), dtype=np.float32)
        xt = torch.from_numpy(x).cuda()
        with torch.no_grad():
            ms32 = timed(lambda: forward(xt, w32), 20 if B == 1 else 5)
            y32 = forward(xt, w32).float().cpu().numpy()
            xh = xt.half()
            ms16 = timed(lambda: forward(xh, w16), 20 if B == 1 else 5)
            y16 = forward(xh, w16).float().cpu().numpy()
        # libqmri on the same weights and input: the raw device entry point ([B][C][W][H] tensors: transpose once, outside the timing)
        d_in = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 1, 3, 2))).cuda()
        d_out = torch.empty_like(d_in)
        torch.cuda.synchronize()
        import ctypes as C

        def qf():
            eng._check(eng.L.qmri_net_forward_dev(eng.h, C.c_void_p(d_in.data_ptr()), B, C.c_void_p(d_out.data_ptr())))
        for _ in range(3):
            qf()
        eng.synchronize()
        reps = 20 if B == 1 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            qf()
        eng.synchronize()
        msq = (time.perf_counter() - t0) / reps * 1e3
        yq = d_out.cpu().numpy().transpose(0, 1, 3, 2)
        rel = lambda a, b: float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()))
        out[f"batch_{B}"] = {
            "vendor_fp32_ms_per_slice": round(ms32 / B, 4), "vendor_fp32_tflops": round(DENOISER_FLOP * B / (ms32 * 1e-3) / 1e12, 1),
            "vendor_f16_ms_per_slice_context_only": round(ms16 / B, 4),
            "libqmri_ms_per_slice": round(msq / B, 4), "libqmri_fp32_equivalent_tflops": round(DENOISER_FLOP * B / (msq * 1e-3) / 1e12, 1),
            "speedup_over_vendor_fp32": round(ms32 / msq, 2),
            "rel_l2_libqmri_vs_vendor_fp32": rel(yq, y32), "rel_l2_vendor_f16_vs_vendor_fp32": rel(y16, y32)}
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""k_conv6r: forward-pass time (device side, HIP events) against the delay before the first ring fetch and the back-off between attempts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

w = synth.random_weights(seed=1, gain=0.7)
rng = np.random.default_rng(3)
torch.cuda.init()
e = E.Engine(0)
e.set_denoiser(w, 224, 224, max_batch=1)
x = torch.from_numpy(rng.random((10, 224, 224)).astype(np.float32)).cuda()
y = torch.empty_like(x)
stream = torch.cuda.Stream()
torch.cuda.synchronize()
e.set_stream(stream.cuda_stream)


import ctypes as C
import time


def fwd_us(n=200):
    f = lambda: e._check(e.L.qmri_net_forward_dev(e.h, C.c_void_p(x.data_ptr()), 1, C.c_void_p(y.data_ptr())))
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()                                      # (synchronises the stream itself: the f16 range check)
    return (time.perf_counter() - t0) / n * 1e6


e.conv_resident(0)
print(f"one launch per layer: {fwd_us():.1f} us per forward", flush=True)
ref = y.clone()
for pre in (0, 8, 16, 24):
    for back in (0, 8):
        e.conv_resident(((pre << 8) | (back << 16)) if (pre or back) else 1)
        t = fwd_us()
        print(f"resident, pre-delay {pre:3d} x 64 clk, back-off {back:3d} x 64 clk: {t:.1f} us per forward, identical {bool(torch.equal(y, ref))}", flush=True)
print("time-outs", e.conv_resident(1))
e.close()

#!/usr/bin/env python3
"""Phase times inside the resident-tile ResBlock launch (k_conv6r, knob res_stamps): 100 MHz stamps of four workgroups (matrix wave 0 and loader
wave 0), per layer, of the LAST launch of a forward pass (the up path's run)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["QMRI_DEBUG"] = "res_stamps=" + (sys.argv[1] if len(sys.argv) > 1 else "1")          # 1: the up path's launch (ResBlocks + tail), 2: the down path's (head + ResBlocks + down-sampling convolution)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

w = synth.random_weights(seed=1, gain=0.7)
e = E.Engine(0)
e.set_denoiser(w, 224, 224, max_batch=1)
x = np.random.default_rng(3).random((224, 224, 10))
for _ in range(5):
    e.denoise(x)
out = (C.c_ulonglong * 1024)()
e.L.qmri_debug_conv_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
assert e.L.qmri_debug_conv_stamps(e.h, out, -6) == 0
s = np.array(out[:640], dtype=np.int64).reshape(4, 2, 10, 8)            # [workgroup][matrix / loader wave 0][layer < R_MAXL][stamp]; the last launch = the up path's: eight ResBlock layers + the tail
us = lambda a, b: (b - a) / 100.0 if a > 0 and b > 0 else float('nan')      # (a phase a layer does not have leaves no stamp)
print("matrix wave 0: loop | residual operand | epilogue (split, LDS writes, stores) | wait E2 | ring fetch (poll-loads, LDS writes) | wait E3   ;   loader wave 0: E1->E2 | publish (issue) | wait E3     [us]")
for wg in range(4):
    print(f"workgroup {wg * 50}")
    nlay = int((s[wg, 0, :, 0] > 0).sum())
    for l in range(nlay):
        m, ld = s[wg, 0, l], s[wg, 1, l]
        if l < nlay - 1:
            nxt = s[wg, 0, l + 1, 0]
            print(f"  layer {l}: M loop {us(m[0], m[1]):5.2f} res {us(m[1], m[2]):5.2f} epi {us(m[2], m[3]):5.2f} E2 {us(m[3], m[4]):5.2f} fetch {us(m[4], m[5]):5.2f} E3 {us(m[5], m[6]):5.2f}"
                  f" | L E2 {us(ld[0], ld[1]):5.2f} publish {us(ld[1], ld[2]):5.2f} E3 {us(ld[2], ld[6]):5.2f}"
                  f" | layer total {us(m[0], nxt):5.2f}")
        else:
            print(f"  layer {l}: M loop {us(m[0], m[1]):5.2f} res {us(m[1], m[2]):5.2f} epi {us(m[2], m[3]):5.2f}")
e.close()

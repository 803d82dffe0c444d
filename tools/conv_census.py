#!/usr/bin/env python3
"""Diagnostic: where and when do the persistent conv workgroups run?  (knob conv_stamps = 1; the stamps are those of
the LAST conv launch, so this runs a 2-layer SEQ_CONV net whose last layer is the shape of interest.)
    python tools/conv_census.py CHANNELS IMAGE_SIZE"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QMRI_DEBUG"] = "conv_stamps=1"
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

C_, HW = int(sys.argv[1]), int(sys.argv[2])
eng = E.Engine(0)
n = 2 * C_ * C_ * 9
w = ((synth.uniform01(5, n) - 0.5) * 0.1).astype(np.float32)
eng.set_denoiser(w, HW, HW, in_nc=C_, out_nc=C_, nc=(C_, 0, 0, 0), nb=2, arch=1)
x = synth.uniform01(6, HW * HW * C_).reshape(HW, HW, C_)
for _ in range(3):
    eng.denoise(x)
nwg = 512
buf = np.zeros(4096 * 11, np.uint64)
st = eng.L.qmri_debug_conv_stamps(eng.h, buf.ctypes.data_as(C.c_void_p), nwg)
assert st == 0
full = buf
b = buf[: nwg * 5].reshape(nwg, 5)
b = b[b[:, 1] > 0]
t0 = b[:, 0].min()
start = (b[:, 0] - t0) / 100.0
end = (b[:, 1] - t0) / 100.0          # us (s_memrealtime ticks at 100 MHz)
hw = b[:, 2]
xcc = b[:, 3] & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"workgroups {len(b)}; tiles/WG min {b[:, 4].min()} max {b[:, 4].max()} sum {b[:, 4].sum()}")
print(f"start spread {start.max():.1f} us; end min {end.min():.1f} max {end.max():.1f} us; mean duration {np.mean(end - start):.1f}")
u, cnt = np.unique(key, return_counts=True)
print(f"distinct CUs {len(u)}; WGs-per-CU histogram {np.bincount(cnt)}")
print("xcc histogram", np.bincount(xcc.astype(int)))
order = np.argsort(end)
print("last 5 finishing (end us, tiles): ", [(round(float(end[i]), 1), int(b[i, 4])) for i in order[-5:]])
print("first 5 finishing (end us, tiles):", [(round(float(end[i]), 1), int(b[i, 4])) for i in order[:5]])


#!/usr/bin/env python3
"""Probe of the resident-tile ResBlock launch (k_conv6r): same bits as one launch per layer?  how long is a forward pass either way?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

w = synth.random_weights(seed=1, gain=0.7)
rng = np.random.default_rng(3)
e = E.Engine(0)
e.set_denoiser(w, 224, 224, max_batch=1)
x = rng.random((224, 224, 10))
outs = {}
for on in (0, 1, 0, 1):
    e.conv_resident(on)
    y = e.denoise(x)
    t0 = time.perf_counter()
    for _ in range(20):
        y = e.denoise(x)
    dt = (time.perf_counter() - t0) / 20
    print(f"resident {on}: denoise (host buffers) {dt * 1e3:.3f} ms, scheme {e.denoiser_scheme()}, time-outs {e.conv_resident(on)}", flush=True)
    outs.setdefault(on, y)
a, b = outs[0], outs[1]
print("finite", bool(np.all(np.isfinite(a)) and np.all(np.isfinite(b))), "identical", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()),
      "max |y|", float(np.abs(a).max()))
e.close()

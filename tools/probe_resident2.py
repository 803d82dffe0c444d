#!/usr/bin/env python3
"""k_conv6r against one launch per layer, several forward passes: is a difference deterministic (logic) or does it vary (race)?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

w = synth.random_weights(seed=1, gain=0.7)
rng = np.random.default_rng(3)
e = E.Engine(0)
e.set_denoiser(w, 224, 224, max_batch=1)
x = rng.random((224, 224, 10))
e.conv_resident(0)
ref = e.denoise(x)
for mode, name in ((1, "plain"), (16, "producer waits for its stores"), (32, "consumer reads twice"), (64, "producer waits and sleeps 2 x 127")):
    e.conv_resident(mode)
    for k in range(4):
        y = e.denoise(x)
        d = np.abs(y - ref)
        bad = np.argwhere(d.max(axis=2) > 0)
        print(f"{name}: pass {k}: max |diff| {d.max():.4g}, differing pixels {len(bad)}", flush=True)
print("time-outs", e.conv_resident(1))
e.close()

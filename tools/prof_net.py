#!/usr/bin/env python3
"""Run the UNetRes forward a few times (for rocprofv3 --kernel-trace); prints nothing but a checksum.
Usage on the GPU box:  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pn -- python tools/prof_net.py [B] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = E.Engine(0)
eng.set_denoiser(synth.structured_weights(seed=2, eps=0.02), 224, 224, max_batch=B)
x = np.stack([synth.uniform01(9001 + b, 224 * 224 * 10).reshape(224, 224, 10) for b in range(B)], axis=3)
for _ in range(reps):
    y = eng.denoise(x)
print("checksum", float(y.sum()))
eng.close()

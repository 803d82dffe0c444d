#!/usr/bin/env python3
"""qmri_set_denoiser's calibration probe beside a second process that keeps the device busy (round 6: a bench run beside an fp16-GEMM process came up on the
bf16 scheme).  Alone and beside the other process: the probe's decision for a few set-ups with the weights packed on the host / on the device, and the
f32-MFMA fallback kernels' output (the probe's reference) compared with itself across repeated calls."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HOG = r"""
import sys, time, torch
torch.cuda.init()
a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
print('hog ready', flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
"""
CHILD = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from qmri_pnp_recon_poc_amd import engine as E, synth, _lib
L = _lib.lib()
w = synth.structured_weights(seed=2, eps=0.02)
x = np.random.default_rng(0).random((224, 224, 10))
for gpu in (0, 1):
    L.qmri_debug_knob(b'pack_gpu', gpu)
    sch = []
    for i in range(6):
        e = E.Engine(0)
        e.set_denoiser(w, 224, 224)
        sch.append(e.denoiser_scheme()[0])
        e.close()
    print('pack_gpu', gpu, 'schemes after set-up', sch, flush=True)
L.qmri_debug_knob(b'pack_gpu', 1)
for knob in (b'conv_f32', None):
    if knob: L.qmri_debug_knob(knob, 1)
    e = E.Engine(0)
    e.set_denoiser(w, 224, 224)
    ys = [e.denoise(x) for _ in range(12)]
    bad = [i for i in range(1, 12) if not np.array_equal(ys[i], ys[0])]
    print('path', 'f32-MFMA kernels' if knob else 'default', 'scheme', e.denoiser_scheme(), 'calls differing from the first', bad,
          [float(np.abs(ys[i] - ys[0]).max()) for i in bad], flush=True)
    e.close()
    if knob: L.qmri_debug_knob(knob, 0)
""" % ROOT
for label, hog in (("alone", False), ("beside fp16 GEMMs", True)):
    print("==", label, flush=True)
    h = None
    if hog:
        h = subprocess.Popen([sys.executable, "-c", HOG, "150"], stdout=subprocess.PIPE, text=True)
        h.stdout.readline()
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900)
    print(r.stdout, r.stderr[-2000:], flush=True)
    if h:
        h.terminate(); h.wait()

#!/usr/bin/env python3
"""Barrier-arrival timeline of k_conv6 (knob conv_stamps = 1) for the last conv launch of a UNetRes forward.  GPU only."""
import ctypes as C
import os
import sys

os.environ['QMRI_DEBUG'] = 'conv_stamps=1,conv_stamp_launch=' + os.environ.get('STAMP_LAUNCH', '-1')
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1                # slices per launch
eng = E.Engine(0)
eng.set_denoiser(synth.structured_weights(seed=2, eps=0.02), 224, 224, max_batch=B)
x = synth.uniform01(9001, 224 * 224 * 10).reshape(224, 224, 10)
if B > 1:
    x = np.stack([x] * B, axis=3)
for _ in range(2):
    y = eng.denoise(x)
buf = np.zeros((4096 * 11,), np.uint64)
eng.L.qmri_debug_conv_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert eng.L.qmri_debug_conv_stamps(eng.h, buf.ctypes.data, 0) == 0
s = buf[:8 * 4 * 128].reshape(8, 4, 128).astype(np.int64)
tmin = min(int(s[wg, 0][0]) for wg in range(8) if s[wg, 0][0] > 0)
for wg in range(8):
    m = s[wg, 0]; n = int((m > 0).sum())
    print('WG %3d: first stamp +%.2f us, loop end +%.2f, epilogue end +%.2f' % (wg * 13, (m[0] - tmin) / 100.0, (m[13] - tmin) / 100.0, (m[14] - tmin) / 100.0))
for wg in (0,):
    m, l, w0, w1 = s[wg, 0], s[wg, 1], s[wg, 2], s[wg, 3]
    t0 = min(m[0], l[0])
    print('WG', wg * 13)
    print('  barrier  mfma_arrive  load_arrive | loader: issued-loads  data-arrived  (us)')
    for k in range(14):
        print('  %2d  %7.2f  %7.2f | %7.2f  %7.2f' % (k, (m[k] - t0) / 100.0, (l[k] - t0) / 100.0, (w0[k] - t0) / 100.0, (w1[k] - t0) / 100.0))

m = s[0, 0]
cyc = m[65:65 + 12]
print('core cycles per step (WG 0):', ' '.join(str(int(cyc[i + 1] - cyc[i])) for i in range(11)))
print('wall us per step          :', ' '.join('%.2f' % ((m[i + 2] - m[i + 1]) / 100.0) for i in range(11)))
seq = buf[8192:8192 + 512].reshape(128, 4).astype(np.int64)
order = np.argsort(seq[:, 0])
prev = None
print('k_conv6 launches (WG 0): cfg steps | prologue+loop  epilogue  total | gap to previous launch   (us)')
for i in order:
    a, l, e, info = seq[i]
    if a == 0: continue
    print('  cfg %d steps %3d | %6.2f %6.2f %6.2f | %s' % (info // 1000, info % 1000, (l - a) / 100.0, (e - l) / 100.0, (e - a) / 100.0,
                                                       '%.2f' % ((a - prev) / 100.0) if prev else '-'))
    prev = e

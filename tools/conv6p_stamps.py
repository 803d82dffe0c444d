#!/usr/bin/env python3
"""Per-step timeline of k_conv6p (the persistent 3x3 conv kernel) from its diagnostic build: who waits for whom at the per-step
barrier, and what a loader step spends its time on.  GPU only.

    python tools/conv6p_stamps.py [B=15] [launch=60]

The diagnostic build (template parameter STAMP) is launched in place of the production kernel for ONE launch
(knob conv_stamp_launch: running number of k_conv6 / k_conv6p launches of the process; with one context, launches 0..57 are the
calibration forward of qmri_set_denoiser, 58 the head layer of the first real forward, 59 / 60 the first ResBlock's two 3x3 layers
at the 224 x 224 level).  Its run time is NOT the production kernel's: read the shares, not the length."""
import ctypes as C
import os
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 else 15
os.environ['QMRI_DEBUG'] = 'conv_stamps=1,conv_stamp_launch=' + (sys.argv[2] if len(sys.argv) > 2 else '60')
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

eng = E.Engine(0)
eng.set_denoiser(synth.structured_weights(seed=2, eps=0.02), 224, 224, max_batch=B)
x = synth.uniform01(9001, 224 * 224 * 10).reshape(224, 224, 10)
x = np.stack([x * (1.0 + 0.01 * i) for i in range(B)], axis=3)
y = eng.denoise(x)
buf = np.zeros((4096 * 11,), np.uint64)
eng.L.qmri_debug_conv_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert eng.L.qmri_debug_conv_stamps(eng.h, buf.ctypes.data, 0) == 0
s = buf[:4 * 10 * 256].reshape(4, 10, 256).astype(np.int64)
names = ['mfma arrive', 'mfma exit', 'ld issued', 'ld waited', 'ld stored', 'ld arrive', 'ld exit']
for wg in range(4):
    n = int((s[wg, 0] > 0).sum())
    if n < 24:
        print('WG sample %d: %d steps recorded' % (wg, n)); continue
    t0 = s[wg, 1, 0]
    ma, me, li, lw, ls, la, le = [s[wg, k, :n] for k in range(7)]
    step = np.diff(me) / 100.0                                  # us per step (barrier exit to barrier exit)
    print('WG sample %d: %d steps, %.2f us per step (median), %.2f us per 12-step tile' % (wg, n, np.median(step), np.median(step) * 12))
    print('   mfma waves wait at the barrier %.2f us/step (median), loaders wait %.2f us/step' % (np.median(me - ma) / 100.0, np.median(le - la) / 100.0))
    print('   loader step: issue %.2f, wait for operands %.2f, split+store %.2f, epilogue slice %.2f us (median; epilogue over the steps that have one: %.2f)'
          % (np.median(li[1:] - le[:-1]) / 100.0, np.median(lw - li) / 100.0, np.median(ls - lw) / 100.0, np.median(la - ls) / 100.0,
             np.median((la - ls)[(la - ls) > 5]) / 100.0 if ((la - ls) > 5).any() else 0.0))
    iA, iB = s[wg, 7, :n], s[wg, 8, :n]
    print('   issue phase: A requests %.2f, B requests %.2f, residual requests %.2f us (median)' % (np.median(iA[1:] - le[:-1]) / 100.0, np.median(iB - iA) / 100.0, np.median(li - iB) / 100.0))
    if wg == 0:
        print('   step: mfma_busy  mfma_wait | ld_issue ld_wait ld_store ld_epi ld_barrier_wait   (us)')
        for i in range(1, min(n, 40)):
            print('   %3d:  %6.2f    %6.2f   |  %6.2f  %6.2f  %6.2f  %6.2f  %6.2f' % (i, (ma[i] - me[i - 1]) / 100.0, (me[i] - ma[i]) / 100.0,
                  (li[i] - le[i - 1]) / 100.0, (lw[i] - li[i]) / 100.0, (ls[i] - lw[i]) / 100.0, (la[i] - ls[i]) / 100.0, (le[i] - la[i]) / 100.0))
eng.close()

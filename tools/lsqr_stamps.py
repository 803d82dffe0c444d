#!/usr/bin/env python3
"""Phase breakdown of the fused LSQR kernels from in-kernel stamps (knob lsqr_stamps = 1).  GPU only."""
import os, sys, ctypes
os.environ['QMRI_DEBUG'] = 'lsqr_stamps=1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qmri_pnp_recon_poc_amd import synth, engine as E
from qmri_pnp_recon_poc_amd._lib import lib as load

dic, q, X0 = synth.make_case(N=224, T=200, s=10, K=(128, 64), slice_seed=0)
fp, k = E.build_spiral(224, 771, 200)
eng = E.Engine(0)
eng.set_operator(224, 224, dic["V"], fp, k, max_batch=1)
y = eng.forward(X0)
x0 = eng.adjoint(y)
x, it, flag = eng.xupdate(y, x0 * 0.9, 0.05, 1e-4, int(sys.argv[1]) if len(sys.argv) > 1 else 6, x0=x0)
lib = load()
lib.qmri_debug_lsqr_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = np.zeros((2, 512, 16), np.uint64)
assert lib.qmri_debug_lsqr_stamps(eng.h, buf.ctypes.data) == 0
print('lsqr iters', it, 'flag', flag)
for kid, name, order in ((0, 'k_ks_a', [0, 1, 2, 3]), (1, 'k_ks_b', [0, 1, 2, 3, 4])):
    nb = int((buf[kid, :, 0] != 0).sum())
    s = buf[kid, :nb][:, order].astype(np.int64)
    nst = len(order)
    t0 = s[:, 0].min()
    rel = (s - t0) / 100.0                     # us
    print(name, 'blocks', nb, ' kernel span %.1f us' % rel[:, -1].max())
    print('  start: min %.1f max %.1f' % (rel[:, 0].min(), rel[:, 0].max()))
    d = np.diff(rel, axis=1)
    for k in range(nst - 1):
        print('  phase %2d->%2d : mean %.2f  max %.2f (block %d)' % (order[k], order[k + 1], d[:, k].mean(), d[:, k].max(), d[:, k].argmax()))
    tot = rel[:, -1] - rel[:, 0]
    print('  block total: mean %.1f max %.1f (block %d)' % (tot.mean(), tot.max(), tot.argmax()))

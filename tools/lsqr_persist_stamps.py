import os, sys, ctypes
os.environ['QMRI_DEBUG'] = 'lsqr_stamps=1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qmri_pnp_recon_poc_amd import synth, engine as E
from qmri_pnp_recon_poc_amd._lib import lib as load
dic = synth.make_dictionary(T=200, n_t1=16, n_t2=8, s=10)
fp, k = E.build_spiral(224, 771, 200)
eng = E.Engine(0)
eng.set_operator(224, 224, dic["V"], fp, k, max_batch=1)
rng = np.random.default_rng(0)
y = rng.standard_normal(eng.m) + 1j * rng.standard_normal(eng.m)
z = rng.standard_normal((224, 224, 10)) + 0j
x0 = eng.adjoint(y)
for _ in range(3): x, it, flag = eng.xupdate(y, z, 0.05, 1e-30, 100, x0=x0)
lib = load()
lib.qmri_debug_lsqr_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
buf = np.zeros((2, 512, 16), np.uint64)
assert lib.qmri_debug_lsqr_stamps(eng.h, buf.ctypes.data) == 0
s = buf[0].astype(np.int64)
nb = int((s[:, 0] != 0).sum())
s = s[:nb, :9]
names = ['stage v in LDS + A v sums (before the wait)', '|v|^2 all-reduce + barrier + finish u', 'block_sum2', 'publish + scatter group sums', '|u|^2 all-reduce + barrier', 'scalars', 'vector updates', 'block_sum + publish + barrier']
d = np.diff(s, axis=1) / 100.0
print('iteration 50 of k_ks_persist, %d workgroups; us per phase: mean / max' % nb)
for k in range(8): print('  %-44s %6.2f %6.2f' % (names[k], d[:, k].mean(), d[:, k].max()))
print('  iteration total mean %.2f; first top %.2f .. last top %.2f us spread' % ((s[:, 8] - s[:, 0]).mean() / 100.0, 0.0, (s[:, 0].max() - s[:, 0].min()) / 100.0))

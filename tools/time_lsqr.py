#!/usr/bin/env python3
"""100 LSQR iterations per x-update (tol far below reach) on the headline operator, a few times: run under rocprofv3 --kernel-trace --stats to read
the per-iteration cost of k_ks_persist (one launch = 100 iterations) or of k_ks_a + k_ks_b (QMRI_DEBUG="lsqr_persist=0")."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth
dic = synth.make_dictionary(T=200, n_t1=16, n_t2=8, s=10)
fp, k = E.build_spiral(224, 771, 200)
e = E.Engine(0)
e.set_operator(224, 224, dic["V"], fp, k)
rng = np.random.default_rng(0)
y = rng.standard_normal(e.m) + 1j * rng.standard_normal(e.m)
z = rng.standard_normal((224, 224, 10)) + 0j
x0 = e.adjoint(y)
for _ in range(2): e.xupdate(y, z, 0.05, 1e-30, 100, x0, solver="lsqr")
t0 = time.perf_counter()
for _ in range(5): x, it, fl = e.xupdate(y, z, 0.05, 1e-30, 100, x0, solver="lsqr")
print("iterations", it, "flag", fl, "host ms per x-update", (time.perf_counter() - t0) / 5 * 1e3)

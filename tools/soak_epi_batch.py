import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from qmri_pnp_recon_poc_amd import engine as E, synth
N, s, T = 224, 10, 200
dic = synth.make_dictionary(T=T, n_t1=32, n_t2=16, s=s)
fp, k = E.build_epi(N, N, 1 / 65, T)
w = synth.structured_weights(in_nc=11, out_nc=10, seed=5, eps=0.02)
e = E.Engine(0)
e.set_operator(N, N, dic["V"], fp, k, max_batch=15)
e.set_denoiser(w, N, N, in_nc=11, out_nc=10, max_batch=15)
rng = np.random.default_rng(0)
X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=1), dic)
y0 = e.forward(X0.astype(np.float64))
ys = np.stack([synth.awgn_measured(y0, 30.0, seed=i) for i in range(30)])
for rep in range(3):
    t0 = time.perf_counter()
    X, li = e.pnp_admm_batch(ys, slices_per_launch=15, iters=100, multi_level=True)
    dt = time.perf_counter() - t0
    print("rep", rep, "seconds", round(dt, 3), "slice-iterations/s", round(30 * 100 / dt, 1), "lsqr iters mean", float(li.mean()), "finite", bool(np.isfinite(X).all()), flush=True)
e.close()

#!/usr/bin/env python3
"""The two drop-in modes of INTEGRATION.md section 1 side by side on the headline problem (224 x 224 x 10, spiral mask, T = 200):

  per-call mode -- the reference's own PnP_ADMM.m keeps running on the host and calls the handles: every F.forward / F.adjoint / param.net is
                   one boundary crossing with host arrays (what MATLAB's lsqr does through afun: one forward + one adjoint per LSQR iteration,
                   PnP_ADMM.m:102,153-171; one param.net per ADMM iteration, :124);
  fused mode    -- PnP_ADMM_hip: one crossing per reconstruction (qmri_pnp_admm).

Times the host-array entry points (copies over PCIe and synchronisation included, as a MEX call pays them) and prints one JSON line: ms per
call, the iteration time the per-call mode implies for the LSQR counts the fused run reports, and the fused mode's own time per iteration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

N, s, T, ITERS = 224, 10, 200, 20
dic = synth.make_dictionary(T=T, n_t1=32, n_t2=16, s=s)
fp, k = E.build_spiral(N, 771, T)
w = synth.structured_weights(seed=2, eps=0.02)
e = E.Engine(0)
e.set_operator(N, N, dic["V"], fp, k)
e.set_denoiser(w, N, N)
X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=1), dic)
y = synth.awgn_measured(e.forward(X0.astype(np.float64)), 30.0, seed=1)
x = e.adjoint(y)


def ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


# the C entry points themselves on prepared column-major buffers (a MEX call hands MATLAB's arrays over as they are; no Python reshaping inside the clock)
import ctypes as C  # noqa: E402
L, h = e.L, e.h
vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
xb = np.ascontiguousarray(x.ravel(order="F"))
yb = np.ascontiguousarray(np.asarray(y, np.complex128).ravel())
yo = np.empty_like(yb); xo = np.empty_like(xb)
ri = np.ascontiguousarray(np.real(xb)); ro = np.empty_like(ri)
it, fl = C.c_int32(0), C.c_int32(0)


def chk(st):
    assert st == 0, L.qmri_last_error(h)


t_fwd = ms(lambda: chk(L.qmri_forward(h, vp(xb), 1, vp(yo))))
t_adj = ms(lambda: chk(L.qmri_adjoint(h, vp(yb), vp(xo))))
t_net = ms(lambda: chk(L.qmri_denoise(h, dp(ri), N, N, s, 1, dp(ro))))


def xup():
    xo[:] = xb
    chk(L.qmri_xupdate(h, vp(yb), vp(xb), 0.05, 1e-4, 100, 0, vp(xo), C.byref(it), C.byref(fl)))


t_xup = ms(xup, reps=10)
e.pnp_admm(y, iters=3)
t0 = time.perf_counter()
_, _, li = e.pnp_admm(y, iters=ITERS)
t_fused = (time.perf_counter() - t0) / ITERS * 1e3
lsqr_mean = float(np.mean(li))
# per-call mode: MATLAB's lsqr calls afun(v,'notransp') and afun(u,'transp') once per iteration (+ one of each to start), then one param.net
per_call_iter = (lsqr_mean + 1.0) * (t_fwd + t_adj) + t_net
print(json.dumps({
    "problem": "224 x 224 x 10, spiral mask, T = 200, 10-channel UNetRes; host arrays in and out of every call",
    "ms_per_call": {"F.forward (8 MB in, 2 MB out)": round(t_fwd, 3), "F.adjoint (2 MB in, 8 MB out)": round(t_adj, 3),
                    "param.net (single H x W x 10 in and out)": round(t_net, 3),
                    "x-update as ONE call (qmri_xupdate: y, z, x0 in, x out)": round(t_xup, 3)},
    "lsqr_iters_mean_first_%d_admm_iterations" % ITERS: round(lsqr_mean, 2),
    "per_call_mode_ms_per_admm_iteration_implied": round(per_call_iter, 2),
    "per_call_mode_with_xupdate_call_ms_per_admm_iteration": round(t_xup + t_net, 2),
    "fused_mode_ms_per_admm_iteration_host_clock": round(t_fused, 3),
    "fused_over_per_call": round(per_call_iter / t_fused, 1),
}))
e.close()

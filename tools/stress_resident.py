#!/usr/bin/env python3
"""Stress of k_conv6r's tile hand-offs: many forward passes, every output word compared with the one-launch-per-layer result (torch.equal on the
device), alone on the chip and beside a stream of unrelated kernels (uneven load: the hand-offs then see neighbours that arrive late and caches
that other work has touched).  A torn or stale granule shows up as a mismatch; a lost one as a time-out.
  python3 tools/stress_resident.py [passes]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = synth.random_weights(seed=1, gain=0.7)
rng = np.random.default_rng(5)
torch.cuda.init()
e = E.Engine(0)
e.set_denoiser(w, 224, 224)
xs = [torch.from_numpy(rng.random((10, 224, 224)).astype(np.float32)).cuda() for _ in range(4)]
y = torch.empty_like(xs[0])
fwd = lambda x, dst: e._check(e.L.qmri_net_forward_dev(e.h, C.c_void_p(x.data_ptr()), 1, C.c_void_p(dst.data_ptr())))
e.conv_resident(0)
refs = []
for x in xs:
    r = torch.empty_like(x); fwd(x, r); torch.cuda.synchronize(); refs.append(r.clone())
e.conv_resident(1)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
bad = 0
t0 = time.perf_counter()
for phase, load in (("alone", False), ("beside a stream of fp16 GEMMs and copies", True)):
    nb = 0
    for i in range(n_pass // 2):
        if load and i % 3 == 0:
            with torch.cuda.stream(side):
                b = a @ a if i % 6 == 0 else a.clone()                # (uneven: a GEMM that wants every CU, or a copy that wants the memory system)
        k = i % len(xs)
        fwd(xs[k], y)
        if not torch.equal(y, refs[k]):
            nb += 1
            if nb <= 3:
                d = (y - refs[k]).abs()
                print(f"  MISMATCH pass {i} ({phase}): {int((d > 0).sum())} words differ, max {float(d.max()):.3g}", flush=True)
    torch.cuda.synchronize()
    print(f"{phase}: {n_pass // 2} forward passes, {nb} mismatches, hand-off time-outs so far {e.conv_resident(1)}, scheme {e.denoiser_scheme()}", flush=True)
    bad += nb
print(f"{time.perf_counter() - t0:.1f} s")
e.close()
sys.exit(1 if bad else 0)

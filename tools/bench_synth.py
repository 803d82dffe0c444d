#!/usr/bin/env python3
"""Timing of the TSMI synthesis step (main_synthesize_tsmis.m:82-100) at the reference's size: one 230 x 230 slice against a
K = 98 304 entry look-up table.  Prints one JSON line; run under rocprofv3 --kernel-trace --stats for the kernel time."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

dic = synth.make_dictionary(T=200, n_t1=384, n_t2=256, s=10)
q = synth.make_phantom_qmaps(230, seed=0)
eng = E.Engine(0)
eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
eng.synthesize_tsmi(q)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    X, idx = eng.synthesize_tsmi(q)
dt = (time.perf_counter() - t0) / reps
K, npix = int(dic["K"]), 230 * 230
out = {"metric": "TSMI synthesis slices/sec (230x230 maps, nearest of K look-up-table entries)", "value": round(1.0 / dt, 2), "unit": "slices/s", "K": K,
       "npix": npix, "ms_per_slice": round(dt * 1e3, 3), "distance_evaluations": K * npix, "entry_point": "qmri_synthesize_tsmi (host buffers)"}
if "--cpu" in sys.argv:
    from oracle import oracle as O
    O.build()
    t0 = time.perf_counter()
    Xo, io = O.synthesize_tsmi(q, dic["D"], dic["normD"], dic["lut"])
    tc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": round(1.0 / tc, 3), "unit": "slices/s", "cores": O.num_threads(), "kind": "port", "sample": f"the same slice, exhaustive search, {tc:.1f} s"}
    out["indices_equal_to_oracle"] = bool(np.array_equal(idx, io)) and bool(np.array_equal(X, Xo))
print(json.dumps(out))

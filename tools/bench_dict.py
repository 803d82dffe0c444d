#!/usr/bin/env python3
"""Timing of the dictionary match (mrf_dtm_cpu.m) at the bench size: one 224 x 224 slice against K = 98 304 atoms.

  --s 10     (default) the compressed atoms of the shipped script: f16 filter + exact f32 products on the listed tiles (dict_kernels.hip)
  --s 1000   uncompressed fingerprints, s = T (BASELINE configs[4]; mrf_dtm_cpu.m:41-50 is T-generic): the channel-blocked f32-MFMA GEMM
             of dictw_kernels.hip, 19.7 TFLOP per slice
Prints one JSON line; run under rocprofv3 --kernel-trace --stats for the kernel time.  --cpu adds the oracle on the box's host cores
(s > 16: on a bounded sample of pixels, scaled)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qmri_pnp_recon_poc_amd import engine as E, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--s", type=int, default=10)
ap.add_argument("--n-t1", type=int, default=384)
ap.add_argument("--n-t2", type=int, default=256)
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--dev-reps", type=int, default=0, help="device-resident matches timed (default 5 wide / 20 narrow)")
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--cpu-pixels", type=int, default=1024, help="s > 16: pixels of the slice the CPU oracle is timed on")
args = ap.parse_args()

F32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
s = args.s
wide = s > 16
if wide:
    dic = synth.make_dictionary(T=s, n_t1=args.n_t1, n_t2=args.n_t2, uncompressed=True)
else:
    dic = synth.make_dictionary(T=200, n_t1=args.n_t1, n_t2=args.n_t2, s=s)
X = synth.synthesize_tsmi(synth.make_phantom_qmaps(224, seed=0), dic).astype(np.complex128)
X *= np.exp(0.3j)
if wide:                                                     # (some noise: the uncompressed match is what one runs on un-denoised data)
    rng = np.random.default_rng(0)
    X += 0.02 * X.real.std() * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
torch.cuda.init()                                            # (torch's HIP runtime first: the library then shares it)
eng = E.Engine(0)
eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
m = eng.dict_match(X)
reps = args.reps or (3 if wide else 5)
t0 = time.perf_counter()
for _ in range(reps):
    m = eng.dict_match(X)
dt = (time.perf_counter() - t0) / reps
K, npix = int(dic["K"]), 224 * 224
flop = 2 * 2 * npix * K * s
# device-resident timing (HIP events around the launches of one match), the figure the roofline is quoted on
dX = torch.from_numpy(np.ascontiguousarray(X.reshape(-1, s).T)).cuda()     # (s, Npix) complex128 = column-major Npix x s
o_q = torch.empty((2, npix), dtype=torch.float32, device="cuda"); o_pd = torch.empty((npix, 2), dtype=torch.float32, device="cuda")
o_dm = torch.empty(npix, dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()                                 # (a stream of torch's the engine launches on: the events see the kernels)
torch.cuda.synchronize()
eng.set_stream(stream.cuda_stream)


def dev_ms(reps):
    with torch.cuda.stream(stream):
        for _ in range(1 if wide else 3):
            eng.dict_match_dev(dX.data_ptr(), npix, o_q.data_ptr(), o_pd.data_ptr(), 0, o_dm.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            eng.dict_match_dev(dX.data_ptr(), npix, o_q.data_ptr(), o_pd.data_ptr(), 0, o_dm.data_ptr())
        e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms_f = dev_ms(args.dev_reps or (5 if wide else 20))
assert np.array_equal(o_dm.cpu().numpy(), m["dm"].ravel(order="C"))
out = {"metric": f"dictionary match slices/sec (224x224x{s} TSMI, K atoms)", "value": round(1.0 / dt, 3), "unit": "slices/s", "K": K, "npix": npix, "s": s,
       "ms_per_slice": round(dt * 1e3, 3), "flop_per_slice": flop,
       "entry_point": f"qmri_dict_match (host buffers: {X.nbytes / 1e6:.0f} MB in, 1.4 MB out)",
       "f32_mfma_peak_tflops": F32_MFMA_PEAK_TFLOPS}
if wide:
    traffic, tsrc = None, None
    try:                                                     # bytes leaving the L2s per launch, from the committed PMC passes of this size
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "dictw_traffic.json")) as f:
            tj = json.load(f)
        if (tj["s"], tj["K"], tj["npix"]) == (s, K, npix):
            traffic, tsrc = tj["corrected_bytes_per_launch"], tj["source"] + "; " + tj["note"]
    except (OSError, KeyError, ValueError):
        pass
    out["device_resident"] = {"ms_per_slice": round(ms_f, 3),
                              "what": "qmri_dict_match_dev, HIP events over 5 matches: k_dictw_pack_x (single(x) into fragment order) + k_dictw_match + k_dict_merge"}
    out["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS, "achieved": round(flop / (ms_f * 1e-3) / 1e12, 1),
                       "frac": round(flop / (ms_f * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3), "traffic": traffic, "traffic_source": tsrc,
                       "algorithmic_bytes": 4 * (K * s + 2 * npix * s) + 16 * npix,
                       "note": "algorithmic = executed flops 2*2*Npix*K*s of the single-precision product mrf_dtm_cpu.m:91 on v_mfma_f32_32x32x2_f32 (channels "
                               "padded to 16: + 0.8 % executed at s = 1000) against the f32 MFMA peak; whole match incl. the X conversion and the merge"}
else:
    eng.dict_filter(False)
    ms_x = dev_ms(20)
    eng.dict_filter(True)
    # what the default path EXECUTES: per (32-atom tile, 32-pixel tile) pair 6 v_mfma_f32_32x32x16_f16 (3 products x (re, im); K = 16 >= s) in the
    # filter, plus the exact f32 products of the ~1 % listed tiles (not counted)
    pairs = ((K + 31) // 32) * ((npix + 31) // 32)
    exec_f16 = pairs * 6 * 32 * 32 * 16 * 2
    out["device_resident"] = {"ms_per_slice": round(ms_f, 4), "ms_per_slice_exact_products_only": round(ms_x, 4),
                              "what": "qmri_dict_match_dev, HIP events over 20 matches; default = f16 filter + exact products on the listed tiles"}
    out["roofline"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": F16_MFMA_PEAK_TFLOPS,
                       "achieved": round(exec_f16 / (ms_f * 1e-3) / 1e12, 1), "frac": round(exec_f16 / (ms_f * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 3),
                       "executed_flop": exec_f16, "traffic": None,
                       "note": "default path, against the pipe it uses: executed f16-MFMA flops of the filter (6 x v_mfma_f32_32x32x16_f16 per 32 x 32 tile pair, "
                               f"{(16 - s) / 16:.0%} of them K-padding {s} -> 16) / time, against the 2.5 PFLOP/s dense f16 peak",
                       "algorithmic_f32_tflops": round(flop / (ms_f * 1e-3) / 1e12, 1),
                       "algorithmic_over_f32_mfma_peak": round(flop / (ms_f * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3),
                       "exact_products_only": {"bound": "mfma", "peak": F32_MFMA_PEAK_TFLOPS, "achieved": round(flop / (ms_x * 1e-3) / 1e12, 1),
                                               "frac": round(flop / (ms_x * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3),
                                               "note": "qmri_debug_dict_filter(ctx, 0, ..): every tile through v_mfma_f32_32x32x2_f32, algorithmic = executed"}}
if args.cpu:
    from oracle import oracle as O
    O.build()
    if wide:                                                 # bounded sample: cpu-pixels pixels spread over the slice, scaled to the slice
        n = min(args.cpu_pixels, npix)
        sel = np.linspace(0, npix - 1, n).astype(np.int64)
        Xs = X.reshape(npix, s, order="F")[sel]
        t0 = time.perf_counter()
        mo = O.dict_match(Xs, dic["D"], dic["normD"], dic["lut"])
        tc = (time.perf_counter() - t0)
        out["cpu_baseline"] = {"value": round(n / npix / tc, 5), "unit": "slices/s", "cores": O.num_threads(), "kind": "port",
                               "sample": f"{n} of the slice's {npix} pixels against all {K} atoms, {tc:.1f} s, scaled by {npix / n:.0f}",
                               "gflops": round(flop * n / npix / tc / 1e9, 1)}
        out["indices_equal_to_oracle"] = bool(np.array_equal(m["dm"].ravel(order="F")[sel], mo["dm"]))
    else:
        t0 = time.perf_counter()
        mo = O.dict_match(X, dic["D"], dic["normD"], dic["lut"])
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / tc, 3), "unit": "slices/s", "cores": O.num_threads(), "kind": "port", "sample": f"the same slice, {tc:.1f} s"}
        out["indices_equal_to_oracle"] = bool(np.array_equal(m["dm"], mo["dm"]))
print(json.dumps(out))

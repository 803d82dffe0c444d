#!/usr/bin/env python3
"""Measurement of the LRTV solver option in bench.py's JSON format (one line).

Step = one FISTA iteration of FISTA_deep (main_recon_tsmis_FFT.m:273-282: K = 4e-5, backtracking, TV prox with its default
tolerance) on the headline slice (cut3, 224 x 224 x 10, spiral S = 771, 30 dB noise).  `value` comes from one qmri_lrtv call
through the host-pointer entry point (2 MB of measurements in, 8 MB of image out, workspace allocation included).
roofline: the prox_tv iteration kernel k_tv_iter -- HBM/L2 streaming of ten R x C double arrays per launch (reads b, r, s,
pold, qold; writes r, s, pold, qold, sol), duration live from the kernels' dispatch timestamps (qmri_profile_enable(2)).
cpu_baseline: the CPU oracle (oracle.fista_lrtv, `kind: port`) on the box's host cores, first iterations of the same problem.

    python tools/bench_lrtv.py [--steps 50] [--cpu-iters 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md
# HBM-side bytes per full k_tv_iter launch from two separate rocprofv3 --pmc passes (profiles/r01_i_pmc_lrtv_traffic.txt), raw counters
TV_ITER_PMC_TRAFFIC_BYTES = (24545 + 39231) * 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from qmri_pnp_recon_poc_amd import engine as E, synth
    N, T, s, S = 224, 200, 10, 771
    dic = synth.make_dictionary(T=T, n_t1=128, n_t2=64, s=s)
    fp, k = E.build_spiral(N, S, T)
    eng = E.Engine(0)
    eng.set_operator(N, N, dic["V"], fp, k)
    X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)
    y = synth.awgn_measured(eng.forward(X0), 30.0, seed=0)
    eng.lrtv(y, iters=max(args.warmup, 1))
    t0 = time.perf_counter()
    x, info = eng.lrtv(y, iters=args.steps, tol=1e-300)                  # (tol: run exactly --steps iterations)
    dt = time.perf_counter() - t0
    eng.profile_enable(2)
    eng.lrtv(y, iters=min(args.steps, 10), tol=1e-300)
    pr = eng.profile_get(reset=True)
    eng.profile_enable(0)
    R, C = 2 * N, N * s
    bytes_per_launch = 10 * R * C * 8
    avg_s = pr["ms_tv_iter"] / max(pr["n_tv_iter"], 1) * 1e-3
    roof = {"kernel": "k_tv_iter (one prox_tv iteration: sol = b - gamma div(r,s), objective shares, dual update + projection + momentum)",
            "bound": "hbm", "achieved": round(bytes_per_launch / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bytes_per_launch / avg_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": TV_ITER_PMC_TRAFFIC_BYTES, "avg_launch_us": round(avg_s * 1e6, 2),
            "launches_timed": int(pr["n_tv_iter"]), "bytes_per_launch": bytes_per_launch}
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        O.build()
        fo, ko = O.spiral_mask(N, S, T)
        op = O.Operator(N, N, dic["V"], fo, ko)
        t0 = time.perf_counter()
        xo, io = O.fista_lrtv(op, y, iters=args.cpu_iters)
        tc = time.perf_counter() - t0
        cpu = {"value": round(args.cpu_iters / tc, 4), "unit": "FISTA iters/s", "cores": O.num_threads(), "kind": "port",
               "sample": f"first {args.cpu_iters} FISTA iterations of the same slice (operator + TV prox fp64), {tc:.1f} s"}
    print(json.dumps({"metric": "LRTV FISTA iters/sec (224x224x10 TSMI, spiral mask)", "value": round(args.steps / dt, 3), "unit": "FISTA iters/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "cut3 224x224x10 single slice, spiral mask S=771 T=200, LRTV (FISTA_deep: K=4e-5, backtracking, prox_tv tol 10e-4)",
                                 "entry_point": "qmri_lrtv (host buffers: H2D 2 MB, D2H 8 MB and workspace allocation inside the timed call)"},
                      "roofline": roof, "cpu_baseline": cpu, "prox_iters_per_call": round(info["prox_iters_total"] / max(info["prox_calls"], 1), 2),
                      "halvings": info["halvings"]}))


if __name__ == "__main__":
    main()

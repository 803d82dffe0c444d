#!/usr/bin/env python3
"""Benchmark of the PnP-ADMM MRF hot path on MI355X (contract: see the task description / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE PnP-ADMM iteration (LSQR x-update + normalise + UNetRes denoiser + un-normalise + dual update,
PnP_ADMM.m:93-146) on one 224 x 224 x 10 spiral-masked TSMI slice (BASELINE.json configs[1]: cut3, T = 200, S = 771,
30 dB measured AWGN, gamma = 0.05, LSQR tol 1e-4 / maxit 100, single-level 10-channel DRUNet).  Every rank owns
one GPU and reconstructs its own slice (slices are independent: weak scaling, no collective in the data path);
`value` = ADMM iterations of all ranks / wall time of the slowest rank.  Inputs (y) are resident in HBM when the
timed region starts.  Data and weights are synthetic (seeded; the reference ships neither).

The JSON line also carries
  roofline     -- the dominant kernel (k_conv6: conv3x3 as implicit GEMM on v_mfma_f32_32x32x16_f16, every fp32 operand
                  split into two f16 pieces (hi, scaled residual), three MFMA products per fp32-equivalent product; with
                  QMRI_CONV_SCHEME=bf16x6: three bf16 pieces, six products): MFMA FLOP executed per launch (3 x, resp. 6 x
                  the algorithmic 2*Cout*Cin*9*H*W) / mean launch duration measured live with HIP events on the launch
                  stream, against the 2.5 PFLOP/s dense f16/bf16 MFMA peak; `fp32_equivalent_tflops` is the algorithmic rate
  cpu_baseline -- the CPU oracle (a C/OpenMP restatement of the shipped algorithm, `kind: port`) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
`--workload slices` instead times whole slices (100 ADMM iterations + dictionary match) over a per-GPU batch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32 matrix peak (= f32 vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 / f16 matrix peak (v_mfma_f32_32x32x16_{bf16,f16}, 32 cycles)
SUSTAINED_MFMA_TFLOPS = round(2 * 32 * 32 * 16 * 1024 / 22.1e-9 / 1e12, 1)      # 1024 SIMDs x one 32x32x16 MFMA per 22.1 ns = 1518.3
BF16X6 = os.environ.get("QMRI_CONV_SCHEME", "") == "bf16x6"
SPLIT_PRODUCTS = 6 if BF16X6 else 3   # MFMA products per fp32-equivalent product (conv6_kernels.hip: bf16 x 6 / f16 x 3)
SCHEME_TEXT = ("v_mfma_f32_32x32x16_bf16, operands split 3-way into bf16, 6 products, f32 accumulate" if BF16X6 else
               "v_mfma_f32_32x32x16_f16, operands split into f16 (hi, scaled residual), 3 products, f32 accumulate")
# HBM-side bytes per launch of the dominant kernel at the 224 x 224 x 64 level, from two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; profiles/r01_g_pmc_conv_traffic.txt): 32 803 KB + 11 956 KB, raw counters.  Algorithmic: input with
# halo 16.3 MB + residual 12.8 MB (every second layer) + output 12.8 MB + weights 0.2 MB per XCD.
CONV6_PMC_TRAFFIC_BYTES = (32803 + 11956) * 1024
CONV3X3_FLOP = 2 * 64 * 64 * 9 * 224 * 224      # 3 699 376 128: identical at all four UNetRes levels
DENOISER_FLOP = 213_253_619_712                 # SURVEY.md section 8d (10-channel UNetRes at 224 x 224)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["admm", "slices"], default="admm")
    ap.add_argument("--slices-per-gpu", type=int, default=15)
    ap.add_argument("--batch", type=int, default=15, help="slices advanced together on one GPU (workload=slices): the whole per-GPU share in one launch sequence "
                    "(measured 8.8 / 9.7 / 10.1 slices/s at 5 / 8 / 15)")
    ap.add_argument("--solver", choices=["lsqr", "direct"], default="lsqr")
    ap.add_argument("--dict-k", type=int, nargs=2, default=[384, 256], help="dictionary grid n_t1 n_t2 (K = product)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend (gloo + --one-device: rehearsal of the "
                    "multi-rank path on a single-GPU box; the ranks then share device 0, so the value is not a scaling result)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (rehearsal only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if args.one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libqmri has no CPU path")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from qmri_pnp_recon_poc_amd import engine as E, synth
    import ctypes as C
    from qmri_pnp_recon_poc_amd._lib import AdmmParams

    N, T, s, S = 224, 200, 10, 771
    B = args.batch if args.workload == "slices" else 1
    dic = synth.make_dictionary(T=T, n_t1=args.dict_k[0] if args.workload == "slices" else 128,
                                n_t2=args.dict_k[1] if args.workload == "slices" else 64, s=s)
    fp, k = E.build_spiral(N, S, T)
    weights = synth.structured_weights(seed=2, eps=0.02)
    eng = E.Engine(local_rank)
    eng.set_operator(N, N, dic["V"], fp, k, max_batch=B)
    eng.set_denoiser(weights, N, N, max_batch=B)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def make_y(seed):
        q = synth.make_phantom_qmaps(N, seed=seed)
        X0 = synth.synthesize_tsmi(q, dic)
        y = eng.forward(X0)
        return synth.awgn_measured(y, 30.0, seed=seed)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = N * N * s
    result = {}
    if args.workload == "admm":
        y = make_y(rank)
        d_y = torch.from_numpy(np.ascontiguousarray(y).view(np.float64)).to(dev)
        d_x = torch.empty(2 * n, dtype=torch.float64, device=dev)
        li = np.zeros(max(args.steps, args.warmup, 1), np.int32)

        def run(iters):
            p = AdmmParams(0.05, iters, 1e-4, 100, 0 if args.solver == "lsqr" else 1, 0, 0.01, 0)
            st = eng.L.qmri_pnp_admm_dev(eng.h, 1, C.c_void_p(d_y.data_ptr()), C.byref(p), None, None, C.c_void_p(d_x.data_ptr()),
                                         None, li.ctypes.data_as(C.POINTER(C.c_int32)))
            eng._check(st)

        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        steps_done = args.steps
        lsqr_mean = float(li[: args.steps].mean()) if args.steps else 0.0
        unit_count = args.steps                                  # ADMM iterations per rank
        result_extra = {"lsqr_iters_mean": lsqr_mean}
    else:
        nsl = args.slices_per_gpu
        eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
        ys = np.stack([make_y(rank * nsl + i) for i in range(nsl)])
        d_y = torch.from_numpy(np.ascontiguousarray(ys).view(np.float64)).to(dev)
        d_x = torch.empty((B, 2 * n), dtype=torch.float64, device=dev)
        d_q = torch.empty(N * N * 2, dtype=torch.float32, device=dev)
        d_pd = torch.empty(N * N * 2, dtype=torch.float32, device=dev)
        m = eng.m

        def run_slices(count, iters):
            p = AdmmParams(0.05, iters, 1e-4, 100, 0 if args.solver == "lsqr" else 1, 0, 0.01, 0)
            for s0 in range(0, count, B):
                cnt = min(B, count - s0)
                yptr = d_y.data_ptr() + s0 * m * 16
                eng._check(eng.L.qmri_pnp_admm_dev(eng.h, cnt, C.c_void_p(yptr), C.byref(p), None, None, C.c_void_p(d_x.data_ptr()), None, None))
                for i in range(cnt):
                    eng._check(eng.L.qmri_dict_match_dev(eng.h, C.c_void_p(d_x.data_ptr() + i * n * 16), N * N, C.c_void_p(d_q.data_ptr()),
                                                         C.c_void_p(d_pd.data_ptr()), None, None))

        run_slices(min(B, nsl), max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        run_slices(nsl, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        unit_count = nsl
        result_extra = {"admm_iters_per_slice": args.steps, "dict_K": int(dic["K"]), "slices_per_gpu": nsl, "batch": B}

    # max over ranks
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream ----------
    roof = None
    if rank == 0 and not args.no_roofline:
        eng.profile_get(reset=True)
        eng.profile_enable(2)
        d_in = torch.rand(B * s * N * N, dtype=torch.float32, device=dev)
        d_out = torch.empty(B * s * N * N, dtype=torch.float32, device=dev)
        for _ in range(3):
            eng._check(eng.L.qmri_net_forward_dev(eng.h, C.c_void_p(d_in.data_ptr()), B, C.c_void_p(d_out.data_ptr())))
        torch.cuda.synchronize()
        pr = eng.profile_get(reset=True)
        eng.profile_enable(0)
        if pr["n_conv3x3"] > 0:
            avg_s = pr["ms_conv3x3"] / pr["n_conv3x3"] * 1e-3
            f32_path = bool(int(os.environ.get("QMRI_CONV_F32", "0")))
            if f32_path:
                ach = CONV3X3_FLOP * B / avg_s / 1e12
                roof = {"kernel": "k_conv<3x3> (implicit-GEMM conv3x3 on v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                        "achieved": round(ach, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": None, "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": int(pr["n_conv3x3"]),
                        "flop_per_launch": CONV3X3_FLOP * B}
            else:
                ach = SPLIT_PRODUCTS * CONV3X3_FLOP * B / avg_s / 1e12
                roof = {"kernel": f"k_conv6 (implicit-GEMM conv3x3 on {SCHEME_TEXT})",
                        "bound": "mfma", "achieved": round(ach, 3), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": CONV6_PMC_TRAFFIC_BYTES * B, "avg_launch_us": round(avg_s * 1e6, 2),
                        "launches_timed": int(pr["n_conv3x3"]), "flop_per_launch": SPLIT_PRODUCTS * CONV3X3_FLOP * B,
                        "fp32_equivalent_tflops": round(CONV3X3_FLOP * B / avg_s / 1e12, 3), "fp32_equivalent_flop_per_launch": CONV3X3_FLOP * B,
                        # what the chip sustains on toggling operands with every SIMD issuing MFMAs back to back: 22.1 ns per 32x32x16
                        # MFMA and SIMD (tools/ubench/mfma_f16x3_loop.hip, profiles/r01_g_ubench_mfma_f16x3_loop.txt) -- not the roofline
                        # peak, reported beside it
                        "sustained_mfma_tflops_measured": SUSTAINED_MFMA_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_MFMA_TFLOPS, 4)}
        # stage split of one short run (profile level 1 synchronises per stage; not part of the timed region)
        if args.workload == "admm":
            eng.profile_enable(1)
            run(min(10, max(args.steps, 1)))
            pr = eng.profile_get(reset=True)
            eng.profile_enable(0)
            it = max(pr["admm_iters"], 1)
            result_extra["stage_ms_per_iter"] = {"xupdate": round(pr["ms_xupdate"] / it, 4), "denoiser": round(pr["ms_denoiser"] / it, 4),
                                                 "elementwise": round(pr["ms_elementwise"] / it, 4)}

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        O.build()
        fo, ko = O.spiral_mask(N, S, T)
        op = O.Operator(N, N, dic["V"], fo, ko)
        net = O.Net(weights)
        y0 = make_y(0) if args.workload != "admm" else y
        cores = O.num_threads()
        t0 = time.perf_counter()
        O.pnp_admm(op, net, y0, gamma=0.05, iters=args.cpu_iters, cg_tol=1e-4, cg_maxit=100, solver="lsqr")
        tc = time.perf_counter() - t0
        if args.workload == "admm":
            cpu = {"value": round(args.cpu_iters / tc, 4), "unit": "ADMM iters/s", "cores": cores, "kind": "port",
                   "sample": f"first {args.cpu_iters} PnP-ADMM iterations of the same slice (LSQR tol 1e-4 fp64 + UNetRes fp32), {tc:.1f} s"}
        else:
            cpu = {"value": round(args.cpu_iters / tc / args.steps, 6), "unit": "slices/s", "cores": cores, "kind": "port",
                   "sample": f"{args.cpu_iters} PnP-ADMM iterations of one slice extrapolated to {args.steps} per slice, match excluded, {tc:.1f} s"}

    if rank == 0:
        total_units = unit_count * world
        if args.workload == "admm":
            metric, unit = "ADMM iters/sec (224x224x10 TSMI, spiral mask)", "ADMM iters/s"
            cfg = {"workload": "cut3 224x224x10 single slice per GPU, spiral mask S=771 T=200, PnP-ADMM + 10-channel UNetRes (DRUNet) denoiser",
                   "solver": args.solver, "admm_iters": args.steps, "dc_dtype": "f64",
                   "denoiser_arith": "f32 results: all convs on " + SCHEME_TEXT,
                   "parallelism": f"slice-parallel x{world} (no collective)"}
            ms_per_step = dt / max(args.steps, 1) * 1e3
        else:
            metric, unit = "slices/sec (120-slice synthetic batch: 100 ADMM iterations + dictionary match per slice)", "slices/s"
            cfg = {"workload": f"cut3 {unit_count}-slice batch per GPU, spiral mask, PnP-ADMM + UNetRes + dictionary match K={int(dic['K'])}",
                   "solver": args.solver, "admm_iters": args.steps, "dc_dtype": "f64", "parallelism": f"slice-parallel x{world} (no collective)"}
            ms_per_step = dt / max(unit_count, 1) * 1e3
        out = {"metric": metric, "value": round(total_units / dt, 4), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": cfg, "roofline": roof, "cpu_baseline": cpu}
        out.update(result_extra)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

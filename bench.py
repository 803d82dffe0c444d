#!/usr/bin/env python3
"""Benchmark of the PnP-ADMM MRF hot path on MI355X (contract: see the task description / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no launcher (WORLD_SIZE unset): this process -- which never touches the GPU -- starts N worker
processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per GPU), waits for them and exits with their
worst return code; rank 0 prints the JSON line.  Under `torch.distributed.run` the ranks are the workers directly.

A "step" is ONE PnP-ADMM iteration (LSQR x-update + normalise + UNetRes denoiser + un-normalise + dual update,
PnP_ADMM.m:93-146) on one 224 x 224 x 10 spiral-masked TSMI slice (BASELINE.json configs[1]: cut3, T = 200, S = 771,
30 dB measured AWGN, gamma = 0.05, LSQR tol 1e-4 / maxit 100, single-level 10-channel DRUNet).  Every rank owns
one GPU and reconstructs its own slice (slices are independent: weak scaling, no collective in the data path);
`value` = ADMM iterations of all ranks / wall time of the slowest rank.  Inputs (y) are resident in HBM when the
timed region starts.  Data and weights are synthetic (seeded; the reference ships neither).

The JSON line also carries
  roofline     -- the dominant kernel family (the 3x3 convolutions: conv3x3 as implicit GEMM on v_mfma_f32_32x32x16_f16, every fp32
                  operand split into two f16 pieces (hi, scaled residual), three MFMA products per fp32-equivalent product; with
                  QMRI_DEBUG="conv_scheme=3": three bf16 pieces, six products).  Pure measurement, no apportioning: every launch
                  of the family is timed from its own dispatch timestamps as ONE unit -- a layer launched alone (a split-K layer: from
                  the convolution's start to the end of the reduce kernel that completes it), or a resident-tile launch of a whole run
                  of layers with whatever else rides in it (head, tail, down-sampling convolution) -- and `achieved` = the MFMA FLOP
                  those units execute (3 x, resp. 6 x their algorithmic 2*Cout*Cin*taps*H*W) / the sum of their durations, against
                  the 2.5 PFLOP/s dense f16/bf16 MFMA peak.  `whole_denoiser` is the same ratio for the entire forward pass (3 x
                  213.25 GFLOP / its duration between two stream events).  `traffic` = HBM-side bytes per launch from the committed
                  rocprofv3 --pmc passes (profiles/, corrected as MI355X_MICROARCH.md section HBM prescribes; tools/pmc_traffic.py)
  xupdate      -- the data-consistency stage against the HBM roofline: algorithmic bytes of one LSQR iteration in the k-space
                  formulation (v, d, x, u(m+1:end) on the sampled k locations and u(1:m) on the samples, each read and written once,
                  complex fp64 as the reference's arithmetic; + the sample descriptors) / the measured duration of the iteration's
                  kernels, against 8 TB/s; and us per LSQR iteration
  epi_batch15, cut0 -- BASELINE.json configs[2] (EPI mask, 11-channel multi-level denoiser, 15 slices advanced together) and the reference-side part
                  of configs[4] (cut0: T = 1000, one slice): a short timed run each (value, stage split, x-update roofline)
  cpu_baseline -- the CPU oracle (a C/OpenMP restatement of the shipped algorithm, `kind: port`) timed on this
                  box's host cores on a bounded sample of the same workload (rank 0, N = 1 only): all usable threads and
                  one thread, per-stage split, diagnostics on (PnP_ADMM.m:106-109), CPU model string
  parity       -- SURVEY.md section 8(d) metric 3, computed outside the timed region by the same oracle run that is the CPU
                  baseline sample: mean per-channel PSNR of |x_gpu| against |x_oracle| (peak 1, main_recon_tsmis_FFT.m:362-367),
                  fraction of pixels whose matched atom (T1/T2) is identical, PD relative error
`--workload slices` instead times whole slices (100 ADMM iterations + dictionary match) over a per-GPU batch.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense f32 matrix peak (= f32 vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 / f16 matrix peak (v_mfma_f32_32x32x16_{bf16,f16}, 32 cycles)
# what the chip sustains on toggling f16 operands at its power limit, in the conv kernels' own instruction mix (f16 x 3 products on a 64 x 64 wave
# tile, every fragment read from LDS, one wave per SIMD on all 256 CUs): 1325-1345 TFLOP/s at 1.75-1.92 GHz (tools/ubench/mfma_shape_f16.hip,
# profiles/r03_h_ubench_mfma_shape_f16.txt; bare MFMAs without LDS reads: 22.1 ns each = 1518, profiles/r01_g_ubench_mfma_f16x3_loop.txt)
SUSTAINED_MFMA_TFLOPS = 1335.0
SUSTAINED_MFMA_TFLOPS_BARE = 1518.3   # ... bare MFMAs, no LDS reads (the figure `frac_of_sustained` used up to round 2; both are reported so rounds stay comparable)


_DICT_CACHE = {}


def cached_dictionary(synth, T, n_t1, n_t2, s):
    """synth.make_dictionary, once per parameter set and process (the K = 98 304 dictionary of the cold start, the slices phase and the CPU leg is one object)."""
    key = (int(T), int(n_t1), int(n_t2), int(s))
    if key not in _DICT_CACHE:
        _DICT_CACHE[key] = synth.make_dictionary(T=T, n_t1=n_t1, n_t2=n_t2, s=s)
    return _DICT_CACHE[key]


def debug_knob(name: str, default: int) -> int:
    """The library's A/B switches as it reads them itself: QMRI_DEBUG="name=value,name=value" (csrc/api_core.cpp)."""
    for kv in os.environ.get("QMRI_DEBUG", "").split(","):
        k, _, v = kv.partition("=")
        if k.strip() == name and v.strip().lstrip("-").isdigit():
            return int(v)
    return default


BF16X6 = debug_knob("conv_scheme", 2) == 3
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E peak (6.3 TB/s is what a float4 copy achieves)
SPLIT_PRODUCTS = 6 if BF16X6 else 3   # MFMA products per fp32-equivalent product (conv6_kernels.hip: bf16 x 6 / f16 x 3)
SCHEME_TEXT = ("v_mfma_f32_32x32x16_bf16, operands split 3-way into bf16, 6 products, f32 accumulate" if BF16X6 else
               "v_mfma_f32_32x32x16_f16, operands split into f16 (hi, scaled residual), 3 products, f32 accumulate")
CONV3X3_FLOP = 2 * 64 * 64 * 9 * 224 * 224      # 3 699 376 128: identical at all four UNetRes levels
DENOISER_FLOP = 213_253_619_712                 # SURVEY.md section 8d (10-channel UNetRes at 224 x 224)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "conv_traffic.json")   # written by tools/pmc_traffic.py from rocprofv3 --pmc passes


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["admm", "slices"], default="admm")
    ap.add_argument("--slices-per-gpu", type=int, default=15)
    ap.add_argument("--total-slices", type=int, default=0, help="workload=slices: a FIXED total (north_star: 120) sharded over the ranks in contiguous "
                    "blocks (batch.shard_slices) and walked in batches of --batch on each GPU: the same job at every N (scaling: strong); "
                    "0: --slices-per-gpu on every rank (weak)")
    ap.add_argument("--batch", type=int, default=15, help="slices advanced together on one GPU (workload=slices): the whole per-GPU share in one launch sequence "
                    "(measured 8.8 / 9.7 / 10.1 slices/s at 5 / 8 / 15)")
    ap.add_argument("--solver", choices=["lsqr", "direct"], default="lsqr")
    ap.add_argument("--dict-k", type=int, nargs=2, default=[384, 256], help="dictionary grid n_t1 n_t2 (K = product; workload=slices)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and with it the parity numbers)")
    ap.add_argument("--cpu-iters", type=int, default=0, help="ADMM iterations of the CPU sample (0: as many of --steps as fit --cpu-budget-s)")
    ap.add_argument("--cpu-budget-s", type=float, default=75.0, help="wall-clock bound of the all-threads CPU sample")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-slices", action="store_true", help="workload=admm: skip the `slices` object (north_star's second metric: a fixed 120-slice batch, "
                    "100 ADMM iterations + dictionary match at K = 98 304 per slice, sharded over the ranks; ~10 s on one GPU)")
    ap.add_argument("--no-secondary", action="store_true", help="workload=admm: skip the `epi_batch15` and `cut0` objects (BASELINE configs[2] and the reference-side "
                    "part of configs[4]: a few seconds each)")
    ap.add_argument("--secondary-steps", type=int, default=20, help="... ADMM iterations timed for each of them")
    ap.add_argument("--slices-total", type=int, default=120, help="... its slice count")
    ap.add_argument("--slices-batch", type=int, default=0, help="... slices advanced together per launch; 0 = auto: 30 where every rank holds at least 30 slices "
                    "(N <= 4 of the 120-slice batch), else 15 -- what a rank holds at 8 GPUs (A/B on one box: 13.26 vs 12.83 slices/s, profiles/r06_f_*)")
    ap.add_argument("--slices-iters", type=int, default=100, help="... ADMM iterations per slice (PnP_ADMM.m: param.iter = 100)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="gloo", help="process-group backend of the barrier and the max over ranks -- the data path has "
                    "no collective (north_star: slices shard without RCCL), so the default is gloo on the host; nccl (= RCCL) does the same two things on the "
                    "device (gloo + --one-device: rehearsal of the multi-rank path on a single-GPU box; the ranks then share device 0, so the value is not a scaling result)")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the `cold_start` object (set-up times and time to the first reconstructed slice, rank 0)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (rehearsal only)")
    ap.add_argument("--plumbing-only", action="store_true", help="rank start-up, rendezvous, barrier and max-over-ranks only: no engine, no GPU "
                    "(CPU test of the --gpus N path; the line carries value null)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# --gpus N without a launcher: N worker processes of this script, one per GPU.  The parent initialises nothing on the GPU
# (a process that has must never be replaced or forked into workers), it only waits.
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n: int) -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:            # a rank failed: the others would wait at the barrier for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def psnr_peak1(a, b) -> float:
    """MATLAB psnr(A, ref) for doubles: peak 1 (main_recon_tsmis_FFT.m:362-367)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0.0 else 10.0 * np.log10(1.0 / mse)


def parity_numbers(x_gpu, x_cpu, maps_gpu, maps_cpu, iters, dic=None) -> dict:
    """SURVEY.md section 8(d) metric 3: GPU output against the CPU restatement's output of the same reconstruction."""
    s = x_gpu.shape[-1]
    ps = [psnr_peak1(np.abs(x_gpu[..., c]), np.abs(x_cpu[..., c])) for c in range(s)]
    same = maps_gpu["dm"] == maps_cpu["dm"]
    diff = ~same
    t1g, t1c = maps_gpu["qmap"][..., 0], maps_cpu["qmap"][..., 0]
    t2g, t2c = maps_gpu["qmap"][..., 1], maps_cpu["qmap"][..., 1]
    pg, pc = np.abs(maps_gpu["pd"]).astype(np.float64), np.abs(maps_cpu["pd"]).astype(np.float64)
    finite = [p for p in ps if np.isfinite(p)]
    grid = {}
    if dic is not None and diff.any():
        # distance between the two atoms of a differing pixel in steps of the (T1, T2) grid (atom index = i_t1 * n_t2 + i_t2)
        n2 = int(dic["t2_grid"].size)
        ig, ic = maps_gpu["dm"][diff].astype(np.int64) - 1, maps_cpu["dm"][diff].astype(np.int64) - 1
        d1, d2 = np.abs(ig // n2 - ic // n2), np.abs(ig % n2 - ic % n2)
        grid = {"t1_grid_steps_mean_on_differing_px": round(float(d1.mean()), 3), "t2_grid_steps_mean_on_differing_px": round(float(d2.mean()), 3),
                "grid_steps_max_on_differing_px": int(max(d1.max(), d2.max()))}
    if dic is not None:
        grid["atom_index_identical_frac_bound_at_this_K"] = round(atom_tolerance(int(dic["K"])), 4)
        grid["dict_K"] = int(dic["K"])
    return {"admm_iters_compared": int(iters), **grid,
            "tsmi_rel_l2": float(np.linalg.norm((x_gpu - x_cpu).ravel()) / np.linalg.norm(x_cpu.ravel())),
            "tsmi_psnr_db_mean": round(float(np.mean(finite)), 2) if finite else None,
            "tsmi_psnr_db_min": round(float(np.min(finite)), 2) if finite else None,
            "atom_index_identical_frac": round(float(same.mean()), 6),
            "t1_mae_on_differing_px_s": round(float(np.abs(t1g[diff] - t1c[diff]).mean()), 6) if diff.any() else 0.0,
            "t2_mae_on_differing_px_s": round(float(np.abs(t2g[diff] - t2c[diff]).mean()), 6) if diff.any() else 0.0,
            "pd_rel_err": float(np.linalg.norm(pg - pc) / max(np.linalg.norm(pc), 1e-300)),
            "tolerance": "LSQR tol 1e-4 leaves a stop-rule ambiguity of ~2e-4 per x-update (SURVEY 8 a7): tsmi_rel_l2 <= 1e-3 is parity"}


def load_traffic(B: int):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes.  The passes are taken per kernel and batch size
    (recorded in the file): profiles/conv_traffic.json = k_conv6<0, 2> at one slice per launch, profiles/conv_traffic_batch<B>.json = the
    persistent k_conv6p of slice batches.  A run whose dominant kernel / batch has no pass reports null, not another grid's bytes."""
    path = TRAFFIC_FILE if B == 1 else os.path.join(ROOT, "profiles", f"conv_traffic_batch{B}.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if int(t.get("batch", 1)) != B:
            return None, f"no PMC pass for batch {B}"
        fam = t.get("family_per_forward")
        if fam:                                                    # round 5: the whole 3x3 family of a forward pass, the set of launches the roofline times
            return int(fam["corrected_bytes"]), ("per FORWARD PASS (all units of the roofline together): " + fam.get("what", "") + "; " + t.get("source", ""))
        return int(t["corrected_bytes_per_launch_per_slice"] * B), (t.get("kernel", "") + ": " + t.get("source", ""))
    except (OSError, KeyError, ValueError):
        return None, (None if B == 1 else f"no PMC pass for batch {B} (profiles/conv_traffic_batch{B}.json)")


def atom_tolerance(K: int) -> float:
    """Stated lower bound on the fraction of pixels matched to the IDENTICAL atom after a full reconstruction, as a function of the
    dictionary size.  The match is bit-exact for equal X; the two reconstructions differ by ~2e-5 (LSQR stop rule + fp32 summation
    order), which flips a pixel between neighbouring atoms whose correlations differ by less than that -- and the number of such
    near-ties grows with the grid density: 99.7 % measured at K = 8 192, 96.3 % at K = 98 304 (12 x denser).  Bound: 1 - 0.01 * K / 8192,
    floored at 0.85; what is asserted beside it is that differing pixels sit within two grid steps of each other."""
    return max(0.85, 1.0 - 0.01 * K / 8192.0)


def conv_roofline(pr, B, kernel_text, traffic=None, traffic_source=None):
    """`roofline` of the 3x3 convolution family from a level-2 profile (qmri_profile): executed MFMA flop of the timed units / their summed durations."""
    if pr["n_conv3x3"] <= 0 or pr["ms_conv3x3"] <= 0:
        return None
    t3 = pr["ms_conv3x3"] * 1e-3
    ach = SPLIT_PRODUCTS * pr["flop_conv3x3"] / t3 / 1e12
    leq = pr["flop_conv3x3"] / (CONV3X3_FLOP * B)                   # 64 -> 64 layers (of B slices) the timed units amount to
    nf = max(pr["n_net_forward"], 1)
    roof = {"kernel": kernel_text, "bound": "mfma", "achieved": round(ach, 3), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "traffic_unit": "HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE) of ALL timed units of one forward pass, when the committed PMC file carries them; "
                            "else per launch of the kernel named in traffic_source",
            "timing": "every launch of the family is ONE unit, timed from its own dispatch timestamps (hipExtLaunchKernelGGL events): a split-K layer from the "
                      "convolution's start to its reduce kernel's end, a resident-tile launch whole; achieved = executed flop of the units / sum of their durations",
            "units_timed": int(pr["n_conv3x3"]), "units_per_forward": round(pr["n_conv3x3"] / nf, 2), "ms_timed_per_forward": round(pr["ms_conv3x3"] / nf, 4),
            "executed_flop_per_forward": SPLIT_PRODUCTS * pr["flop_conv3x3"] / nf, "fp32_equivalent_flop_per_forward": pr["flop_conv3x3"] / nf,
            "layer_equivalents_per_forward": round(leq / nf, 3), "us_per_layer_equivalent": round(t3 / leq * 1e6, 3),
            "avg_launch_us": round(t3 / pr["n_conv3x3"] * 1e6, 2), "launches_timed": int(pr["n_conv3x3"]),
            "flop_per_launch": SPLIT_PRODUCTS * pr["flop_conv3x3"] / pr["n_conv3x3"],
            "fp32_equivalent_tflops": round(pr["flop_conv3x3"] / t3 / 1e12, 3),
            # what the chip sustains on toggling operands at its power limit (see SUSTAINED_MFMA_TFLOPS) -- not the roofline peak, reported beside it
            "sustained_mfma_tflops_measured": SUSTAINED_MFMA_TFLOPS, "frac_of_sustained": round(ach / SUSTAINED_MFMA_TFLOPS, 4),
            "sustained_mfma_tflops_bare_mfma_loop": SUSTAINED_MFMA_TFLOPS_BARE, "frac_of_sustained_bare_mfma_loop": round(ach / SUSTAINED_MFMA_TFLOPS_BARE, 4)}
    if pr["n_conv2x2"] > 0 and pr["ms_conv2x2"] > 0:
        tall, fall = t3 + pr["ms_conv2x2"] * 1e-3, pr["flop_conv3x3"] + pr["flop_conv2x2"]
        roof["all_conv_launches"] = {"what": "the same ratio over every convolution launch of the forward pass (3x3 family + the 2x2 / stride-2 layers launched alone)",
                                     "ms_per_forward": round(tall * 1e3 / nf, 4), "achieved": round(SPLIT_PRODUCTS * fall / tall / 1e12, 3),
                                     "frac": round(SPLIT_PRODUCTS * fall / tall / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
    if pr["n_net_forward"] > 0 and pr["ms_net_forward"] > 0:
        tf = pr["ms_net_forward"] / pr["n_net_forward"] * 1e-3
        ex = SPLIT_PRODUCTS * DENOISER_FLOP * B
        roof["whole_denoiser"] = {"what": "the whole forward pass between two stream events (all 64 layers, the |output| report, every launch gap): 3 x 213.25 GFLOP "
                                          "per slice / its duration -- UNDER THE PER-LAUNCH PROFILE, whose event bookkeeping makes the pass 20 - 25 % longer than inside "
                                          "the ADMM loop: `frac` above comes from the units' own dispatch timestamps, `whole_denoiser_in_admm_loop` from the loop's own stage",
                                  "ms_per_forward": round(tf * 1e3, 4), "executed_flop": ex,
                                  "achieved": round(ex / tf / 1e12, 3), "frac": round(ex / tf / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
    return roof


def xupdate_bytes_per_lsqr_iteration(ns: int, s: int, m: int) -> int:
    """ALGORITHMIC bytes of one LSQR iteration of the k-space formulation, per slice: the four vectors on the sampled k
    locations (v, d, x, u(m+1:end): ns * s complex doubles each) and u(1:m) (m complex doubles) are read and written once, the sample
    descriptors (4 bytes each) are read once."""
    return 16 * (2 * 4 * ns * s + 2 * m) + 4 * m


XUPDATE_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "xupdate_traffic.json")   # written by tools/pmc_xupdate.py from rocprofv3 --pmc passes


def load_xupdate_traffic(key):
    """HBM-side bytes of the LSQR iteration kernels per LSQR iteration (all slices of the launch sequence) from the committed PMC passes, by configuration
    key ("spiral_T200_B1", "epi_T200_B15", ...).  None when no pass exists for this configuration."""
    try:
        with open(XUPDATE_TRAFFIC_FILE) as f:
            t = json.load(f)
        e = t["configs"][key]
        return int(e["bytes_per_lsqr_iteration"]), e.get("kernels", "") + "; " + t.get("source", ""), t.get("one_launch_iteration_phases_us")
    except (OSError, KeyError, ValueError):
        return None, f"no PMC pass for {key} in profiles/xupdate_traffic.json (tools/pmc_xupdate.sh)", None


def xupdate_roofline(pr, ns, s, m, B, one_launch, traffic_key=None):
    """`xupdate` object from a level-1 + level-2 profile of an ADMM run: stage time, LSQR iterations, and the iteration kernels alone against HBM."""
    it = max(pr["admm_iters"], 1)
    li = max(pr["lsqr_iters"], 1)                                   # summed over slices
    byt = xupdate_bytes_per_lsqr_iteration(ns, s, m)
    out = {"lsqr_iters_per_xupdate": round(pr["lsqr_iters"] / it / B, 2), "bytes_per_lsqr_iteration_per_slice": byt,
           "form": ("k_ks_persist: all iterations of a solve in one launch, the iteration's state on chip" +
                    ("; the slices of a batch go through it as many per launch as are resident together (EPI: one, the spiral: two)" if B > 1 else ""))
                   if one_launch else
                   "two launches per LSQR iteration (k_ks_a, k_ks_b), the state streams through HBM / Infinity Cache"}
    if pr["n_lsqr_launches"] > 0 and pr["ms_lsqr_kernels"] > 0:
        # per LSQR iteration of the whole batch: the iteration kernels' own dispatch timestamps; iterations = the slowest slice's count summed over
        # the x-updates ~ lsqr_iters / B (slices of a batch iterate together)
        us = pr["ms_lsqr_kernels"] * 1e3 / (li / B)
        gbs = byt * B / (us * 1e-6) / 1e9
        traffic, tsrc, phases = load_xupdate_traffic(traffic_key) if traffic_key else (None, None, None)
        # What bounds this stage is LATENCY, not bandwidth: the one-launch form keeps the iteration's state in registers and LDS and an iteration is two
        # grid-wide all-reduces of tagged granules.  `achieved` is therefore the ALGORITHMIC rate (the contract's definition: algorithmic bytes / duration)
        # and is labelled as an effective rate; `traffic` is what the PMC counters saw leave / enter the L2s per iteration, and `traffic_gbs` that over the
        # same duration -- the figure an HBM roofline would bind on, far below the peak.
        out.update({"us_per_lsqr_iteration": round(us, 2), "us_per_lsqr_iteration_per_slice": round(us / B, 2),
                    "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "effective_gbs": round(gbs, 1), "bytes_per_launch_pair": byt * B, "algorithmic_bytes_per_lsqr_iteration": byt * B,
                                 "traffic": traffic, "traffic_source": tsrc,
                                 "traffic_gbs": round(traffic / (us * 1e-6) / 1e9, 1) if traffic else None,
                                 "traffic_over_algorithmic": round(traffic / (byt * B), 4) if traffic else None,
                                 "binding_limit": "latency: " + ("two grid-wide all-reduces per LSQR iteration inside one launch" if one_launch else
                                                                  "two dependent launches per LSQR iteration") + f" = {us:.2f} us per iteration",
                                 "one_launch_iteration_phases_us": phases if one_launch else None,
                                 "note": "achieved = algorithmic bytes of an iteration (all slices of the batch) / duration of the iteration's kernels from their own "
                                         "dispatch timestamps: an EFFECTIVE rate" + (" (the one-launch form keeps the state on chip; traffic = PMC bytes per iteration)" if one_launch else "")}})
    return out


def secondary_config(args, torch, dev, local_rank, name, T, mask, B, multi, steps):
    """One more BASELINE configuration on the default line: a short timed PnP-ADMM run (value = slice-iterations / s on this GPU), the stage split
    (profile level 1) and the x-update roofline (level 2), rank 0 only."""
    import ctypes as C
    from qmri_pnp_recon_poc_amd import engine as E, synth
    from qmri_pnp_recon_poc_amd._lib import AdmmParams
    N, s = 224, 10
    dic = synth.make_dictionary(T=T, n_t1=32, n_t2=16, s=s)
    fp, k = E.build_spiral(N, 771, T) if mask == "spiral" else E.build_epi(N, N, 1 / 65, T)
    w = synth.structured_weights(in_nc=s + (1 if multi else 0), seed=2, eps=0.02)
    eng = E.Engine(local_rank)
    try:
        return _secondary_config(args, torch, dev, eng, name, T, mask, B, multi, steps, dic, fp, k, w)
    finally:
        eng.close()                                                # (also on an exception: a leaked engine would make the NEXT configuration fail for want of memory)


def _secondary_config(args, torch, dev, eng, name, T, mask, B, multi, steps, dic, fp, k, w):
    import ctypes as C
    from qmri_pnp_recon_poc_amd import synth
    from qmri_pnp_recon_poc_amd._lib import AdmmParams
    N, s = 224, 10
    eng.set_operator(N, N, dic["V"], fp, k, max_batch=B)
    eng.set_denoiser(w, N, N, in_nc=s + (1 if multi else 0), max_batch=B)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    ys = np.stack([synth.awgn_measured(eng.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=i), dic)), 30.0, seed=i) for i in range(B)])
    d_y = torch.from_numpy(np.ascontiguousarray(ys).view(np.float64)).to(dev)
    d_x = torch.empty((B, 2 * N * N * s), dtype=torch.float64, device=dev)
    li = np.zeros(B * max(steps, 3), np.int32)
    torch.cuda.synchronize()

    def run(iters):
        p = AdmmParams(0.05, iters, 1e-4, 100, 0, 1 if multi else 0, 0.01, 0)
        eng._check(eng.L.qmri_pnp_admm_dev(eng.h, B, C.c_void_p(d_y.data_ptr()), C.byref(p), None, None, C.c_void_p(d_x.data_ptr()), None,
                                           li.ctypes.data_as(C.POINTER(C.c_int32))))
        eng.synchronize()

    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    health = eng.health()                                          # (of the TIMED run: counters, and the wall clock of its one qmri_pnp_admm_dev call)
    eng.profile_get(reset=True)
    eng.profile_enable(1)
    run(steps)
    p1 = eng.profile_get(reset=True)
    eng.profile_enable(2)
    run(steps)
    p2 = eng.profile_get(reset=True)
    eng.profile_enable(0)
    it = max(p1["admm_iters"], 1)
    ns = int(np.unique(k).size)
    m = int(fp[-1])
    one_launch = p2["n_lsqr_launches"] > 0 and p2["n_lsqr_launches"] <= p2["admm_iters"]      # (one k_ks_persist launch per x-update)
    out = {"workload": name, "value": round(steps * B / dt, 3), "unit": "slice-iterations/s (ADMM iterations x slices advanced together)",
           "ms_per_admm_iteration": round(dt / steps * 1e3, 4), "ms_per_slice_iteration": round(dt / steps / B * 1e3, 4), "steps": steps, "slices_per_launch": B,
           "T": T, "m": m, "sampled_k_locations": ns, "mask": mask, "denoiser": "11-channel multi-level UNetRes (noise-map channel)" if multi else "10-channel UNetRes",
           "stage_ms_per_iter": {"xupdate": round(p1["ms_xupdate"] / it, 4), "denoiser": round(p1["ms_denoiser"] / it, 4),
                                 "elementwise": round(p1["ms_elementwise"] / it, 4)},
           "xupdate_share_of_iteration": round(p1["ms_xupdate"] / max(p1["ms_xupdate"] + p1["ms_denoiser"] + p1["ms_elementwise"], 1e-9), 3),
           "xupdate": xupdate_roofline({**p2, "admm_iters": p2["admm_iters"], "lsqr_iters": p2["lsqr_iters"]}, ns, s, m, B, one_launch, f"{mask}_T{T}_B{B}"),
           "health": health,
           "reference_switches": "main_recon_tsmis_FFT.m:41-49 (cut, subsampling_pattern), :75-83 (denoiser_type)"}
    out["xupdate"]["us_per_lsqr_iteration_incl_fixed_launches"] = round(p1["ms_xupdate"] / it / max(p1["lsqr_iters"] / it / B, 1e-9) * 1e3, 2)
    return out


def multicoil_config(torch, local_rank, steps, ncoil=8):
    """BASELINE configs[4]'s multi-coil part, a labelled EXTENSION (the reference is single-coil, README.md:63: no counterpart, parity unpinned): cut0
    (T = 1000), `ncoil` coils, PnP-ADMM over the SENSE operator (image-domain LSQR, mc_kernels.hip) with the 10-channel network, one slice, through the
    host-array entry point (copies included: the entry point has no device-resident form).  Not a tuned path: the LSQR scalars travel through the host."""
    from qmri_pnp_recon_poc_amd import engine as E, synth
    N, T, s = 224, 1000, 10
    dic = cached_dictionary(synth, T, 32, 16, s)
    fp, k = E.build_spiral(N, 771, T)
    hh, ww = np.meshgrid(np.linspace(-1, 1, N), np.linspace(-1, 1, N), indexing="ij")
    maps = np.stack([np.exp(-((hh - np.cos(a)) ** 2 + (ww - np.sin(a)) ** 2)) * np.exp(1j * (a + hh * ww)) for a in np.linspace(0, 2 * np.pi, ncoil, endpoint=False)], axis=2)
    maps = maps / np.sqrt(np.sum(np.abs(maps) ** 2, axis=2, keepdims=True))
    eng = E.Engine(local_rank)
    try:
        eng.set_operator(N, N, dic["V"], fp, k, max_batch=4)
        eng.set_coils(maps)
        eng.set_denoiser(synth.structured_weights(seed=2, eps=0.02), N, N)
        X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)
        y_mc = np.stack([synth.awgn_measured(col, 30.0, seed=j) for j, col in enumerate(eng.forward_mc(X0).T)], axis=1)
        eng.pnp_admm_mc(y_mc, iters=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, li = eng.pnp_admm_mc(y_mc, iters=steps)
        dt = time.perf_counter() - t0
        return {"workload": f"EXTENSION (no reference counterpart, parity unpinned): cut0 (T = 1000, m = {eng.m} per coil) x {ncoil} coils, PnP-ADMM over the multi-coil "
                            "operator (image-domain LSQR) + 10-channel UNetRes, one slice, host arrays in and out",
                "value": round(steps / dt, 3), "unit": "ADMM iters/s", "steps": steps, "ms_per_admm_iteration": round(dt / steps * 1e3, 2),
                "lsqr_iters_mean": round(float(np.mean(li)), 2), "ms_per_lsqr_iteration": round(dt / max(float(np.sum(li)), 1.0) * 1e3, 3),
                "rel_err_to_ground_truth": round(float(np.linalg.norm((x - X0).ravel()) / np.linalg.norm(X0.ravel())), 4), "health": eng.health()}
    finally:
        eng.close()


def slices_phase(args, rank, local_rank, world, dev, torch, dist, total, batch, iters, warmup_iters, dict_grid=(384, 256)):
    """north_star metric 2: a FIXED batch of `total` slices (120 = 8 subjects x 15, BASELINE configs[3]) sharded over the ranks in contiguous
    blocks; each rank walks its block in launches of `batch` slices (k_conv6p, batched LSQR), every slice = `iters` PnP-ADMM iterations + the
    dictionary match at K = dict_grid product.  Inputs resident in HBM, barrier + synchronize on both sides, max over ranks.  Returns the
    object that goes into the JSON line (rank 0: with the roofline of the batched conv kernel and of the dictionary match)."""
    import ctypes as C
    from qmri_pnp_recon_poc_amd import engine as E, synth
    from qmri_pnp_recon_poc_amd._lib import AdmmParams
    from qmri_pnp_recon_poc_amd.batch import shard_slices
    N, T, s, S = 224, 200, 10, 771
    B = batch
    dic = cached_dictionary(synth, T, dict_grid[0], dict_grid[1], s)
    fp, k = E.build_spiral(N, S, T)
    weights = synth.structured_weights(seed=2, eps=0.02)
    eng = E.Engine(local_rank)
    eng.set_operator(N, N, dic["V"], fp, k, max_batch=B)
    eng.set_denoiser(weights, N, N, max_batch=B)
    eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    stream = torch.cuda.Stream()                               # a torch stream the engine launches on: torch events then see its kernels
    eng.set_stream(stream.cuda_stream)
    mine = shard_slices(total, world, rank)
    nsl = len(mine)
    n, m = N * N * s, eng.m

    def make_y(seed):
        X0 = synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=seed), dic)
        return synth.awgn_measured(eng.forward(X0), 30.0, seed=seed)

    ys = np.stack([make_y(i) for i in mine]) if nsl else np.zeros((0, m), np.complex128)
    d_y = torch.from_numpy(np.ascontiguousarray(ys).view(np.float64)).to(dev)
    # results of a launch (x and the maps of its B slices) are copied back to pinned host memory on a copy stream while the next launch
    # computes ("results copied back per slice"; what qmri_recon_batch does): two sets of buffers in rotation
    cstream = torch.cuda.Stream()
    d_x = [torch.empty((B, 2 * n), dtype=torch.float64, device=dev) for _ in range(2)]
    d_q = [torch.empty((B, N * N * 2), dtype=torch.float32, device=dev) for _ in range(2)]
    d_pd = [torch.empty((B, N * N * 2), dtype=torch.float32, device=dev) for _ in range(2)]
    h_x = [torch.empty((B, 2 * n), dtype=torch.float64, pin_memory=True) for _ in range(2)]
    h_q = [torch.empty((B, N * N * 2), dtype=torch.float32, pin_memory=True) for _ in range(2)]
    h_pd = [torch.empty((B, N * N * 2), dtype=torch.float32, pin_memory=True) for _ in range(2)]
    torch.cuda.synchronize()

    def params(it):
        return AdmmParams(0.05, it, 1e-4, 100, 0 if args.solver == "lsqr" else 1, 0, 0.01, 0)

    def match(j, i):
        eng._check(eng.L.qmri_dict_match_dev(eng.h, C.c_void_p(d_x[j].data_ptr() + i * n * 16), N * N, C.c_void_p(d_q[j].data_ptr() + i * N * N * 8),
                                             C.c_void_p(d_pd[j].data_ptr() + i * N * N * 8), None, None))

    copied = [None, None]                                      # event on the copy stream behind the copies out of buffer set j

    batch_s = []                                               # host clock of every launch of the last run() (qmri_pnp_admm_dev returns when its slices are done)
    launch_health = []                                         # ... and the library's own account of it (qmri_get_health: wall clock, stage marks, counters)

    def run(count, it):
        p = params(it)
        batch_s.clear()
        launch_health.clear()
        for bi, s0 in enumerate(range(0, count, B)):
            tb = time.perf_counter()
            j, cnt = bi & 1, min(B, count - s0)
            if copied[j] is not None:                              # set j is written again: its copies (two launches ago) must be over, as in recon_worker (api_net.cpp)
                stream.wait_event(copied[j])
                copied[j].synchronize()                            # (... and the pinned results consumed: here they are simply dropped)
            eng._check(eng.L.qmri_pnp_admm_dev(eng.h, cnt, C.c_void_p(d_y.data_ptr() + s0 * m * 16), C.byref(p), None, None,
                                               C.c_void_p(d_x[j].data_ptr()), None, None))
            batch_s.append(round(time.perf_counter() - tb, 4))
            launch_health.append(eng.health())
            for i in range(cnt):
                match(j, i)
            ev = torch.cuda.Event()
            ev.record(stream)
            cstream.wait_event(ev)
            with torch.cuda.stream(cstream):
                h_x[j][:cnt].copy_(d_x[j][:cnt], non_blocking=True)
                h_q[j][:cnt].copy_(d_q[j][:cnt], non_blocking=True)
                h_pd[j][:cnt].copy_(d_pd[j][:cnt], non_blocking=True)
                copied[j] = torch.cuda.Event()
                copied[j].record(cstream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    if nsl:
        run(min(B, nsl), max(warmup_iters, 1))
    barrier()
    # stage MARKS (profile level 3): event records at the stage boundaries of every ADMM iteration, read after each launch's own final synchronisation --
    # the stage split of the TIMED launches without a wait inside them (three records per 11.5-ms iteration: ~0.1 %)
    eng.profile_get(reset=True)
    eng.profile_enable(3)
    t0 = time.perf_counter()
    run(nsl, iters)
    barrier()
    dt = time.perf_counter() - t0
    eng.profile_enable(0)
    dt_own = dt
    h_end = eng.health()
    # every rank's own seconds and counters travel to rank 0: at N = 8 the value is the max over ranks, and a slow rank must be nameable
    mine_rec = [dt_own, float(h_end["denoiser_fallbacks"]), float(h_end["resident_tile_timeouts"]), float(h_end["lsqr_one_launch_timeouts"]),
                float(h_end["repeated_calls"]), float(max(batch_s) if batch_s else 0.0), float(nsl)]
    per_rank = [mine_rec]
    if world > 1:
        gdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=gdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine_t = torch.tensor(mine_rec, dtype=torch.float64, device=gdev)
        gathered = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(gathered, mine_t)
        per_rank = [g.cpu().tolist() for g in gathered]
    out = None
    if rank == 0:
        K = int(dic["K"])
        out = {"metric": f"slices/sec ({total}-slice synthetic batch: {iters} ADMM iterations + dictionary match per slice)", "value": round(total / dt, 4),
               "unit": "slices/s", "n_gpus": world, "scaling": "strong", "seconds": round(dt, 3), "ms_per_slice": round(dt / max(nsl, 1) * 1e3, 3),
               "slices_on_rank0": nsl, "total_slices": total, "batch": B, "admm_iters_per_slice": iters, "dict_K": K,
               "launch_seconds_rank0": list(batch_s),        # (one entry per launch of `batch` slices: a stalled launch would show here)
               "rank_seconds": {"min": round(min(r[0] for r in per_rank), 3), "median": round(float(np.median([r[0] for r in per_rank])), 3),
                                "max": round(max(r[0] for r in per_rank), 3), "slowest_rank": int(np.argmax([r[0] for r in per_rank])),
                                "per_rank": [round(r[0], 3) for r in per_rank], "slowest_launch_s_per_rank": [round(r[5], 4) for r in per_rank],
                                "slices_per_rank": [int(r[6]) for r in per_rank]},
               "health": {**{k_: v for k_, v in h_end.items() if not k_.startswith("last_call") and k_ != "set_denoiser_ms"},
                          "all_ranks": {"denoiser_fallbacks": int(sum(r[1] for r in per_rank)), "resident_tile_timeouts": int(sum(r[2] for r in per_rank)),
                                        "lsqr_one_launch_timeouts": int(sum(r[3] for r in per_rank)), "repeated_calls": int(sum(r[4] for r in per_rank))},
                          "what": "qmri_get_health after the timed launches: every fast path that can give up and repeat its work on the slower path counts it here "
                                  "(zero on a clean run); a slow launch with zero counters was slow for a reason outside the library"},
               "sharding": "fixed total, contiguous blocks (batch.shard_slices), no collective in the data path",
               "results": "x (8 MB) and the T1 / T2 / PD maps of every slice copied to pinned host memory inside the timed region (copy stream, overlapped)",
               "workload": f"cut3 {total}-slice batch over {world} GPU(s), {B} slices advanced together, spiral mask, PnP-ADMM + UNetRes + dictionary match K={K}"}
        if launch_health:
            # the slowest launch of rank 0, stage by stage (device time between the stage marks) beside its host wall clock
            worst = int(np.argmax(batch_s))
            lh = launch_health[worst]
            st_sum = sum(lh["last_call_stage_ms"].values())
            out["slowest_launch_rank0"] = {"launch": worst, "host_seconds": batch_s[worst], "library_wall_ms": lh["last_call_wall_ms"],
                                           "stage_ms": lh["last_call_stage_ms"], "stages_sum_ms": round(st_sum, 3),
                                           "outside_the_stages_ms": round(lh["last_call_wall_ms"] - st_sum, 3),
                                           "median_launch_stage_ms": {k_: round(float(np.median([h["last_call_stage_ms"][k_] for h in launch_health])), 3)
                                                                      for k_ in lh["last_call_stage_ms"]},
                                           "what": "stage marks (profile level 3) of every timed launch; a launch that is slow INSIDE a stage names the stage, one that "
                                                   "is slow outside them (host, queue, another process on the device) shows in outside_the_stages_ms"}
        if not args.no_roofline and nsl:
            # the batched conv kernel (k_conv6p): live dispatch durations of three forwards of B slices
            eng.profile_get(reset=True)
            eng.profile_enable(2)
            d_in = torch.rand(B * s * N * N, dtype=torch.float32, device=dev)
            d_out = torch.empty(B * s * N * N, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):
                eng._check(eng.L.qmri_net_forward_dev(eng.h, C.c_void_p(d_in.data_ptr()), B, C.c_void_p(d_out.data_ptr())))
            eng.synchronize()
            pr = eng.profile_get(reset=True)
            eng.profile_enable(0)
            traffic, tsrc = load_traffic(B)
            out["roofline"] = conv_roofline(pr, B, f"3x3 convolution launches of a {B}-slice forward pass: k_conv6p (persistent implicit GEMM, every 64 -> 64 ... 512 -> 512 "
                                                   f"layer) + the head and tail on k_conv6, on {SCHEME_TEXT}", traffic, tsrc)
            # the dictionary match of one reconstructed slice (d_x[0][0]: the first slice of an earlier launch), HIP events on the engine's stream
            with torch.cuda.stream(stream):
                for _ in range(3):
                    match(0, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(20):
                    match(0, 0)
                e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            pairs = ((K + 31) // 32) * ((N * N + 31) // 32)
            exec_f16 = pairs * 6 * 32 * 32 * 16 * 2               # filter: 6 x v_mfma_f32_32x32x16_f16 per (32-atom, 32-pixel) tile pair
            alg = 2 * 2 * N * N * K * s
            out["dict_match"] = {"ms_per_slice": round(ms, 4), "K": K,
                                 "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": BF16_MFMA_PEAK_TFLOPS, "achieved": round(exec_f16 / (ms * 1e-3) / 1e12, 1),
                                              "frac": round(exec_f16 / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), "executed_flop": exec_f16, "traffic": None,
                                              "note": "against the pipe it uses: executed f16-MFMA flops of the filter (K-padding 10 -> 16 included; the exact f32 products "
                                                      "of the ~1 % listed tiles not counted) / time; maps bit-identical to the oracle's",
                                              "algorithmic_f32_tflops": round(alg / (ms * 1e-3) / 1e12, 1),
                                              "algorithmic_over_f32_mfma_peak": round(alg / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3)}}
    eng.close()
    return out


def cold_start(torch, local_rank):
    """What ONE reconstruction costs from nothing, as the reference runs it (one slice per run, the network loaded every run:
    main_recon_tsmis_FFT.m:37-38,138-152,285-317): qmri_create, the three plans, then y (host) -> 100 PnP-ADMM iterations -> dictionary match at
    K = 98 304 -> x and the maps on the host.  Host wall clock, first use of the library in this process (code objects are loaded on the way)."""
    from qmri_pnp_recon_poc_amd import engine as E, synth
    N, T, s, S = 224, 200, 10, 771
    dic = cached_dictionary(synth, T, 384, 256, s)
    fp, k = E.build_spiral(N, S, T)
    w = synth.structured_weights(seed=2, eps=0.02)
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    eng = E.Engine(local_rank)
    try:
        t.append(time.perf_counter())
        eng.set_operator(N, N, dic["V"], fp, k)
        t.append(time.perf_counter())
        eng.set_denoiser(w, N, N)
        t.append(time.perf_counter())
        eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
        t.append(time.perf_counter())
        split = eng.health()["set_denoiser_ms"]
        y = synth.awgn_measured(eng.forward(synth.synthesize_tsmi(synth.make_phantom_qmaps(N, seed=0), dic)), 30.0, seed=0)   # (input synthesis: not part of a reconstruction)
        t_in = time.perf_counter()
        x, _, li = eng.pnp_admm(y, iters=100)
        t_rec = time.perf_counter()
        maps = eng.dict_match(x)
        t_out = time.perf_counter()
        setup_s = t[4] - t[0]
        recon_s = t_out - t_in
        # the same reconstruction again on the warm context: what the first one paid for being first
        t2 = time.perf_counter()
        eng.pnp_admm(y, iters=100)
        eng.dict_match(x)
        warm_s = time.perf_counter() - t2
        return {"what": "one slice from nothing on rank 0, host wall clock, first use of the library in the process: create + plans, then y on the host -> 100 PnP-ADMM "
                        "iterations -> dictionary match (K = 98 304) -> x and T1 / T2 / PD maps on the host (the reference's unit of work: main_recon_tsmis_FFT.m:37-38, "
                        "138-152, 285-317)",
                "setup_ms": {"qmri_create": round((t[1] - t[0]) * 1e3, 2), "qmri_set_operator": round((t[2] - t[1]) * 1e3, 2),
                             "qmri_set_denoiser": round((t[3] - t[2]) * 1e3, 2), "qmri_set_denoiser_split": split,
                             "qmri_set_dictionary": round((t[4] - t[3]) * 1e3, 2), "total": round(setup_s * 1e3, 2)},
                "first_reconstruction_s": round(recon_s, 4), "of_which_admm_s": round(t_rec - t_in, 4), "of_which_match_and_maps_s": round(t_out - t_rec, 4),
                "time_to_first_slice_s": round(setup_s + recon_s, 4), "same_reconstruction_again_s": round(warm_s, 4),
                "setup_over_reconstruction": round(setup_s / max(warm_s, 1e-9), 2), "dict_K": int(dic["K"]), "lsqr_iters_mean": float(np.mean(li)),
                "atoms_matched": int(np.unique(maps["dm"]).size)}
    finally:
        eng.close()


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus = {world}", file=sys.stderr)
    import torch
    import torch.distributed as dist
    if args.one_device:
        local_rank = 0
    if args.plumbing_only:
        # the rank plumbing alone (CPU-runnable): rendezvous, barrier, max over ranks, one line from rank 0
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        from qmri_pnp_recon_poc_amd.batch import shard_slices
        if args.workload == "admm" and not args.no_slices:          # the default line's `slices` object shards args.slices_total the same way
            args.total_slices = args.slices_total
        mine = shard_slices(args.total_slices, world, rank) if args.total_slices > 0 else list(range(rank * args.slices_per_gpu, (rank + 1) * args.slices_per_gpu))
        cover = torch.zeros(max(args.total_slices, args.slices_per_gpu * world), dtype=torch.int64)
        cover[mine] += 1                                          # every slice id must be owned by exactly one rank
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            dist.all_reduce(cover, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"metric": "ADMM iters/sec (224x224x10 TSMI, spiral mask)", "value": None, "unit": "ADMM iters/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "plumbing_only": True, "max_rank_seconds": round(dt, 4),
                              "slices_on_rank0": len(mine), "slices_owned_once": int((cover == 1).sum()), "slices_total": int(cover.numel()),
                              "batches_on_rank0": -(-len(mine) // max(args.batch, 1)), "scaling": "strong" if args.total_slices > 0 else "weak"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
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

    cold = None
    if rank == 0 and args.workload == "admm" and not args.no_cold_start:
        try:                                                        # (first: the library has not been used in this process yet)
            cold = cold_start(torch, local_rank)
        except Exception as exc:                                    # noqa: BLE001
            cold = {"error": f"{type(exc).__name__}: {exc}"[:400]}
            print(f"bench.py: cold-start phase failed: {exc}", file=sys.stderr)

    N, T, s, S = 224, 200, 10, 771
    B = args.batch if args.workload == "slices" else 1
    dic = synth.make_dictionary(T=T, n_t1=args.dict_k[0] if args.workload == "slices" else 128,
                                n_t2=args.dict_k[1] if args.workload == "slices" else 64, s=s)
    fp, k = E.build_spiral(N, S, T)
    weights = synth.structured_weights(seed=2, eps=0.02)
    eng = E.Engine(local_rank)
    eng.set_operator(N, N, dic["V"], fp, k, max_batch=B)
    eng.set_denoiser(weights, N, N, max_batch=B)
    eng.set_dictionary(dic["D"], dic["normD"], dic["lut"])
    # torch's default stream is the NULL stream: the engine then keeps its own non-blocking stream, which is NOT ordered with
    # torch's.  Every hand-over between torch work and engine work below is therefore bracketed by a device synchronize.
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    def make_y(seed, want_gt=False):
        q = synth.make_phantom_qmaps(N, seed=seed)
        X0 = synth.synthesize_tsmi(q, dic)
        y = synth.awgn_measured(eng.forward(X0), 30.0, seed=seed)
        return (y, X0) if want_gt else y

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()

    def admm_params(iters, want_diag=0):
        return AdmmParams(0.05, iters, 1e-4, 100, 0 if args.solver == "lsqr" else 1, 0, 0.01, want_diag)

    def x_from_device(t):
        return t.cpu().numpy().view(np.complex128).reshape((N, N, s), order="F")

    n = N * N * s
    if args.workload == "admm":
        y, X_gt = make_y(rank, want_gt=True)
        y_parity = y
        d_y = torch.from_numpy(np.ascontiguousarray(y).view(np.float64)).to(dev)
        d_gt = torch.from_numpy(np.ascontiguousarray(X_gt.astype(np.complex128).ravel(order="F")).view(np.float64)).to(dev)
        d_x = torch.empty(2 * n, dtype=torch.float64, device=dev)
        li = np.zeros(max(args.steps, args.warmup, 1), np.int32)
        diag = np.zeros((max(args.steps, args.warmup, 1), 2), np.float64)
        torch.cuda.synchronize()

        def run(iters, want_diag=0):
            p = admm_params(iters, want_diag)
            st = eng.L.qmri_pnp_admm_dev(eng.h, 1, C.c_void_p(d_y.data_ptr()), C.byref(p), None,
                                         C.c_void_p(d_gt.data_ptr()) if want_diag else None, C.c_void_p(d_x.data_ptr()),
                                         diag.ctypes.data_as(C.POINTER(C.c_double)) if want_diag else None, li.ctypes.data_as(C.POINTER(C.c_int32)))
            eng._check(st)

        def gpu_x_after(iters):
            run(iters)
            eng.synchronize()
            return x_from_device(d_x)

        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        lsqr_mean = float(li[: args.steps].mean()) if args.steps else 0.0
        unit_count = args.steps                                  # ADMM iterations per rank
        total_units = args.steps * world
        h0 = eng.health()                                        # (of the timed call: which fast paths ran, whether anything was repeated)
        rec = [dt, float(h0["denoiser_fallbacks"]), float(h0["resident_tile_timeouts"]), float(h0["lsqr_one_launch_timeouts"]), float(h0["repeated_calls"])]
        per_rank = [rec]
        if world > 1:
            gdev = dev if args.backend == "nccl" else "cpu"
            mine_t = torch.tensor(rec, dtype=torch.float64, device=gdev)
            gathered = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(gathered, mine_t)
            per_rank = [g.cpu().tolist() for g in gathered]
        result_extra = {"lsqr_iters_mean": lsqr_mean,
                        "health": {**{k_: v for k_, v in h0.items() if k_ != "set_denoiser_ms"},
                                   "all_ranks": {"denoiser_fallbacks": int(sum(r[1] for r in per_rank)), "resident_tile_timeouts": int(sum(r[2] for r in per_rank)),
                                                 "lsqr_one_launch_timeouts": int(sum(r[3] for r in per_rank)), "repeated_calls": int(sum(r[4] for r in per_rank))},
                                   "rank_seconds": [round(r[0], 4) for r in per_rank]}}
        # the same K steps with the reference's two per-iteration diagnostics on (PnP_ADMM.m:106-109: |y - A x| / |y| and
        # |x_gt - x| / |x_gt|, which the reference always computes and the CPU baseline below runs): a second timed region,
        # reported beside `value` (the headline keeps the diagnostics off: they are print-outs, not part of the iteration)
        barrier()
        t0 = time.perf_counter()
        run(args.steps, want_diag=1)
        barrier()
        dt_diag = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_diag], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_diag = float(t.item())
        result_extra["with_diagnostics"] = {"value": round(args.steps * world / dt_diag, 4), "unit": "ADMM iters/s",
                                            "ms_per_step": round(dt_diag / max(args.steps, 1) * 1e3, 4),
                                            "what": "same steps with want_diag = 1 (PnP_ADMM.m:106-109: data-fidelity and ground-truth errors per iteration)",
                                            "last_data_fidelity_rel": float(diag[max(args.steps, 1) - 1, 0]), "last_gt_rel_err": float(diag[max(args.steps, 1) - 1, 1])}
    else:
        from qmri_pnp_recon_poc_amd.batch import shard_slices
        if args.total_slices > 0:
            mine = shard_slices(args.total_slices, world, rank)   # fixed total: contiguous block of this rank
            total_units = args.total_slices
        else:
            mine = list(range(rank * args.slices_per_gpu, (rank + 1) * args.slices_per_gpu))
            total_units = args.slices_per_gpu * world
        nsl = len(mine)
        ys = np.stack([make_y(i) for i in mine]) if nsl else np.zeros((0, eng.m), np.complex128)
        y_parity = ys[0]
        d_y = torch.from_numpy(np.ascontiguousarray(ys).view(np.float64)).to(dev)
        d_x = torch.empty((B, 2 * n), dtype=torch.float64, device=dev)
        d_q = torch.empty(N * N * 2, dtype=torch.float32, device=dev)
        d_pd = torch.empty(N * N * 2, dtype=torch.float32, device=dev)
        m = eng.m
        torch.cuda.synchronize()

        def run_slices(count, iters):
            p = admm_params(iters)
            for s0 in range(0, count, B):
                cnt = min(B, count - s0)
                yptr = d_y.data_ptr() + s0 * m * 16
                eng._check(eng.L.qmri_pnp_admm_dev(eng.h, cnt, C.c_void_p(yptr), C.byref(p), None, None, C.c_void_p(d_x.data_ptr()), None, None))
                for i in range(cnt):
                    eng._check(eng.L.qmri_dict_match_dev(eng.h, C.c_void_p(d_x.data_ptr() + i * n * 16), N * N, C.c_void_p(d_q.data_ptr()),
                                                         C.c_void_p(d_pd.data_ptr()), None, None))

        def gpu_x_after(iters):
            p = admm_params(iters)
            eng._check(eng.L.qmri_pnp_admm_dev(eng.h, 1, C.c_void_p(d_y.data_ptr()), C.byref(p), None, None, C.c_void_p(d_x.data_ptr()), None, None))
            eng.synchronize()
            return x_from_device(d_x[0])

        if nsl:
            run_slices(min(B, nsl), max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        run_slices(nsl, args.steps)
        barrier()
        dt = time.perf_counter() - t0
        unit_count = nsl
        result_extra = {"admm_iters_per_slice": args.steps, "dict_K": int(dic["K"]), "slices_on_rank0": nsl, "batch": B,
                        "total_slices": int(total_units), "sharding": "fixed total, contiguous blocks" if args.total_slices > 0 else "fixed per GPU"}

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
        torch.cuda.synchronize()                                  # d_in is filled on torch's stream, read on the engine's
        for _ in range(3):
            eng._check(eng.L.qmri_net_forward_dev(eng.h, C.c_void_p(d_in.data_ptr()), B, C.c_void_p(d_out.data_ptr())))
        eng.synchronize()
        pr = eng.profile_get(reset=True)
        eng.profile_enable(0)
        if pr["n_conv3x3"] > 0:
            f32_path = debug_knob("conv_f32", 0) > 0
            if f32_path:
                t3 = pr["ms_conv3x3"] * 1e-3
                ach = pr["flop_conv3x3"] / t3 / 1e12
                roof = {"kernel": "k_conv<3x3> (implicit-GEMM conv3x3 on v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                        "achieved": round(ach, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": None, "avg_launch_us": round(t3 / pr["n_conv3x3"] * 1e6, 2), "launches_timed": int(pr["n_conv3x3"]),
                        "flop_per_launch": pr["flop_conv3x3"] / pr["n_conv3x3"]}
            else:
                traffic, tsrc = load_traffic(B)
                resident = debug_knob("conv_resident", 1) != 0 and B == 1
                roof = conv_roofline(pr, B, (f"3x3 convolution launches of a forward pass on {SCHEME_TEXT}: k_conv6 (one launch per layer; the 28 x 28 level's split-K "
                                             "layers with their reduce kernel) + k_conv6r (the 16 full-resolution ResBlock layers with the head, the tail and the level's "
                                             "down-sampling convolution as TWO resident-tile launches)") if resident else
                                     f"3x3 convolution launches of a forward pass (k_conv6 / k_conv6p, split-K layers with their reduce kernel) on {SCHEME_TEXT}", traffic, tsrc)
        # stage split of one more run of the SAME workload (profile level 1 synchronises per stage; not part of the timed region): all
        # args.steps iterations, because the x-update is not uniform over a reconstruction -- LSQR needs 16 iterations in the first
        # x-updates and 5-8 in the later ones (lsqr_iters_mean)
        if args.workload == "admm":
            eng.profile_enable(1)
            run(max(args.steps, 1))
            pr = eng.profile_get(reset=True)
            eng.profile_enable(0)
            it = max(pr["admm_iters"], 1)
            result_extra["stage_ms_per_iter"] = {"xupdate": round(pr["ms_xupdate"] / it, 4), "denoiser": round(pr["ms_denoiser"] / it, 4),
                                                 "elementwise": round(pr["ms_elementwise"] / it, 4), "over_admm_iters": int(it)}
            if roof is not None and pr["ms_denoiser"] > 0:        # the denoiser stage of the ADMM loop itself (stage timer), beside the level-2 figure
                ex = SPLIT_PRODUCTS * DENOISER_FLOP
                roof["whole_denoiser_in_admm_loop"] = {"ms": round(pr["ms_denoiser"] / it, 4), "achieved": round(ex / (pr["ms_denoiser"] / it * 1e-3) / 1e12, 3),
                                                       "frac": round(ex / (pr["ms_denoiser"] / it * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
            # the x-update against the HBM roofline: one more run of the same steps at profile level 2 (the LSQR kernels' own dispatch timestamps)
            eng.profile_enable(2)
            run(max(args.steps, 1))
            p2 = eng.profile_get(reset=True)
            eng.profile_enable(0)
            ns_k = int(np.unique(k).size)
            result_extra["xupdate"] = xupdate_roofline(p2, ns_k, s, int(fp[-1]), 1, p2["n_lsqr_launches"] <= p2["admm_iters"], f"spiral_T{T}_B1")
            result_extra["xupdate"]["us_per_lsqr_iteration_incl_fixed_launches"] = round(pr["ms_xupdate"] / it / max(pr["lsqr_iters"] / it, 1e-9) * 1e3, 2)

    # ---- CPU baseline + parity: the oracle on this box's host cores, bounded sample -----------------------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        O.build()
        fo, ko = O.spiral_mask(N, S, T)
        op = O.Operator(N, N, dic["V"], fo, ko)
        net = O.Net(weights)
        cores = O.num_threads()

        def oracle_admm(iters):
            t0 = time.perf_counter()
            xo, _, lio = O.pnp_admm(op, net, y_parity, gamma=0.05, iters=iters, cg_tol=1e-4, cg_maxit=100, solver=args.solver, want_diag=True)
            return xo, lio, time.perf_counter() - t0, O.admm_stage_seconds()

        _, _, t_probe, _ = oracle_admm(2)                         # probe (also warms caches and the OpenMP pool)
        per_iter = t_probe / 2
        n_cpu = args.cpu_iters if args.cpu_iters > 0 else int(max(3, min(args.steps, args.cpu_budget_s / max(per_iter, 1e-6))))
        xo, lio, tc, stages = oracle_admm(n_cpu)
        t0 = time.perf_counter()
        maps_o = O.dict_match(xo, dic["D"], dic["normD"], dic["lut"])
        t_match = time.perf_counter() - t0
        O.set_num_threads(1)
        _, _, t1, _ = oracle_admm(1)
        O.set_num_threads(cores)
        stage_ms = {k_: round(v / n_cpu * 1e3, 2) for k_, v in stages.items()}
        common = {"cores": cores, "kind": "port", "cpu_model": cpu_model(), "one_thread_value": None,
                  "stage_ms_per_iter": stage_ms, "dict_match_s": round(t_match, 3), "dict_K": int(dic["K"]), "diagnostics": "on (PnP_ADMM.m:106-109)"}
        if args.workload == "admm" and not args.no_slices:
            # like for like with the GPU `slices` phase (K = 98 304 there, 8 192 in the parity leg above): ONE oracle match of this slice at that K
            dic_big = cached_dictionary(synth, T, args.dict_k[0], args.dict_k[1], s)
            t0 = time.perf_counter()
            O.dict_match(xo, dic_big["D"], dic_big["normD"], dic_big["lut"])
            common["dict_match_s_at_slices_K"] = round(time.perf_counter() - t0, 3)
            common["slices_K"] = int(dic_big["K"])
            common["slice_s_extrapolated_at_slices_K"] = round(tc / n_cpu * args.slices_iters + common["dict_match_s_at_slices_K"], 2)
            del dic_big
        if args.workload == "admm":
            cpu = {"value": round(n_cpu / tc, 4), "unit": "ADMM iters/s",
                   "sample": f"first {n_cpu} PnP-ADMM iterations of the same slice (LSQR tol 1e-4 fp64 + UNetRes fp32, diagnostics on), {tc:.1f} s; "
                             f"one thread: 1 iteration, {t1:.1f} s", **common}
            cpu["one_thread_value"] = round(1.0 / t1, 4)
        else:
            per_slice = tc / n_cpu * args.steps + t_match
            cpu = {"value": round(1.0 / per_slice, 6), "unit": "slices/s",
                   "sample": f"{n_cpu} PnP-ADMM iterations of slice 0 ({tc:.1f} s) extrapolated to {args.steps} per slice + its dictionary match "
                             f"({t_match:.2f} s); one thread: 1 iteration, {t1:.1f} s", **common}
            cpu["one_thread_value"] = round(1.0 / (t1 * args.steps + t_match * cores), 6)
        # parity on the slice that was timed, after the same number of iterations the oracle ran
        xg = gpu_x_after(n_cpu)
        maps_g = eng.dict_match(xg)
        parity = parity_numbers(xg, xo, maps_g, maps_o, n_cpu, dic)
        # the network itself with weights under which every level matters (the bench network above is ADMM-stable and therefore
        # numerically head + tail: DESIGN.md section 7): one forward of random_weights(seed=1, gain=0.7), GPU against the oracle
        wsens = synth.random_weights(seed=1, gain=0.7)
        e2 = E.Engine(local_rank)
        e2.set_denoiser(wsens, N, N)
        xin = synth.golden224_input(10).transpose(1, 2, 0).astype(np.float64)
        yg, yo_ = e2.denoise(xin), O.Net(wsens).denoise(xin)
        parity["net_rel_l2_random_weights"] = float(np.linalg.norm((yg - yo_).ravel()) / np.linalg.norm(yo_.ravel()))
        parity["net_random_weights_scheme"] = list(e2.denoiser_scheme())
        # ... and TIMED with them: the bench network's interior layers sit at 1.5e-4 of its output, i.e. their f16 pieces barely
        # toggle; a trained DRUNet's activations do.  Same call path and the same input for both weight sets: 10 forward passes at profile level 2
        # (ms_net_forward: the whole pass between two stream events).
        if args.workload == "admm":
            d_rin = torch.from_numpy(np.ascontiguousarray(synth.golden224_input(10).transpose(0, 2, 1))).to(dev)      # [c][w][h] fp32, the device layout
            d_rout = torch.empty(s * N * N, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()

            def forward_ms(e):
                for _ in range(3):
                    e._check(e.L.qmri_net_forward_dev(e.h, C.c_void_p(d_rin.data_ptr()), 1, C.c_void_p(d_rout.data_ptr())))
                e.profile_get(reset=True)
                e.profile_enable(2)
                per, acc = [], None
                for _ in range(10):                                 # one profile read per pass: the MEDIAN pass (a pass between two stream events also
                    e._check(e.L.qmri_net_forward_dev(e.h, C.c_void_p(d_rin.data_ptr()), 1, C.c_void_p(d_rout.data_ptr())))      # contains whatever the host
                    q = e.profile_get(reset=True)                   # did not enqueue in time; on a box with a busy host single passes took 4 - 9 ms)
                    per.append(q["ms_net_forward"] / max(q["n_net_forward"], 1))
                    acc = q if acc is None else {k_: acc[k_] + q[k_] for k_ in q}
                e.profile_enable(0)
                return float(np.median(per)), conv_roofline(acc, 1, ""), [round(v, 3) for v in per]

            ms_s, roof_s, per_s = forward_ms(eng)
            ms_r, roof_r, per_r = forward_ms(e2)
            ms_step = dt / max(args.steps, 1) * 1e3
            result_extra["denoiser_weights_timing"] = {
                "what": "one 224 x 224 forward pass, same input, same call path (qmri_net_forward_dev, profile level 2), with the bench network "
                        "(structured_weights(eps=0.02): ADMM-stable, interior layers at 1.5e-4 of the output) and with random_weights(seed=1, gain=0.7), under "
                        "which every layer's activations are of order one -- what a trained DRUNet resembles",
                "denoiser_ms_structured_weights": round(ms_s, 4), "denoiser_ms_random_weights": round(ms_r, 4), "statistic": "median of 10 passes",
                "passes_ms_structured_weights": per_s, "passes_ms_random_weights": per_r,
                "conv3x3_frac_structured_weights": roof_s["frac"] if roof_s else None, "conv3x3_frac_random_weights": roof_r["frac"] if roof_r else None,
                "random_over_structured": round(ms_r / max(ms_s, 1e-9), 4),
                "implied_admm_iters_per_s_with_random_weights": round(1e3 / max(ms_step - ms_s + ms_r, 1e-9) * world, 2),
                "scheme_random_weights": list(e2.denoiser_scheme())}
        e2.close()
        if args.workload == "admm" and n_cpu <= len(li):
            parity["lsqr_iteration_counts_identical"] = bool(np.array_equal(li[:n_cpu], lio[:n_cpu]))

    # ---- north_star's second metric in the same line: the fixed 120-slice batch over all ranks -----------------------------
    slices_obj = None
    if args.workload == "admm" and not args.no_slices and args.slices_total > 0:
        sb = args.slices_batch if args.slices_batch > 0 else (30 if args.slices_total // max(world, 1) >= 30 else 15)
        slices_obj = slices_phase(args, rank, local_rank, world, dev, torch, dist, args.slices_total, sb, args.slices_iters, 2)
    # ---- the other single-GPU configurations of BASELINE.json on the same line (rank 0; the other ranks wait at the final barrier) ----------------
    secondary = {}
    if args.workload == "admm" and not args.no_secondary and rank == 0:
        # (rank 0 only, no collective inside: a failure here -- say, no memory left on a shared box -- is reported in the object instead of costing
        #  the line its headline metric)
        for key, cfg_args in (("epi_batch15", ("BASELINE configs[2]: cut3, EPI mask (m = 134 400), 15 slices advanced together, "
                                               "PnP-ADMM + 11-channel multi-level UNetRes", 200, "epi", 15, True)),
                              ("cut0", ("BASELINE configs[4] as far as the reference goes: cut0 (T = 1000, m = 618 000), spiral mask, one "
                                        "slice, PnP-ADMM + 10-channel UNetRes", 1000, "spiral", 1, False))):
            try:
                secondary[key] = secondary_config(args, torch, dev, local_rank, *cfg_args, args.secondary_steps)
            except Exception as exc:                                # noqa: BLE001
                secondary[key] = {"workload": cfg_args[0], "value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
                print(f"bench.py: secondary configuration {key} failed: {exc}", file=sys.stderr)
        try:
            secondary["cut0_multicoil8"] = multicoil_config(torch, local_rank, max(2, min(args.secondary_steps, 5)))
        except Exception as exc:                                    # noqa: BLE001
            secondary["cut0_multicoil8"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"[:400]}
            print(f"bench.py: multi-coil configuration failed: {exc}", file=sys.stderr)
    if rank == 0:
        strong = args.workload == "slices" and args.total_slices > 0
        if args.workload == "admm":
            metric, unit = "ADMM iters/sec (224x224x10 TSMI, spiral mask)", "ADMM iters/s"
            cfg = {"workload": "cut3 224x224x10 single slice per GPU, spiral mask S=771 T=200, PnP-ADMM + 10-channel UNetRes (DRUNet) denoiser",
                   "solver": args.solver, "admm_iters": args.steps, "dc_dtype": "f64",
                   "denoiser_arith": "f32 results: all convs on " + SCHEME_TEXT,
                   "parallelism": f"slice-parallel x{world} (no collective)"}
            ms_per_step = dt / max(args.steps, 1) * 1e3
        else:
            metric = f"slices/sec ({total_units}-slice synthetic batch: {args.steps} ADMM iterations + dictionary match per slice)"
            unit = "slices/s"
            cfg = {"workload": (f"cut3 {total_units}-slice batch sharded over {world} GPU(s) in contiguous blocks, {B} slices advanced together" if strong else
                                f"cut3 {unit_count}-slice batch per GPU") + f", spiral mask, PnP-ADMM + UNetRes + dictionary match K={int(dic['K'])}",
                   "solver": args.solver, "admm_iters": args.steps, "dc_dtype": "f64",
                   "denoiser_arith": "f32 results: all convs on " + SCHEME_TEXT, "parallelism": f"slice-parallel x{world} (no collective)"}
            ms_per_step = dt / max(unit_count, 1) * 1e3              # per slice on one rank
        out = {"metric": metric, "value": round(total_units / dt, 4), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": cfg, "roofline": roof, "cpu_baseline": cpu, "parity": parity}
        out.update(result_extra)
        if cold is not None:
            out["cold_start"] = cold
        if slices_obj is not None:
            out["slices"] = slices_obj
        out.update(secondary)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    worker(args)


if __name__ == "__main__":
    main()

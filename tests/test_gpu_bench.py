"""GPU: bench.py end to end on the box -- the two-rank path without a launcher, and the fields of the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_without_a_launcher():
    """`bench.py --gpus 2` alone: two worker processes (here both on device 0 over gloo: one-GPU box), one line, n_gpus 2."""
    out = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--steps", "4", "--warmup", "1", "--no-roofline", "--no-cpu-baseline",
                "--slices-total", "5", "--slices-iters", "2", "--batch", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["parallelism"].startswith("slice-parallel x2")
    sl = out["slices"]                                               # the fixed total sharded over the two ranks: 3 + 2
    assert sl["n_gpus"] == 2 and sl["total_slices"] == 5 and sl["slices_on_rank0"] == 3 and sl["value"] > 0 and sl["scaling"] == "strong"


def test_bench_line_has_roofline_cpu_baseline_and_parity():
    out = _run(["--steps", "6", "--warmup", "2", "--cpu-iters", "6", "--slices-total", "6", "--slices-iters", "3", "--batch", "3"])
    assert out["n_gpus"] == 1 and out["unit"] == "ADMM iters/s" and out["dtype"] == "f32"
    # north_star's second metric rides in the same line (default: 120 slices x 100 iterations; here 6 x 3): slices/s, the batched conv
    # kernel's roofline and the dictionary match against the f16 pipe it runs on (a fraction <= 1)
    sl = out["slices"]
    assert sl["unit"] == "slices/s" and sl["value"] > 0 and sl["total_slices"] == 6 and sl["batch"] == 3 and sl["dict_K"] == 98304
    assert sl["roofline"]["bound"] == "mfma" and 0 < sl["roofline"]["frac"] < 1 and "k_conv6p" in sl["roofline"]["kernel"]
    dmr = sl["dict_match"]["roofline"]
    assert 0 < dmr["frac"] < 1 and dmr["peak"] == 2500.0 and sl["dict_match"]["ms_per_slice"] > 0
    rf, cb, pa = out["roofline"], out["cpu_baseline"], out["parity"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["one_thread_value"] > 0 and cb["cpu_model"]
    assert set(cb["stage_ms_per_iter"]) == {"xupdate", "diagnostics", "denoiser", "elementwise"}
    # parity of the timed slice after the 6 iterations both sides ran (tolerances: DESIGN.md section 7)
    assert pa["admm_iters_compared"] == 6 and pa["tsmi_rel_l2"] < 1e-3 and pa["lsqr_iteration_counts_identical"]
    assert pa["atom_index_identical_frac"] > pa["atom_index_identical_frac_bound_at_this_K"] >= 0.85 and pa["pd_rel_err"] < 1e-3 and pa["tsmi_psnr_db_mean"] > 60
    assert pa["net_rel_l2_random_weights"] < 2e-5 and pa["net_random_weights_scheme"] == [2, 0]      # every level of the network matters here
    wd = out["with_diagnostics"]                                                                      # PnP_ADMM.m:106-109 on the GPU side too
    # (6 steps each: which of the two short timed regions is faster is noise; a sanity bound)
    assert 0 < wd["value"] <= out["value"] * 1.5 and 0 < wd["last_data_fidelity_rel"] < 1 and 0 < wd["last_gt_rel_err"] < 1


def test_bench_slices_fixed_total_sharding():
    """--workload slices --total-slices: a fixed total walked in batches on one GPU (here 4 slices, 3 at a time, 2 iterations)."""
    out = _run(["--workload", "slices", "--total-slices", "4", "--batch", "3", "--steps", "2", "--warmup", "1", "--dict-k", "64", "32",
                "--no-roofline", "--no-cpu-baseline"])
    assert out["scaling"] == "strong" and out["total_slices"] == 4 and out["slices_on_rank0"] == 4 and out["value"] > 0 and out["unit"] == "slices/s"

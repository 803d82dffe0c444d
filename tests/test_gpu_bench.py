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
    out = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--steps", "4", "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--no-secondary",
                "--slices-total", "5", "--slices-iters", "2", "--slices-batch", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["parallelism"].startswith("slice-parallel x2")
    sl = out["slices"]                                               # the fixed total sharded over the two ranks: 3 + 2
    assert sl["n_gpus"] == 2 and sl["total_slices"] == 5 and sl["slices_on_rank0"] == 3 and sl["value"] > 0 and sl["scaling"] == "strong"


def test_bench_many_ranks_rehearsed_on_one_device():
    """The driver's N = 8 run rehearsed on the one-GPU box, as far as the box allows: its process guard admits six GPU processes at once and this
    test session is one of them, so FIVE ranks share device 0 over gloo (10 slices -> 2 per rank, one batch of 2 each).  What the 8-GPU run adds
    over the two-rank test is exercised: a rendezvous of many ranks started by a parent that never touches the GPU, a shard per rank, many contexts
    on one device at once -- both co-residency kernels (k_ks_persist, k_conv6r) must run or fall back cleanly when they do not get the chip to
    themselves -- and the max-over-ranks line.  (N = 8 itself: tests/test_dist_gloo.py, --plumbing-only, on the CPU.)"""
    out = _run(["--gpus", "5", "--backend", "gloo", "--one-device", "--steps", "3", "--warmup", "1", "--no-roofline", "--no-cpu-baseline", "--no-secondary",
                "--slices-total", "10", "--slices-iters", "2", "--slices-batch", "2"], timeout=1200)
    assert out["n_gpus"] == 5 and out["steps"] == 3 and out["value"] > 0 and out["config"]["parallelism"].startswith("slice-parallel x5")
    sl = out["slices"]
    assert sl["n_gpus"] == 5 and sl["total_slices"] == 10 and sl["slices_on_rank0"] == 2 and sl["value"] > 0 and sl["scaling"] == "strong"


def test_bench_line_has_roofline_cpu_baseline_and_parity():
    out = _run(["--steps", "6", "--warmup", "2", "--cpu-iters", "6", "--slices-total", "6", "--slices-iters", "3", "--slices-batch", "3", "--secondary-steps", "4"])
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
    # the roofline is a pure measurement: executed flop of the timed units / the sum of their durations; a forward pass has 2 resident-tile launches
    # + 40 one-launch layers (8 of them with their split-K reduce inside the unit) = 42 units worth ~58 layer-equivalents (head, tail, down conv included)
    assert abs(rf["achieved"] - rf["executed_flop_per_forward"] / (rf["ms_timed_per_forward"] * 1e-3) / 1e12) < 0.02 * rf["achieved"]
    assert 41.5 <= rf["units_per_forward"] <= 42.5 and 56 < rf["layer_equivalents_per_forward"] < 58
    wdn = rf["whole_denoiser"]
    assert 0 < wdn["frac"] <= rf["frac"] and abs(wdn["executed_flop"] - 3 * 213253619712) < 1 and rf["all_conv_launches"]["frac"] <= rf["frac"] * 1.02
    xu = out["xupdate"]                                              # the data-consistency stage against the HBM roofline
    assert xu["roofline"]["bound"] == "hbm" and xu["roofline"]["peak"] == 8000.0 and 0 < xu["roofline"]["frac"] and xu["us_per_lsqr_iteration"] > 0
    assert xu["bytes_per_lsqr_iteration_per_slice"] == 16 * (8 * 11051 * 10 + 2 * 123604) + 4 * 123604
    dw = out["denoiser_weights_timing"]
    assert dw["denoiser_ms_structured_weights"] > 0 and dw["denoiser_ms_random_weights"] > 0 and dw["implied_admm_iters_per_s_with_random_weights"] > 0
    # BASELINE configs[2] and cut0 on the same line
    ep, c0 = out["epi_batch15"], out["cut0"]
    assert ep["slices_per_launch"] == 15 and ep["m"] == 134400 and ep["sampled_k_locations"] == 224 * 224 and ep["value"] > 0
    assert c0["slices_per_launch"] == 1 and c0["T"] == 1000 and c0["m"] > 600000 and c0["value"] > 0
    for o in (ep, c0):
        assert set(o["stage_ms_per_iter"]) == {"xupdate", "denoiser", "elementwise"} and 0 < o["xupdate_share_of_iteration"] < 1
        assert o["xupdate"]["roofline"]["bound"] == "hbm" and 0 < o["xupdate"]["roofline"]["frac"] < 1 and o["xupdate"]["us_per_lsqr_iteration"] > 0
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["one_thread_value"] > 0 and cb["cpu_model"]
    assert set(cb["stage_ms_per_iter"]) == {"xupdate", "diagnostics", "denoiser", "elementwise"}
    # parity of the timed slice after the 6 iterations both sides ran (tolerances: DESIGN.md section 7)
    assert pa["admm_iters_compared"] == 6 and pa["tsmi_rel_l2"] < 1e-3 and pa["lsqr_iteration_counts_identical"]
    assert pa["atom_index_identical_frac"] > pa["atom_index_identical_frac_bound_at_this_K"] >= 0.85 and pa["pd_rel_err"] < 1e-3 and pa["tsmi_psnr_db_mean"] > 60
    assert pa["net_rel_l2_random_weights"] < 2e-5 and pa["net_random_weights_scheme"] == [2, 0]      # every level of the network matters here
    # round 6: a slow or repeated phase must be able to say why -- every phase object carries the library's own health record, zero on a clean run
    clean = {"denoiser_fallbacks": 0, "resident_tile_timeouts": 0, "lsqr_one_launch_timeouts": 0, "repeated_calls": 0}
    for h in (out["health"], sl["health"], ep["health"], c0["health"]):
        assert h["denoiser_scheme"] == "f16x3" and all(h[k] == v for k, v in clean.items()), h
    assert out["health"]["resident_tile_launch_armed"] and out["health"]["lsqr_one_launch"] == "armed" and out["health"]["all_ranks"] == clean
    assert sl["health"]["all_ranks"] == clean and len(sl["rank_seconds"]["per_rank"]) == 1 and sl["rank_seconds"]["max"] >= sl["rank_seconds"]["min"] > 0
    sw = sl["slowest_launch_rank0"]                                  # stage marks of the timed launches: the stages account for the launch
    assert set(sw["stage_ms"]) == {"xupdate", "denoiser", "elementwise", "diagnostics"} and sw["stage_ms"]["denoiser"] > sw["stage_ms"]["xupdate"] > 0
    assert 0.5 * sw["library_wall_ms"] < sw["stages_sum_ms"] <= sw["library_wall_ms"] * 1.001
    # ... the cold start of ONE reconstruction (the reference's unit of work) is on the line
    cs = out["cold_start"]
    assert set(cs["setup_ms"]) >= {"qmri_create", "qmri_set_operator", "qmri_set_denoiser", "qmri_set_dictionary", "total"} and cs["dict_K"] == 98304
    assert cs["time_to_first_slice_s"] > cs["first_reconstruction_s"] > 0 and cs["same_reconstruction_again_s"] > 0
    assert set(cs["setup_ms"]["qmri_set_denoiser_split"]) == {"pack_and_upload", "tensors_and_buffers", "calibration_probe"}
    # ... the x-update's roofline object says what binds it and carries PMC traffic where a committed pass exists for the configuration
    for xo in (xu, ep["xupdate"], c0["xupdate"]):
        r = xo["roofline"]
        # (no `traffic: null` in these objects -- the committed PMC passes cover the headline, configs[2] and cut0)
        assert r["effective_gbs"] == r["achieved"] and r["binding_limit"].startswith("latency") and r["traffic"] > 0 and 0 < r["traffic_over_algorithmic"] < 2
    assert cb["slices_K"] == 98304 and cb["dict_match_s_at_slices_K"] > 0
    mcx = out["cut0_multicoil8"]                                     # configs[4]'s multi-coil part: a labelled extension with its own figure
    assert mcx["value"] > 0 and mcx["workload"].startswith("EXTENSION") and mcx["lsqr_iters_mean"] > 0 and mcx["health"]["repeated_calls"] == 0
    wd = out["with_diagnostics"]                                                                      # PnP_ADMM.m:106-109 on the GPU side too
    # (6 steps each: which of the two short timed regions is faster is noise; a sanity bound)
    assert 0 < wd["value"] <= out["value"] * 1.5 and 0 < wd["last_data_fidelity_rel"] < 1 and 0 < wd["last_gt_rel_err"] < 1


def test_bench_slices_fixed_total_sharding():
    """--workload slices --total-slices: a fixed total walked in batches on one GPU (here 4 slices, 3 at a time, 2 iterations)."""
    out = _run(["--workload", "slices", "--total-slices", "4", "--batch", "3", "--steps", "2", "--warmup", "1", "--dict-k", "64", "32",
                "--no-roofline", "--no-cpu-baseline"])
    assert out["scaling"] == "strong" and out["total_slices"] == 4 and out["slices_on_rank0"] == 4 and out["value"] > 0 and out["unit"] == "slices/s"

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
    out = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--steps", "4", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["parallelism"].startswith("slice-parallel x2")


def test_bench_line_has_roofline_cpu_baseline_and_parity():
    out = _run(["--steps", "6", "--warmup", "2", "--cpu-iters", "6"])
    assert out["n_gpus"] == 1 and out["unit"] == "ADMM iters/s" and out["dtype"] == "f32"
    rf, cb, pa = out["roofline"], out["cpu_baseline"], out["parity"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["one_thread_value"] > 0 and cb["cpu_model"]
    assert set(cb["stage_ms_per_iter"]) == {"xupdate", "diagnostics", "denoiser", "elementwise"}
    # parity of the timed slice after the 6 iterations both sides ran (tolerances: DESIGN.md section 7)
    assert pa["admm_iters_compared"] == 6 and pa["tsmi_rel_l2"] < 1e-3 and pa["lsqr_iteration_counts_identical"]
    assert pa["atom_index_identical_frac"] > 0.99 and pa["pd_rel_err"] < 1e-3 and pa["tsmi_psnr_db_mean"] > 60

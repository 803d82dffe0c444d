"""CPU: the N > 1 path (slice sharding + barrier + max-over-ranks timing) with world_size 2 on gloo."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from qmri_pnp_recon_poc_amd.dist import Group
    from qmri_pnp_recon_poc_amd.batch import shard_slices
    g = Group("gloo")
    mine = shard_slices(15, g.world, g.rank)
    # each rank "reconstructs" its slices; rank 1 is slower: the reported time must be the slowest rank's
    dt = g.timed(lambda: time.sleep(0.05 * (1 + 3 * g.rank)))
    tot = g.max_over_ranks(float(len(mine)))
    print(json.dumps({"rank": g.rank, "world": g.world, "slices": mine, "dt": dt, "maxlen": tot}), flush=True)
    g.close()
""") % ROOT


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    import json
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["slices"] == list(range(0, 8)) and outs[1]["slices"] == list(range(8, 15))
    assert abs(outs[0]["dt"] - outs[1]["dt"]) < 1e-9 and outs[0]["dt"] >= 0.2     # max over ranks, identical on both
    assert outs[0]["maxlen"] == 8.0


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher must itself become 2 ranks (the parent never touches the GPU, it starts
    N workers with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and relays rank 0's line).  --plumbing-only keeps the engine out,
    so the rank start-up, rendezvous, barrier and max-over-ranks run here without a GPU; tests/test_gpu_bench.py runs the
    real two-rank bench on the GPU box."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # one JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["plumbing_only"] is True and out["steps"] == 3
    assert out["max_rank_seconds"] >= 0.02                 # the slower rank (rank 1 sleeps 20 ms) sets the time
    # north_star's slices workload: a FIXED total of 120 slices sharded over the ranks (the same job at every N), 15 at a time per GPU
    for n, on0, batches in ((1, 120, 8), (2, 60, 4), (3, 40, 3), (8, 15, 1)):     # (8: one batch of 15 per GPU, BASELINE configs[3])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--plumbing-only", "--workload", "slices",
                            "--total-slices", "120", "--batch", "15"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == n and out["scaling"] == "strong" and out["slices_total"] == 120 and out["slices_owned_once"] == 120
        assert out["slices_on_rank0"] == on0 and out["batches_on_rank0"] == batches
    # a failing rank must take the job down with a non-zero exit code, not hang it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--no-roofline", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0                           # no GPU here: the workers refuse (no CPU path), the parent reports it

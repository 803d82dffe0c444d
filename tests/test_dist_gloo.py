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

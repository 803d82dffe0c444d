#!/usr/bin/env python3
"""Provoke a slow `slices` phase on purpose and keep the line it produces: the same short slices phase of bench.py three times
on one box -- alone, beside a second process that keeps the CUs busy with fp16 GEMMs (what another tenant of the device looks like), and beside
a process that churns pinned host memory and host<->device copies (pressure on the copy path the results leave by).  For every run: slices/s, the
host clock of every launch, the health counters and the stage split of the slowest launch -- the fields that must name the cause.

    python3 tools/provoke_straggler.py > gpurun_out/provoke.txt          (on the GPU box; ~2 minutes)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOG_GEMM = r"""
import sys, time, torch
torch.cuda.init()
a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
print('hog ready', flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
"""
HOG_COPY = r"""
import sys, time, torch
torch.cuda.init()
d = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
print('hog ready', flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    h = torch.empty(1 << 28, dtype=torch.uint8, pin_memory=True)      # 256 MB of pinned memory allocated and released every round
    for _ in range(4):
        d.copy_(h, non_blocking=True); h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    del h
"""
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-cold-start", "--no-roofline",
         "--slices-total", "30", "--slices-iters", "100", "--slices-batch", "15"]


def run(label, hog_src, seconds=75):
    hog = None
    if hog_src:
        hog = subprocess.Popen([sys.executable, "-c", hog_src, str(seconds)], stdout=subprocess.PIPE, text=True)
        hog.stdout.readline()                                       # 'hog ready'
    t0 = time.time()
    r = subprocess.run(BENCH, capture_output=True, text=True, timeout=900)
    dt = time.time() - t0
    if hog:
        hog.terminate()
        hog.wait()
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        print(json.dumps({"run": label, "error": r.stderr[-1500:]}))
        return
    o = json.loads(line[0])
    sl = o["slices"]
    print(json.dumps({"run": label, "bench_wall_s": round(dt, 1), "admm_iters_per_s": o["value"], "headline_health": {k: v for k, v in o["health"].items() if k != "all_ranks"},
                      "slices_per_s": sl["value"], "launch_seconds_rank0": sl["launch_seconds_rank0"], "rank_seconds": sl["rank_seconds"]["per_rank"],
                      "health": {k: v for k, v in sl["health"].items() if k not in ("what",)}, "slowest_launch_rank0": {k: v for k, v in sl["slowest_launch_rank0"].items() if k != "what"},
                      "stderr_tail": [l for l in r.stderr.splitlines() if "libqmri" in l][-6:]}), flush=True)


if __name__ == "__main__":
    run("alone", None)
    run("beside a process running fp16 GEMMs on the same device", HOG_GEMM)
    run("beside a process churning pinned memory and host<->device copies", HOG_COPY)

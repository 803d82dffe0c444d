import json, os, subprocess, sys, time
ROOT = sys.argv[1]
HOG = r"""
import sys, time, torch
torch.cuda.init()
a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
print('hog ready', flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
"""
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary", "--no-cold-start", "--no-slices"]
for hog in (False, True):
    for dbg in ("", "lsqr_persist=0", "conv_resident=0", "lsqr_persist=0,conv_resident=0"):
        h = None
        if hog:
            h = subprocess.Popen([sys.executable, "-c", HOG, "120"], stdout=subprocess.PIPE, text=True); h.stdout.readline()
        r = subprocess.run(BENCH, capture_output=True, text=True, env=dict(os.environ, QMRI_DEBUG=dbg), timeout=600)
        if h: h.terminate(); h.wait()
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("hog" if hog else "alone", dbg, "FAILED", r.stderr[-300:]); continue
        o = json.loads(line[0])
        print(("beside GEMMs" if hog else "alone       "), "%-32s" % (dbg or "(default)"), "%8.1f it/s" % o["value"], o.get("stage_ms_per_iter"), {k: o["health"][k] for k in ("denoiser_scheme", "lsqr_one_launch", "resident_tile_launch_armed", "lsqr_one_launch_timeouts", "resident_tile_timeouts", "repeated_calls")}, flush=True)

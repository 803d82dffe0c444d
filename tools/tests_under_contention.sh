#!/bin/bash
# The GPU parity tests beside a second process that keeps the device busy with fp16 GEMMs (round 6: the straggler provocation found a set-up race that
# only showed beside another tenant; this runs the parity suite in that situation).  Usage (GPU box):  bash tools/tests_under_contention.sh [seconds] [pytest args]
SECS=${1:-1000}; shift
python3 - $SECS <<'PY' &
import sys, time, torch
torch.cuda.init()
a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
PY
HOG=$!
sleep 8
timeout -k 10 $SECS python3 -m pytest "$@" -q -m gpu -p no:cacheprovider
RC=$?
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
echo "pytest rc $RC"

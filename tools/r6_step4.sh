#!/bin/bash
set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${1:-r6f}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_operator.py -x -q -m gpu -k "multi_coil" > $OUT/pytest_mc.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_mc.log
tail -15 $OUT/pytest_mc.log
for rep in 1 2; do for b in 15 30; do
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-cold-start --no-roofline --batch $b > /tmp/b.json 2>/tmp/b.err || { echo "bench failed"; tail -3 /tmp/b.err; }
  python3 -c "
import json; o=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0]); s=o['slices']; print('batch $b run $rep:', s['value'], 'slices/s', s['launch_seconds_rank0'])" | tee -a $OUT/ab_slices_per_launch.txt
done; done

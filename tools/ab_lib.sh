#!/bin/bash
# A/B on one box of two builds of the library: the ADMM bench line with QMRI_LIBQMRI=<other build> and with the tree's own, alternating.
# Usage (on the GPU box): tools/ab_lib.sh tools/ab/libqmri_<name>.so [steps]
R=$PWD
OTHER=$1
STEPS=${2:-50}
for rep in 1 2; do
  for which in other tree; do
    if [ $which = other ]; then export QMRI_LIBQMRI=$R/$OTHER; else unset QMRI_LIBQMRI; fi
    timeout -k 10 280 python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 5 --no-slices --no-cpu-baseline > /tmp/abl.json 2>/tmp/abl.err || { echo "bench failed ($which)"; tail -5 /tmp/abl.err; exit 1; }
    python3 - "$which" "$rep" <<'PY'
import json, sys
o = json.loads([l for l in open('/tmp/abl.json') if l.startswith('{')][0])
r = o.get('roofline') or {}
print(f"{sys.argv[1]:5s} (run {sys.argv[2]}): {o['value']:.1f} it/s  {o['ms_per_step']:.4f} ms/step  conv layer {r.get('avg_launch_us')} us  frac {r.get('frac')}  stages {o.get('stage_ms_per_iter')}")
PY
  done
done

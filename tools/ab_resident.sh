#!/bin/bash
# A/B on one box: the ADMM bench line with the resident-tile launch of the full-resolution ResBlocks (k_conv6r) on and off, alternating.
# Usage (on the GPU box): tools/ab_resident.sh [steps] > gpurun_out/ab_resident.txt
R=$PWD
STEPS=${1:-50}
for rep in 1 2; do
  for flag in 0 1; do
    QMRI_CONV_RESIDENT=$flag timeout -k 10 280 python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 5 --no-slices --no-cpu-baseline > /tmp/abr.json 2>/tmp/abr.err || { echo "bench failed (resident $flag)"; tail -5 /tmp/abr.err; exit 1; }
    python3 - "$flag" "$rep" <<'PY'
import json, sys
o = json.loads([l for l in open('/tmp/abr.json') if l.startswith('{')][0])
r = o.get('roofline') or {}
print(f"QMRI_CONV_RESIDENT={sys.argv[1]} (run {sys.argv[2]}): {o['value']:.1f} it/s  {o['ms_per_step']:.4f} ms/step  conv layer {r.get('avg_launch_us')} us  frac {r.get('frac')}  layers timed {r.get('launches_timed')}  "
      f"stages {o.get('stage_ms_per_iter')}  x rel l2 {(o.get('parity') or {}).get('tsmi_rel_l2')}")
PY
  done
done

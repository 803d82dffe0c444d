#!/bin/bash
# A/B on one box of an environment switch of the library (read once per process): the ADMM bench line with VAR=0 and VAR=1, alternating.
# Usage (on the GPU box): tools/ab_env.sh QMRI_RES_HEAD [steps] > gpurun_out/ab_env.txt
R=$PWD
VAR=$1
STEPS=${2:-50}
for rep in 1 2; do
  for flag in 0 1; do
    env $VAR=$flag timeout -k 10 280 python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 5 --no-slices --no-cpu-baseline > /tmp/abe.json 2>/tmp/abe.err || { echo "bench failed ($VAR=$flag)"; tail -5 /tmp/abe.err; exit 1; }
    python3 - "$VAR" "$flag" "$rep" <<'PY'
import json, sys
o = json.loads([l for l in open('/tmp/abe.json') if l.startswith('{')][0])
r = o.get('roofline') or {}
print(f"{sys.argv[1]}={sys.argv[2]} (run {sys.argv[3]}): {o['value']:.1f} it/s  {o['ms_per_step']:.4f} ms/step  conv layer {r.get('avg_launch_us')} us  frac {r.get('frac')}  layers timed {r.get('launches_timed')}  stages {o.get('stage_ms_per_iter')}")
PY
  done
done

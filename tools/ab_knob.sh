#!/bin/bash
# A/B (or sweep) on one box of one knob of the library (QMRI_DEBUG="name=value", csrc/api_core.cpp g_knob_defs): the ADMM bench line for each value,
# the whole list twice, alternating.  Usage (on the GPU box):  tools/ab_knob.sh conv_resident 50 0 1 > gpurun_out/ab_knob.txt
#                                                              tools/ab_knob.sh res_delay 50 0 16 24 32 48
R=$PWD; KNOB=$1; STEPS=${2:-50}; shift 2
[ $# -gt 0 ] || set -- 0 1
for rep in 1 2; do
  for v in "$@"; do
    QMRI_DEBUG="$KNOB=$v" timeout -k 10 280 python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 5 --no-slices --no-secondary --no-cpu-baseline > /tmp/abk.json 2>/tmp/abk.err || { echo "bench failed ($KNOB=$v)"; tail -5 /tmp/abk.err; exit 1; }
    python3 - "$KNOB" "$v" "$rep" <<'PY'
import json, sys
o = json.loads([l for l in open('/tmp/abk.json') if l.startswith('{')][0])
r = o.get('roofline') or {}
print(f"{sys.argv[1]}={sys.argv[2]} (run {sys.argv[3]}): {o['value']:.1f} it/s  {o['ms_per_step']:.4f} ms/step  conv3x3 per layer-equivalent {r.get('us_per_layer_equivalent')} us  frac {r.get('frac')}  "
      f"stages {o.get('stage_ms_per_iter')}  x rel l2 {(o.get('parity') or {}).get('tsmi_rel_l2')}")
PY
  done
done

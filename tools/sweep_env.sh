#!/bin/bash
# an integer environment variable of the library against the ADMM rate on one box:  tools/sweep_env.sh VAR steps v1 v2 ...
R=$PWD; VAR=$1; STEPS=$2; shift 2
for d in "$@"; do
  env $VAR=$d timeout -k 10 200 python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 5 --no-slices --no-cpu-baseline --no-roofline > /tmp/sw.json 2>/tmp/sw.err || { echo "failed at $d"; tail -3 /tmp/sw.err; exit 1; }
  python3 -c "
import json,sys
o=json.loads([l for l in open('/tmp/sw.json') if l.startswith('{')][0]); print('%s=%s: %.1f it/s  %.4f ms/step  stages %s' % (sys.argv[1], sys.argv[2], o['value'], o['ms_per_step'], o.get('stage_ms_per_iter')))" $VAR $d
done

#!/bin/bash
# k_conv6r's pause before the first ring fetch (QMRI_RES_DELAY, units of 64 clocks) against the ADMM rate, on one box
R=$PWD
for d in ${@:-0 16 24 32 40 56}; do
  QMRI_RES_DELAY=$d timeout -k 10 200 python3 $R/bench.py --gpus 1 --steps 50 --warmup 5 --no-slices --no-cpu-baseline --no-roofline > /tmp/sw.json 2>/tmp/sw.err || { echo "failed at $d"; tail -3 /tmp/sw.err; exit 1; }
  python3 -c "
import json,sys
o=json.loads([l for l in open('/tmp/sw.json') if l.startswith('{')][0]); print('QMRI_RES_DELAY=%s: %.1f it/s  %.4f ms/step  denoiser %s' % (sys.argv[1], o['value'], o['ms_per_step'], o.get('stage_ms_per_iter',{}).get('denoiser')))" $d
done

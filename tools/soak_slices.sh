#!/bin/bash
# The default line's `slices` phase several times in a row on one box, with what the line now says about each run (round 6): slices/s, the host clock of
# every launch, per-rank seconds, the health counters and the stage split of the slowest launch.  Usage (GPU box):  bash tools/soak_slices.sh [runs] > gpurun_out/soak.txt
N=${1:-6}
for i in $(seq 1 $N); do
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-cold-start > /tmp/soak_$i.json 2> /tmp/soak_$i.err || { echo "run $i failed"; tail -3 /tmp/soak_$i.err; continue; }
  python3 - $i <<'PY'
import json, sys
o = json.loads([l for l in open('/tmp/soak_%s.json' % sys.argv[1]) if l.startswith('{')][0]); s = o['slices']
h = {k: v for k, v in s['health'].items() if k in ('denoiser_fallbacks', 'resident_tile_timeouts', 'lsqr_one_launch_timeouts', 'repeated_calls')}
w = s['slowest_launch_rank0']
print('run %s: %.2f slices/s (batch %d), launches %s, headline %.1f it/s, counters %s, slowest launch stages %s outside %.1f ms' %
      (sys.argv[1], s['value'], s['batch'], s['launch_seconds_rank0'], o['value'], h, w['stage_ms'], w['outside_the_stages_ms']))
PY
done

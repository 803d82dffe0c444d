#!/bin/bash
# A/B on ONE box of the default bench (ADMM it/s): bash tools/ab_admm.sh <out> <steps> "<ENV=.. LIB>" ...   (LIB "-" = in-tree build)
OUT=$1; STEPS=$2; shift; shift
: > $OUT
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
  [ "$lib" = "-" ] && libenv="" || libenv="QMRI_LIBQMRI=$lib"
    echo "== $spec (run $rep)" >> $OUT
    env $envs $libenv python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-slices 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; s = d.get('stage_ms_per_iter', {})
        print('it/s %.1f  ms/step %.4f  conv launch %.2f us frac %.4f  lsqr mean %.2f  stages %s  with_diag %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d['lsqr_iters_mean'], s, d.get('with_diagnostics', {}).get('value')))
" >> $OUT
done
done
cat $OUT

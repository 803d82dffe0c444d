#!/bin/bash
# A/B of library builds on ONE box: the slices workload (15 slices, few iterations) per build / environment setting.
#   bash tools/ab_slices.sh <out-file> "<ENV=.. LIB>" ...      (LIB = path of a libqmri build, "-" = in-tree)
OUT=$1; shift
: > $OUT
for spec in "$@"; do
  lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
  [ "$lib" = "-" ] && libenv="" || libenv="QMRI_LIBQMRI=$lib"
  for rep in 1 2; do
    echo "== $spec (run $rep)" >> $OUT
    env $envs $libenv python bench.py --workload slices --steps 10 --warmup 1 --no-cpu-baseline --dict-k 64 32 2>>$OUT | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('slices/s %.3f  ms/slice %.2f  conv launch %.1f us  frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']))
" >> $OUT
  done
done
cat $OUT

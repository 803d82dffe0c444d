#!/bin/bash
# A/B on ONE box of the default line's two metrics: bash tools/ab_slices2.sh <out> "<ENV=.. LIB>" ...   (LIB "-" = in-tree build)
OUT=$1; shift
: > $OUT
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec##* }; envs=${spec% *}; [ "$envs" = "$spec" ] && envs=""
  [ "$lib" = "-" ] && libenv="" || libenv="QMRI_LIBQMRI=$lib"
    echo "== $spec (run $rep)" >> $OUT
    env $envs $libenv python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; s = d['slices']
        print('ADMM it/s %.1f  conv launch %.2f us frac %.4f | slices/s %.3f  k_conv6p %.1f us frac %.4f  match %.3f ms' % (d['value'], r['avg_launch_us'], r['frac'], s['value'], s['roofline']['avg_launch_us'], s['roofline']['frac'], s['dict_match']['ms_per_slice']))
" >> $OUT
done
done
cat $OUT

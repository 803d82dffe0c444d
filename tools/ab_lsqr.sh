#!/bin/bash
# per-iteration kernel cost of the LSQR x-update (100 iterations per solve) for environment settings, on one box
OUT=$PWD/$1; shift
R=$PWD; : > $OUT; export TMPDIR=/tmp; cd /tmp
for spec in "$@"; do
  rm -rf /tmp/abl; env $spec rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -- python3 $R/tools/time_lsqr.py > /tmp/abl.log 2>&1
  echo "== $spec" >> $OUT; grep iterations /tmp/abl.log >> $OUT
  python3 - >> $OUT <<'PY'
import csv, glob
for f in glob.glob('/tmp/abl/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_ks_' in r['Name']: print('  %-44s calls %5s avg %9.2f us' % (r['Name'].replace('(anonymous namespace)::','')[:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $OUT

#!/bin/bash
# kernel time of the dictionary match for library builds, on one box: bash tools/ab_dict.sh <out> LIB ...   ("-" = in-tree)
OUT=$PWD/$1; shift
R=$PWD
: > $OUT
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  [ "$lib" = "-" ] && unset QMRI_LIBQMRI || export QMRI_LIBQMRI=$R/$lib
  rm -rf /tmp/abd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abd -- python3 $R/tools/bench_dict.py > /tmp/abd.log 2>&1
  echo "== $lib" >> $OUT
  python3 - >> $OUT <<'PY'
import csv, glob
for f in glob.glob('/tmp/abd/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_dict' in r['Name']: print('  %-40s calls %s avg %.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
cat $OUT

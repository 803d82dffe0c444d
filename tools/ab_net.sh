#!/bin/bash
# Kernel times of one UNetRes forward (single slice) per library build, from a rocprofv3 kernel trace of tools/prof_net.py, on one box:
#   bash tools/ab_net.sh <out> LIB ...     ("-" = in-tree).  Prints the mean duration of the conv kernels and the span of a forward.
OUT=$PWD/$1; shift
R=$PWD
: > $OUT
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  [ "$lib" = "-" ] && unset QMRI_LIBQMRI || export QMRI_LIBQMRI=$R/$lib
  rm -rf /tmp/abn; rocprofv3 --kernel-trace --output-format csv -d /tmp/abn -- python3 $R/tools/prof_net.py 1 6 > /tmp/abn.log 2>&1
  echo "== $lib" >> $OUT
  python3 - >> $OUT <<'PY'
import csv, glob
for f in glob.glob('/tmp/abn/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    conv = [r for r in rows if 'k_conv6<' in r['Kernel_Name']]
    # the last forward = the last 56 launches of k_conv6<...>
    last = conv[-56:]
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last]
    span = (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3
    print('  k_conv6 launches of the last forward: n %d  mean %.2f us  min %.2f  max %.2f;  first start to last end %.1f us' % (len(d), sum(d) / len(d), min(d), max(d), span))
PY
done
cat $OUT

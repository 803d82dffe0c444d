#!/bin/bash
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/pd1 /tmp/pd2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pd1 -- python3 $R/tools/bench_dict.py > /tmp/pd1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pd1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_dict_match' in n:
            kind = 'filter+exact (k_dict_match_f, main)' if 'k_dict_match_f<5, false>' in n or 'Lb0E' in n else 'seed (k_dict_match_f, SEED)' if 'k_dict_match_f' in n else 'exact products only (k_dict_match)'
            acc[(kind, r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print('%-40s %-28s %.4g  (%d launches)' % (k[0], k[1], sum(v)/len(v), len(v)))
PY
tail -3 /tmp/pd1.log

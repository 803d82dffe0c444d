#!/bin/bash
# The wide dictionary match (dictw_kernels.hip, s = 1000, K = 98 304, one 224 x 224 slice) under rocprofv3, four separate passes as
# MI355X_MICROARCH.md prescribes (counters never together with the trace statistics; FETCH_SIZE and WRITE_SIZE in passes of their own):
#   bash tools/pmc_dictw.sh <out-prefix>      -> <prefix>_kernel_stats.csv, <prefix>_pmc.txt
P=${1:-gpurun_out/r04_dictw}
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/dw0 /tmp/dw1 /tmp/dw2 /tmp/dw3
CMD="python3 $R/tools/bench_dict.py --s 1000 --reps 1 --dev-reps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dw0 -- $CMD > /tmp/dw0.log 2>&1 || echo "trace pass failed"
cp $(find /tmp/dw0 -name "*kernel_stats.csv" | head -1) $R/${P}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/dw1 -- $CMD > /tmp/dw1.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/dw2 -- $CMD > /tmp/dw2.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/dw3 -- $CMD > /tmp/dw3.log 2>&1 || echo "write pass failed"
python3 - > $R/${P}_pmc.txt <<'PY'
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n): return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
for d in ('/tmp/dw1', '/tmp/dw2', '/tmp/dw3'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r['Kernel_Name'])
            if n.startswith('k_dictw') or n.startswith('k_dict_merge'): cnt[n][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('/tmp/dw1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n.startswith('k_dictw') or n.startswith('k_dict_merge'): dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('# wide dictionary match, s = 1000, K = 98 304, 50 176 pixels; per kernel: launches, mean duration under the SQ counter pass, effective clock =')
print('# GRBM_GUI_ACTIVE / 8 XCDs / duration, matrix-core busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles per XCD); FETCH_SIZE / WRITE_SIZE in KB as counted')
print('# (gfx950: FETCH_SIZE counts 16-byte-per-lane streaming reads at half their bytes -- MI355X_MICROARCH.md, HBM)')
for n in sorted(cnt):
    c = {k: sum(v) / len(v) for k, v in cnt[n].items()}
    d = sum(dur[n]) / max(len(dur[n]), 1)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    print('%-18s n %3d  %10.1f us  clock %.2f GHz  MFMA busy %.3f  VALU %.3f  LDS-inst %.3f  issue-wait %.2f  parked %.2f   FETCH_SIZE %.0f KB  WRITE_SIZE %.0f KB' % (
        n, len(dur[n]), d / 1e3, gui / d if d else 0, c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / gui if gui else 0,
        4 * c.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / gui if gui else 0, 4 * c.get('SQ_ACTIVE_INST_LDS', 0) / 1024 / gui if gui else 0,
        c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1), c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1),
        c.get('FETCH_SIZE', 0), c.get('WRITE_SIZE', 0)))
PY
grep -h "^{" /tmp/dw0.log | tail -1 > $R/${P}_bench_under_rocprof.json
cat $R/${P}_pmc.txt

#!/bin/bash
# Matrix-core utilisation and effective clock of the conv kernels from one rocprofv3 --pmc pass (SQ + GRBM counters only), per kernel name:
#   bash tools/pmc_conv.sh <batch> <out.txt>        (tools/prof_net.py <batch> 3)
B=${1:-15}; OUT=$PWD/${2:-gpurun_out/pmc_conv.txt}
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pc1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc1 -- python3 $R/tools/prof_net.py $B 3 > /tmp/pc1.log 2>&1
python3 - > $OUT <<'PY'
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('/tmp/pc1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        if n.startswith('k_conv6'):
            cnt[n][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('/tmp/pc1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        if n.startswith('k_conv6'): dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('# per kernel: launches, mean duration (under the counter pass), effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, matrix-core busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI cycles per XCD)')
for n in sorted(cnt):
    c = {k: sum(v) / len(v) for k, v in cnt[n].items()}
    d = sum(dur[n]) / max(len(dur[n]), 1)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    clk = gui / d if d else 0
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
    print('%-34s n %4d  %8.1f us  clock %.2f GHz  MFMA busy %.3f of the launch  (VALU %.3f, LDS-inst %.3f, wave-cycles waiting on issue %.2f, parked %.2f of wave cycles)' % (
        n[:34], len(dur[n]), d / 1e3, clk, mf / gui if gui else 0, 4 * c.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / gui if gui else 0, 4 * c.get('SQ_ACTIVE_INST_LDS', 0) / 1024 / gui if gui else 0,
        c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1), c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1)))
PY
cat $OUT

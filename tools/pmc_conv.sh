#!/bin/bash
# Matrix-core utilisation and effective clock of the conv kernels from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/pc1 -- python3 $R/tools/prof_net.py $B 3 > /tmp/pc1.log 2>&1
python3 - > $OUT <<'PY'
import csv, glob, collections
# Denominators (round 5): the kernel's own DURATION from the kernel trace of the same pass, never GRBM_GUI_ACTIVE (it counts cycles outside the
# kernel: round 4's table showed "clocks" of 2.8 - 6.4 GHz on a 2.4 GHz part for the short launches).
#   matrix-core busy, against the roofline's own denominator:  SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)   -- needs no clock estimate
#   clock the CUs held while busy (a LOWER bound):            4 x SQ_BUSY_CU_CYCLES (quad-cycles, summed over CUs) / (CUs in use x duration); CUs in use =
#                                                              min(256, workgroups) -- a CU that is idle part of the launch lowers it, nothing can raise it above the clock
#   matrix-core busy at that clock:                            the first figure x 2.4 / that clock   (an UPPER bound of the busy share of the cycles the chip actually ran)
PEAK_GHZ, NSIMD, NCU = 2.4, 1024, 256
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
grid = {}
def short(n): return n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
for f in glob.glob('/tmp/pc1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n.startswith('k_conv6'):
            cnt[n][r['Counter_Name']].append(float(r['Counter_Value']))
            try: grid[n] = int(r.get('Grid_Size', 0)) // max(int(r.get('Workgroup_Size', 1)), 1)
            except ValueError: pass
for f in glob.glob('/tmp/pc1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n.startswith('k_conv6'): dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('# per kernel (one rocprofv3 --pmc pass, SQ counters only; durations from the kernel trace of the same pass): launches, mean duration,')
print('#   MFMA busy vs peak = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)  [the quantity roofline.frac measures, by counter];')
print('#   clock >= 4 x SQ_BUSY_CU_CYCLES / (CUs in use x duration)  [lower bound of the clock held, never above 2.4];  busy at that clock <= the first x 2.4 / clock')
for n in sorted(cnt):
    c = {k: sum(v) / len(v) for k, v in cnt[n].items()}
    d = sum(dur[n]) / max(len(dur[n]), 1)                         # ns
    if not d: continue
    cus = min(NCU, grid.get(n, NCU) or NCU)
    mf_peak = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (NSIMD * d * PEAK_GHZ)
    clk = 4 * c.get('SQ_BUSY_CU_CYCLES', 0) / (cus * d)
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    print('%-34s n %4d  %8.1f us  MFMA busy %.3f of peak-clock cycles  clock >= %.2f GHz (%d CUs)  busy at that clock <= %.3f  (VALU inst %.3f, LDS inst %.3f of peak-clock SIMD cycles; '
          'wave cycles: waiting on issue %.2f, parked %.2f)' % (
        n[:34], len(dur[n]), d / 1e3, mf_peak, clk, cus, mf_peak * PEAK_GHZ / clk if clk else 0, 4 * c.get('SQ_ACTIVE_INST_VALU', 0) / (NSIMD * d * PEAK_GHZ),
        4 * c.get('SQ_ACTIVE_INST_LDS', 0) / (NSIMD * d * PEAK_GHZ), c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc))
PY
cat $OUT

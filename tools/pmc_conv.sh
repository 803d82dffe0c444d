#!/bin/bash
# Matrix-core busy share and the clock held by the conv kernels, per kernel name, from ONE counter pass (SQ counters only) of rocprofv3:
#   bash tools/pmc_conv.sh <batch> <out.txt>        (profiles tools/prof_net.py <batch> 3; run through gpurun from the repository root)
B=${1:-15}; OUT=$PWD/${2:-gpurun_out/pmc_conv.txt}
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pc1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pc1 -- python3 $R/tools/prof_net.py $B 3 > /tmp/pc1.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/pc1.log; exit 1; }
python3 - > $OUT <<'PY'
import csv, glob, collections
# Denominators (round 5): the kernel's own DURATION from the kernel trace of the same pass, never GRBM_GUI_ACTIVE (it counts cycles outside the
# kernel: round 4's table showed "clocks" of 2.8 - 6.4 GHz on a 2.4 GHz part for the short launches).
#   matrix-core busy, against the roofline's own denominator:  SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)   -- needs no clock estimate
#   clock the waves ran at (a LOWER bound):                    4 x SQ_WAVE_CYCLES (quad-cycles, summed over waves) / (waves x duration), for launches whose workgroups are all
#                                                              resident from start to end (<= 256 workgroups, one per CU: every kernel of the 3x3 family); a wave that
#                                                              starts late or ends early lowers it, nothing can raise it above the clock.  (SQ_BUSY_CU_CYCLES was tried
#                                                              for this first and is NOT per-CU quad-cycles: it gave 7 - 10 "GHz"; dropped.)
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
            try: grid[n] = (int(r.get('Grid_Size', 0)) // max(int(r.get('Workgroup_Size', 1)), 1), int(r.get('Workgroup_Size', 64)) // 64)
            except ValueError: pass
for f in glob.glob('/tmp/pc1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r['Kernel_Name'])
        if n.startswith('k_conv6'): dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('# per kernel (one rocprofv3 --pmc pass, SQ counters only; durations from the kernel trace of the same pass): launches, mean duration,')
print('#   MFMA busy vs peak = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)  [the quantity roofline.frac measures, by counter];')
print('#   clock >= 4 x SQ_WAVE_CYCLES / (waves x duration), launches of <= 256 workgroups only  [lower bound of the clock the waves ran at];  busy at that clock <= the first x 2.4 / clock')
for n in sorted(cnt):
    c = {k: sum(v) / len(v) for k, v in cnt[n].items()}
    d = sum(dur[n]) / max(len(dur[n]), 1)                         # ns
    if not d: continue
    wgs, wpw = grid.get(n, (NCU, 8))
    mf_peak = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (NSIMD * d * PEAK_GHZ)
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    # (launches under 15 us: the dispatch timestamps' own granularity is several percent of the duration -- a 12 us launch of 128 workgroups read 2.78 "GHz": no figure)
    clk = 4 * wc / (wgs * wpw * d) if (0 < wgs <= NCU and d >= 15000) else 0.0
    ctxt = ('clock >= %.2f GHz (%d workgroups x %d waves)  busy at that clock <= %.3f' % (clk, wgs, wpw, mf_peak * PEAK_GHZ / clk) if clk else
            'clock: n/a (%d workgroups%s)' % (wgs, ': not all resident throughout' if wgs > NCU else ', launch under 15 us'))
    print('%-34s n %4d  %8.1f us  MFMA busy %.3f of peak-clock cycles  %s  (VALU inst %.3f, LDS inst %.3f of peak-clock SIMD cycles; '
          'wave cycles: waiting on issue %.2f, parked %.2f)' % (
        n[:34], len(dur[n]), d / 1e3, mf_peak, ctxt, 4 * c.get('SQ_ACTIVE_INST_VALU', 0) / (NSIMD * d * PEAK_GHZ),
        4 * c.get('SQ_ACTIVE_INST_LDS', 0) / (NSIMD * d * PEAK_GHZ), c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc))
PY
cat $OUT

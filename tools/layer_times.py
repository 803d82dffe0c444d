#!/usr/bin/env python3
"""Per-layer kernel durations of ONE UNetRes forward from a rocprofv3 --kernel-trace of tools/prof_net.py (last forward of the run):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -- python3 tools/prof_net.py 1 6
   python tools/layer_times.py gpurun_out/lt
Prints every launch of the last forward in order (duration, gap to the previous launch) and the means by kernel / residual operand."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# forwards are delimited by k_act_check (last kernel of a forward)
ends = [i for i, r in enumerate(rows) if 'k_act_check' in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
fw = rows[lo:hi]
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:44]
tot = (fw[-1][1] - fw[0][0]) / 1e3
print(f'last forward: {len(fw)} launches, {tot:.1f} us from first start to last end, sum of kernel durations {sum(e - s for s, e, _ in fw) / 1e3:.1f} us')
by = collections.defaultdict(list)
prev_end = None
c3 = 0
for s, e, n in fw:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = short(n)
    tag = ''
    if name.startswith('k_conv6<') or name.startswith('k_conv6p'):
        c3 += 1
    print(f'  {name:44s} {(e - s) / 1e3:8.2f} us   gap {gap:6.2f}')
    by[name].append(((e - s) / 1e3, gap))
    prev_end = e
print('means:')
for k, v in by.items():
    print(f'  {k:44s} n {len(v):3d}  dur {sum(x for x, _ in v) / len(v):7.2f}  gap before {sum(g for _, g in v) / len(v):6.2f}')
